// rt-cli -- command line front end over librt_hip.so with the reference CLI's flags
// (raytracer-cli/src/main.rs:19-42), defaults (raytracer-lib/src/prelude.rs:51-59) and
// call sequence (main.rs:51-104): read scene -> parse_scene -> render -> save.
//
//   rt-cli -i scene.json [-o out.png] [-x W] [-y H] [-r RPP] [-d DEPTH] [--aspect-ratio A]
//          [--disable-shadows] [--render-normals] [--antialias-method normal|jittered|random] [--seed N]
//          [--backend hip] [--device N] [--gpus N [--same-device]] [--mode fast|reference]
//          [--stats-json FILE] [--dump-ids FILE]
//
// Only the HIP backend exists in this program; there is no CPU renderer to fall back to.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include <unistd.h>

#include "../include/rt_hip.h"

namespace {

int fail(const std::string &msg) {
    std::fprintf(stderr, "Error: %s\n", msg.c_str());
    return 1;
}

int fail_rt(const char *what) {
    std::fprintf(stderr, "Error: %s: %s\n", what, rt_last_error());
    return 1;
}

void usage() {
    std::fprintf(stderr,
                 "Raytracer CLI (MI355X backend)\n\n"
                 "Usage: rt-cli [OPTIONS] --scene-path <SCENE_PATH>\n\n"
                 "Options:\n"
                 "  -x, --width <WIDTH>\n  -y, --height <HEIGHT>\n  -i, --scene-path <SCENE_PATH>\n"
                 "  -o, --output <OUTPUT_PATH>          [default: out.png]\n"
                 "  -r, --rays-per-pixel <N>            a perfect square [default: 4]\n"
                 "  -d, --recursion-depth <DEPTH>       [default: 3]\n"
                 "      --aspect-ratio <ASPECT_RATIO>\n      --disable-shadows\n      --render-normals\n"
                 "      --antialias-method <normal|jittered|random>\n"
                 "      --seed <N>                      key of the jittered / random sample offsets\n"
                 "                                      [default: drawn from the operating system, as the\n"
                 "                                      reference's thread_rng is]\n"
                 "      --backend <hip>                 [default: hip]\n"
                 "      --device <ORDINAL>              [default: 0]\n"
                 "      --mode <fast|reference>         traversal order [default: fast]\n"
                 "      --stats-json <FILE>             ray counts and kernel times of the frame\n"
                 "      --gpus <N>                      split the frame over devices --device .. --device + N - 1 [default: 1]\n"
                 "      --same-device                   with --gpus: all parts on --device (checks the split on a one-GPU machine)\n"
                 "      --dump-ids <FILE>               per-sample hit ids of the frame (binary, see rt_cli.cpp)\n");
}

}  // namespace

int main(int argc, char **argv) {
    RtParseArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    pa.recursion_depth = -1;
    std::string scene_path, output = "out.png", stats_json, dump_ids, backend = "hip";
    long rays_per_pixel = 4;  // DEFAULT_RAYS_PER_PIXEL, prelude.rs:56
    int aa = RT_AA_NORMAL, mode = RT_MODE_FAST, device = 0;
    long gpus = 1;
    bool same_device = false;
    bool have_seed = false;
    uint64_t seed = 0;

    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "Error: a value is required for '%s'\n", name);
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "-x" || a == "--width") pa.image_width = (uint32_t)std::strtoul(value("--width"), nullptr, 10);
        else if (a == "-y" || a == "--height") pa.image_height = (uint32_t)std::strtoul(value("--height"), nullptr, 10);
        else if (a == "-i" || a == "--scene-path") scene_path = value("--scene-path");
        else if (a == "-o" || a == "--output") output = value("--output");
        else if (a == "-r" || a == "--rays-per-pixel") rays_per_pixel = std::strtol(value("--rays-per-pixel"), nullptr, 10);
        else if (a == "-d" || a == "--recursion-depth") pa.recursion_depth = (int32_t)std::strtol(value("--recursion-depth"), nullptr, 10);
        else if (a == "--aspect-ratio") pa.aspect_ratio = std::strtod(value("--aspect-ratio"), nullptr);
        else if (a == "--disable-shadows") pa.disable_shadows = 1;
        else if (a == "--render-normals") pa.render_normals = 1;
        else if (a == "--antialias-method") {
            const std::string v = value("--antialias-method");
            if (v == "normal") aa = RT_AA_NORMAL;
            else if (v == "jittered") aa = RT_AA_JITTERED;
            else if (v == "random") aa = RT_AA_RANDOM;
            else return fail("invalid value '" + v + "' for '--antialias-method'");
        } else if (a == "--seed") {
            seed = std::strtoull(value("--seed"), nullptr, 10);
            have_seed = true;
        } else if (a == "--backend") backend = value("--backend");
        else if (a == "--device") device = (int)std::strtol(value("--device"), nullptr, 10);
        else if (a == "--gpus") gpus = std::strtol(value("--gpus"), nullptr, 10);
        else if (a == "--same-device") same_device = true;
        else if (a == "--mode") {
            const std::string v = value("--mode");
            if (v == "fast") mode = RT_MODE_FAST;
            else if (v == "reference") mode = RT_MODE_REFERENCE_ORDER;
            else return fail("invalid value '" + v + "' for '--mode'");
        } else if (a == "--stats-json") stats_json = value("--stats-json");
        else if (a == "--dump-ids") dump_ids = value("--dump-ids");
        else if (a == "-h" || a == "--help") {
            usage();
            return 0;
        } else {
            usage();
            return fail("unexpected argument '" + a + "'");
        }
    }
    if (scene_path.empty()) {
        usage();
        return fail("the following required arguments were not provided: --scene-path <SCENE_PATH>");
    }
    if (backend != "hip") return fail("backend '" + backend + "' is not part of this program (only 'hip' is)");
    if (gpus < 1 || gpus > 64) return fail("invalid value for '--gpus'");
    if (gpus > 1 && !dump_ids.empty()) return fail("'--dump-ids' cannot be used with '--gpus'");

    // main.rs:72-83: rays per pixel must be a perfect square
    if (rays_per_pixel <= 0 || rays_per_pixel > 65535) return fail("rays_per_pixel must be a perfect square");
    const long n = (long)std::sqrt((double)rays_per_pixel);
    if (n * n != rays_per_pixel) return fail("rays_per_pixel must be a perfect square");

    RtScene *scene = nullptr;
    if (rt_parse_scene_file(scene_path.c_str(), &pa, &scene) != RT_OK) return fail_rt("scene");
    RtBackend *be = nullptr;
    if (rt_backend_create(device, &be) != RT_OK) {
        rt_scene_destroy(scene);
        return fail_rt("backend");
    }
    RtDeviceScene *ds = nullptr;
    int rc = rt_scene_upload(be, scene, &ds);
    if (rc != RT_OK) {
        rt_backend_destroy(be);
        rt_scene_destroy(scene);
        return fail_rt("upload");
    }
    // --gpus N: one backend and one copy of the scene per further device (the first is `be` / `ds`)
    std::vector<RtBackend *> all_be(1, be);
    std::vector<RtDeviceScene *> all_ds(1, ds);
    auto destroy_all = [&]() {
        for (size_t k = all_ds.size(); k-- > 0;) rt_scene_free(all_ds[k]);
        for (size_t k = all_be.size(); k-- > 0;) rt_backend_destroy(all_be[k]);
        rt_scene_destroy(scene);
    };
    for (long k = 1; k < gpus; ++k) {
        RtBackend *bk = nullptr;
        RtDeviceScene *dk = nullptr;
        if (rt_backend_create(same_device ? device : device + (int)k, &bk) != RT_OK) {
            destroy_all();
            return fail_rt("backend");
        }
        all_be.push_back(bk);
        if (rt_scene_upload(bk, scene, &dk) != RT_OK) {
            destroy_all();
            return fail_rt("upload");
        }
        all_ds.push_back(dk);
    }
    RtRenderArgs ra;
    std::memset(&ra, 0, sizeof(ra));
    ra.sqrt_rays_per_pixel = (uint16_t)n;
    ra.antialias_method = aa;
    if (!have_seed) {  // antialias.rs:47-48 draws from rand::thread_rng(): a different frame every run
        std::random_device rd;
        seed = ((uint64_t)rd() << 32) | rd();
    }
    ra.seed = seed;
    ra.mode = mode;
    const uint32_t W = scene->image_width, H = scene->image_height;
    std::vector<float> fb((size_t)W * H * 3);
    RtStats st;
    // main.rs:86-100 advances a progress bar from per_pixel_cb; here the count arrives once per
    // finished batch of tiles (shown only on a terminal)
    if (isatty(2))
        ra.progress = [](uint64_t done, uint64_t total, void *) {
            std::fprintf(stderr, "\rRendering: %3.0f %%", total ? 100.0 * (double)done / (double)total : 100.0);
            if (done == total) std::fprintf(stderr, "\n");
        };
    // --dump-ids: the RtRayIds arrays of the frame, for comparing hits with another renderer.  File:
    // "RTID", then u32 version (1), width, height, samples per pixel, point lights, then seven int32
    // arrays: primary_shape, primary_tri, final_shape, final_tri, bounces (one entry per sample, sample
    // index = (i + j * width) * spp + p * n + q), shadow_shape, shadow_tri (point lights entries per sample).
    const size_t samples = (size_t)W * H * (size_t)(n * n);
    const size_t npl = rt_scene_point_light_count(ds);
    std::vector<int32_t> id_store;
    RtRayIds ids;
    std::memset(&ids, 0, sizeof(ids));
    if (!dump_ids.empty()) {
        // ids that are to be compared with another renderer's: the fast mode's shadow rays then run to the closest
        // blocker too (light/point.rs:35) instead of stopping at the first one they find; the picture is the same
        if (mode == RT_MODE_FAST) ra.flags |= RT_RENDER_CLOSEST_BLOCKERS;
        id_store.assign(samples * (5 + 2 * npl), -1);
        int32_t *at = id_store.data();
        ids.primary_shape = at; at += samples;
        ids.primary_tri = at; at += samples;
        ids.final_shape = at; at += samples;
        ids.final_tri = at; at += samples;
        ids.bounces = at; at += samples;
        ids.shadow_shape = at; at += samples * npl;
        ids.shadow_tri = at;
    }
    if (gpus > 1) {
        ra.tile_w = ra.tile_h = 32;
        ra.part0_weight = gpus > 4 ? 7 : 0;  // the first device also collects the tiles
        RtMulti *multi = nullptr;
        rc = rt_multi_create(all_be.data(), all_ds.data(), (uint32_t)gpus, &multi);
        if (rc == RT_OK) rc = rt_render_multi(multi, &ra, fb.data(), &st);
        rt_multi_destroy(multi);
    } else {
        rc = rt_render(be, ds, &ra, fb.data(), dump_ids.empty() ? nullptr : &ids, &st);
    }
    int exit_code = 0;
    if (rc != RT_OK) {
        exit_code = fail_rt("render");
    } else if (rt_save_png(output.c_str(), fb.data(), W, H) != RT_OK) {
        exit_code = fail_rt("save");
    } else {
        const double rays = (double)(st.primary_rays + st.secondary_rays + st.shadow_rays);
        std::fprintf(stderr, "Render complete: %ux%u, %ld rays per pixel, %.0f rays in %.3f ms (%.2f Grays/s)\n", W, H,
                     rays_per_pixel, rays, st.render_ms, rays / (st.render_ms * 1e6));
        if (!dump_ids.empty()) {
            FILE *f = std::fopen(dump_ids.c_str(), "wb");
            const uint32_t head[5] = {1u, W, H, (uint32_t)(n * n), (uint32_t)npl};
            if (!f || std::fwrite("RTID", 1, 4, f) != 4 || std::fwrite(head, sizeof(head), 1, f) != 1 ||
                std::fwrite(id_store.data(), sizeof(int32_t), id_store.size(), f) != id_store.size())
                exit_code = fail("cannot write " + dump_ids);
            if (f) std::fclose(f);
        }
        if (!stats_json.empty()) {
            FILE *f = std::fopen(stats_json.c_str(), "w");
            if (!f) {
                exit_code = fail("cannot open " + stats_json);
            } else {
                std::fprintf(f,
                             "{\"width\": %u, \"height\": %u, \"rays_per_pixel\": %ld, \"primary_rays\": %llu, "
                             "\"secondary_rays\": %llu, \"shadow_rays\": %llu, \"shaded_hits\": %llu, \"render_ms\": %.6f, "
                             "\"kernel_ms\": [%.6f, %.6f, %.6f, %.6f, %.6f, %.6f]}\n",
                             W, H, rays_per_pixel, (unsigned long long)st.primary_rays,
                             (unsigned long long)st.secondary_rays, (unsigned long long)st.shadow_rays,
                             (unsigned long long)st.shaded_hits, st.render_ms, st.kernel_ms[0], st.kernel_ms[1],
                             st.kernel_ms[2], st.kernel_ms[3], st.kernel_ms[4], st.kernel_ms[5]);
                std::fclose(f);
            }
        }
    }
    destroy_all();
    return exit_code;
}
