// hostsim.cpp -- TEST INFRASTRUCTURE.  Compiles the backend's per-thread stage
// bodies (rust-raytracer_amd/csrc/rt_pipeline.h, rt_core.h) and its scene
// flattening (bvh_build.cpp) as ordinary host C++ and runs the stages in plain
// for-loops, one "thread" at a time.
//
// Purpose: exercise the traversal / queue / indexing logic of the device code
// under a CPU debugger and sanitizers, and measure how often its fp32 decisions
// differ from the f64 oracle, in a container that has no GPU.  It is not part
// of librt_hip.so, is never loaded by the package, and is not a fallback: the
// product path fails when the HIP library or a GPU is missing.
//
// fp32 arithmetic here is the host's (SSE, no contraction) and matches the
// device's IEEE add/mul/div/sqrt; powf may differ in the last bits.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

// per-thread work tallies, summed by hostsim_tally()
struct Tally { unsigned long long node = 0, tri = 0, tprim = 0; };
static thread_local Tally g_tally;
#define RT_TALLY(what) (++g_tally.what)
// optional record of every node / triangle index a ray touches (divergence studies)
static thread_local std::vector<int32_t> *g_visit[2];
#define RT_VISIT(kind, index) do { if (g_visit[kind]) g_visit[kind]->push_back((int32_t)(index)); } while (0)

#include "../../include/rt_hip.h"
#include "../../rust-raytracer_amd/csrc/bvh_build.h"
#include "../../rust-raytracer_amd/csrc/rt_pipeline.h"
#include "../../rust-raytracer_amd/csrc/rt_tree.h"

// Optional per-ray work recording (nodes fetched, triangles tested), for divergence studies:
// [0] camera rays per sample, [1] shadow rays per (light, sample); null = off.
static uint16_t *g_work_nodes[2], *g_work_tris[2];
extern "C" void hostsim_record_work(uint16_t *primary_nodes, uint16_t *primary_tris, uint16_t *shadow_nodes,
                                    uint16_t *shadow_tris) {
    g_work_nodes[0] = primary_nodes; g_work_tris[0] = primary_tris;
    g_work_nodes[1] = shadow_nodes; g_work_tris[1] = shadow_tris;
}

namespace {
struct VecStack;
}
extern "C" int hostsim_wave_union(const RtScene *scene, const RtRenderArgs *args, int light, double *out);

namespace {

struct VecStack {
    std::vector<int32_t> v;
    float aux[6];
    void set(int i, int32_t x) { v[(size_t)i] = x; }
    int32_t get(int i) const { return v[(size_t)i]; }
    void set_aux(int i, float x) { aux[i] = x; }
    float get_aux(int i) const { return aux[i]; }
};

struct HostScene {
    rt::FlatScene flat;
    std::vector<int32_t> point_lights;
    RtSceneView view{}, view_fast{};  // reference-order / fast mode (different scene-level trees)
    bool has_mirror = false, has_lit = false, has_ggx = false;
};

// what rt_scene_upload_with's options select on the device: set by hostsim_upload_options
uint32_t g_upload_flags = 0;

std::string make_view(const RtScene &scene, HostScene &hs) {
    std::string err = rt::flatten_scene(scene, hs.flat, (g_upload_flags & RT_UPLOAD_REFERENCE_MESH_TREES) != 0);
    if (!err.empty()) return err;
    for (uint32_t i = 0; i < scene.n_lights; ++i)
        if (scene.lights[i].type == RT_LIGHT_POINT) hs.point_lights.push_back((int32_t)i);
    RtSceneView &V = hs.view;
    V.nodes = reinterpret_cast<const rt_f4 *>(hs.flat.nodes.data());
    V.tprims = reinterpret_cast<const rt_f4 *>(hs.flat.tprims.data());
    V.tris = reinterpret_cast<const rt_f4 *>(hs.flat.tris.data());
    V.tri_xf = reinterpret_cast<const rt_f4 *>(hs.flat.tri_xf.data());
    V.tprim_xf = reinterpret_cast<const rt_f4 *>(hs.flat.tprim_xf.data());
    V.shapes = scene.shapes;
    V.protos = scene.protos;
    V.verts64 = hs.flat.verts64.data();
    V.mesh_tri_base = hs.flat.mesh_tri_base.data();
    V.shaders = scene.shaders;
    V.lights = scene.lights;
    V.point_lights = hs.point_lights.data();
    V.camera = scene.camera;
    std::memcpy(V.background, scene.background, sizeof(V.background));
    V.n_lights = (int32_t)scene.n_lights;
    V.n_point_lights = (int32_t)hs.point_lights.size();
    V.tlas_root = hs.flat.tlas_root;
    V.recursion_depth = scene.recursion_depth;
    V.disable_shadows = scene.disable_shadows;
    V.stack_depth = hs.flat.stack_depth;
    hs.view_fast = V;
    hs.view_fast.tlas_root = hs.flat.tlas_root_fast;
    hs.view_fast.stack_depth = hs.flat.stack_depth_fast;
    for (uint32_t s = 0; s < scene.n_shapes; ++s) {
        const int32_t t = scene.shaders[scene.shapes[s].shader].type;
        hs.has_mirror = hs.has_mirror || t == RT_SHADER_PERFECT_MIRROR;
        hs.has_lit = hs.has_lit || t == RT_SHADER_LAMBERTIAN || t == RT_SHADER_BLINN_PHONG;
        hs.has_ggx = hs.has_ggx || t == RT_SHADER_GGX_MIRROR;
    }
    return "";
}

template <typename T>
T *grab(std::vector<std::vector<char>> &pool, size_t count) {
    pool.emplace_back(std::max<size_t>(count, 1) * sizeof(T));
    return reinterpret_cast<T *>(pool.back().data());
}

template <bool ORDERED, bool ANYHIT, bool EXACT>
void run(const HostScene &hs, const RtScene &scene, const RtRenderArgs &args, float *fb, const RtRayIds *ids,
         RtStats *stats) {
    const RtSceneView &S = EXACT ? hs.view : hs.view_fast;
    const uint32_t n = args.sqrt_rays_per_pixel, spp = n * n;
    const uint32_t W = scene.image_width, H = scene.image_height;
    const uint32_t tw = args.tile_w ? args.tile_w : 32, th = args.tile_h ? args.tile_h : 32;
    const uint32_t tiles_x = (W + tw - 1) / tw, tiles_y = (H + th - 1) / th;
    const uint32_t parts = std::max<uint32_t>(args.part_count, 1);
    std::vector<int32_t> tile_list;
    const RtTileMap map = rt_tile_map(tiles_x, tiles_y, parts, args.part0_weight);
    for (uint32_t ty = 0; ty < tiles_y; ++ty)
        for (uint32_t tx = 0; tx < tiles_x; ++tx)
            if (rt_tile_map_owner(map, tx, ty) == args.part_index) tile_list.push_back((int32_t)(ty * tiles_x + tx));
    const uint32_t n_slots = (uint32_t)tile_list.size();
    const uint32_t cap = n_slots * tw * th * spp;
    const uint32_t npl = (uint32_t)S.n_point_lights;

    std::vector<std::vector<char>> pool;
    RtBuffers Q{};
    for (int w = 0; w < 2; ++w) {
        for (int k = 0; k < 3; ++k) {
            Q.q_o[w][k] = grab<double>(pool, cap);
            Q.q_d[w][k] = grab<double>(pool, cap);
        }
        Q.q_sample[w] = grab<int32_t>(pool, cap);
        Q.q_org_shape[w] = grab<int32_t>(pool, cap);
        Q.q_org_tri[w] = grab<int32_t>(pool, cap);
        Q.q_self_t[w] = grab<float>(pool, cap);
    }
    Q.h_t = grab<float>(pool, cap);
    Q.h_shape = grab<int32_t>(pool, cap);
    Q.h_tri = grab<int32_t>(pool, cap);
    for (int k = 0; k < 3; ++k) {
        Q.l_p[k] = grab<double>(pool, cap);
        Q.l_n[k] = grab<double>(pool, cap);
        Q.l_d[k] = grab<double>(pool, cap);
    }
    Q.l_shape = grab<int32_t>(pool, cap);
    Q.l_tri = grab<int32_t>(pool, cap);
    Q.occ_shape = grab<int32_t>(pool, (size_t)cap * std::max(npl, 1u));
    Q.occ_tri = grab<int32_t>(pool, (size_t)cap * std::max(npl, 1u));
    Q.occ_t = grab<float>(pool, (size_t)cap * std::max(npl, 1u));
    Q.color = grab<float>(pool, (size_t)cap * 3);
    Q.l_self = grab<uint32_t>(pool, cap);
    Q.counters = grab<uint32_t>(pool, RT_CNT_SIZE);
    uint32_t overflow_word = 0;
    Q.overflow = &overflow_word;
    Q.block_counts = grab<uint32_t>(pool, 1);
    Q.lit_list = grab<uint32_t>(pool, 1);
    std::memset(Q.counters, 0, RT_CNT_SIZE * sizeof(uint32_t));
    Q.capacity = cap;
    Q.lit_f32 = args.mode == RT_MODE_REFERENCE_ORDER ? 0u : 1u;  // as rt::render_frame sets it

    RtBatch B{};
    B.tile_list = tile_list.data();
    B.slot0 = 0;
    B.n_pixels = n_slots * tw * th;
    B.tile_w = tw; B.tile_h = th; B.tiles_x = tiles_x;
    B.width = W; B.height = H;
    B.n = n;
    B.compact_fb = 0;  // the simulator always writes the full-image layout
    B.aa_method = args.antialias_method;
    B.seed = args.seed;
    B.light_sums_f32 = args.mode == RT_MODE_REFERENCE_ORDER ? 0u : 1u;  // as rt::render_frame sets it
    if (!(args.flags & RT_RENDER_ROW_MAJOR_PIXELS)) rt_pixel_block(tw, th, spp, B.blk_w_log2, B.blk_h_log2);
    rt_batch_finish(B);

    RtIdOut io{};
    if (ids) {
        io.primary_shape = ids->primary_shape; io.primary_tri = ids->primary_tri;
        io.final_shape = ids->final_shape; io.final_tri = ids->final_tri; io.bounces = ids->bounces;
        io.shadow_shape = ids->shadow_shape; io.shadow_tri = ids->shadow_tri;
        const size_t total = (size_t)W * H * spp * npl;
        for (size_t i = 0; i < total; ++i) {
            if (io.shadow_shape) io.shadow_shape[i] = -2;
            if (io.shadow_tri) io.shadow_tri[i] = -2;
        }
    }
    auto alloc_slot = [&](int counter) {
        uint32_t slot;
#pragma omp atomic capture
        slot = Q.counters[counter]++;
        return slot;
    };

    const long n_samples = (long)B.n_pixels * spp;
    if (hs.has_ggx) {  // the ray-tree path (rt_tree.h), as the backend takes it
        unsigned long long secondary = 0, shadow = 0, shaded = 0, primary = 0;
#pragma omp parallel reduction(+ : secondary, shadow, shaded, primary)
        {
            VecStack stk;
            stk.v.assign((size_t)std::max(S.stack_depth, 1), 0);
            TreeCounts counts = {0ull, 0ull, 0ull};
#pragma omp for schedule(dynamic, 256)
            for (long ls = 0; ls < n_samples; ++ls) {
                uint32_t i, j;
                primary += batch_pixel(B, (uint32_t)(ls / spp), i, j) ? 1 : 0;
                tree_body<VecStack, !EXACT>(S, Q, B, io, (uint32_t)ls, stk, counts);
            }
            secondary += counts.secondary; shadow += counts.shadow; shaded += counts.shaded;
        }
#pragma omp parallel for schedule(static, 4096)
        for (long lp = 0; lp < (long)B.n_pixels; ++lp) resolve_body(Q, B, (uint32_t)lp, fb);
        if (stats) {
            std::memset(stats, 0, sizeof(*stats));
            stats->primary_rays = primary;
            stats->secondary_rays = secondary;
            stats->shadow_rays = shadow;
            stats->shaded_hits = shaded;
            stats->kernel_launches[0] = overflow_word;
        }
        return;
    }
    const int max_bounce = hs.has_mirror ? S.recursion_depth : 0;
    for (int bounce = 0; bounce <= max_bounce; ++bounce) {
        const long count = bounce == 0 ? n_samples : (long)Q.counters[RT_CNT_RAYS0 + bounce];
#pragma omp parallel
        {
            VecStack stk;
            stk.v.assign((size_t)std::max(S.stack_depth, 1), 0);
#pragma omp for schedule(dynamic, 1024)
            for (long idx = 0; idx < count; ++idx) {
                const Tally before = g_tally;
                if (EXACT) trace_closest_body<VecStack, ORDERED, EXACT>(S, Q, B, bounce, (uint32_t)idx, stk);
                else trace_closest_fast_body<VecStack>(S, Q, B, bounce, (uint32_t)idx, stk);
                if (bounce == 0 && g_work_nodes[0]) {
                    g_work_nodes[0][idx] = (uint16_t)(g_tally.node - before.node);
                    g_work_tris[0][idx] = (uint16_t)(g_tally.tri - before.tri);
                }
            }
#pragma omp for schedule(dynamic, 1024)
            for (long idx = 0; idx < count; ++idx) {
                MirrorTmp tmp;
                const int act = shade_classify(S, Q, B, io, bounce, (uint32_t)idx, tmp);
                if (act == RT_ACT_MIRROR) shade_commit_mirror(Q, bounce, alloc_slot(RT_CNT_RAYS0 + bounce + 1), tmp);
            }
        }
    }
    uint64_t primary = 0;
    for (long ls = 0; ls < n_samples; ++ls) {
        uint32_t i, j;
        primary += batch_pixel(B, (uint32_t)(ls / spp), i, j) ? 1 : 0;
    }
    long lit = 0;
    if (hs.has_lit) {
#pragma omp parallel
        {
            VecStack stk;
            stk.v.assign((size_t)std::max(S.stack_depth, 1), 0);
            if (!S.disable_shadows)
                for (uint32_t l = 0; l < npl; ++l) {
#pragma omp for schedule(dynamic, 1024)
                    for (long e = 0; e < n_samples; ++e) {
                        const Tally before = g_tally;
                        if (EXACT) trace_shadow_body<VecStack, ORDERED, ANYHIT, EXACT>(S, Q, l, (uint32_t)e, stk);
                        else trace_shadow_fast_body<VecStack>(S, Q, l, (uint32_t)e, stk);
                        if (g_work_nodes[1]) {
                            g_work_nodes[1][(size_t)l * n_samples + e] = (uint16_t)(g_tally.node - before.node);
                            g_work_tris[1][(size_t)l * n_samples + e] = (uint16_t)(g_tally.tri - before.tri);
                        }
                    }
                }
#pragma omp for schedule(dynamic, 1024) reduction(+ : lit)
            for (long e = 0; e < n_samples; ++e) lit += accumulate_body(S, Q, B, io, (uint32_t)e) ? 1 : 0;
        }
    }
#pragma omp parallel for schedule(static, 4096)
    for (long lp = 0; lp < (long)B.n_pixels; ++lp) resolve_body(Q, B, (uint32_t)lp, fb);

    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->primary_rays = primary;
        for (int b = 1; b <= max_bounce; ++b) stats->secondary_rays += Q.counters[RT_CNT_RAYS0 + b];
        stats->shaded_hits = (uint64_t)lit;
        if (!S.disable_shadows) stats->shadow_rays = (uint64_t)lit * npl;
        stats->kernel_launches[0] = overflow_word;  // reused: nonzero = stack overflow
    }
}

}  // namespace

// Nodes fetched / triangles tested / top-level primitives visited since the last call, over all threads.
static unsigned long long g_total[3];
static void tally_flush() {
#pragma omp parallel
    {
#pragma omp critical
        {
            g_total[0] += g_tally.node; g_total[1] += g_tally.tri; g_total[2] += g_tally.tprim;
            g_tally = Tally();
        }
    }
}
extern "C" void hostsim_tally(unsigned long long out[3]) {
    tally_flush();
    for (int k = 0; k < 3; ++k) { out[k] = g_total[k]; g_total[k] = 0; }
}

extern "C" void hostsim_upload_options(uint32_t upload_flags) { g_upload_flags = upload_flags; }

extern "C" int hostsim_render(const RtScene *scene, const RtRenderArgs *args, float *fb, const RtRayIds *ids,
                              RtStats *stats) {
    HostScene hs;
    if (!make_view(*scene, hs).empty()) return -1;
    if (args->mode == RT_MODE_REFERENCE_ORDER) run<false, false, true>(hs, *scene, *args, fb, ids, stats);
    else run<true, true, false>(hs, *scene, *args, fb, ids, stats);
    return 0;
}

// org_shape / org_tri (may be null): the primitive each ray starts on; org_tri is a
// triangle index in the mesh's FILE order, as the id outputs report it.
extern "C" int hostsim_trace_rays_from(const RtScene *scene, uint64_t n, const double *o, const double *d,
                                       const double *tmin, const double *tmax, int32_t mode,
                                       const int32_t *org_shape, const int32_t *org_tri, float *t_out,
                                       int32_t *shape_out, int32_t *tri_out);

extern "C" int hostsim_trace_rays(const RtScene *scene, uint64_t n, const double *o, const double *d,
                                  const double *tmin, const double *tmax, int32_t mode, float *t_out,
                                  int32_t *shape_out, int32_t *tri_out) {
    return hostsim_trace_rays_from(scene, n, o, d, tmin, tmax, mode, nullptr, nullptr, t_out, shape_out, tri_out);
}

extern "C" int hostsim_trace_rays_from(const RtScene *scene, uint64_t n, const double *o, const double *d,
                                       const double *tmin, const double *tmax, int32_t mode,
                                       const int32_t *org_shape, const int32_t *org_tri, float *t_out,
                                       int32_t *shape_out, int32_t *tri_out) {
    HostScene hs;
    if (!make_view(*scene, hs).empty()) return -1;
    VecStack stk;
    stk.v.assign((size_t)std::max(std::max(hs.view.stack_depth, hs.view_fast.stack_depth), 1), 0);
    for (uint64_t r = 0; r < n; ++r) {
        HitF hit;
        const D3 ro{o[r * 3], o[r * 3 + 1], o[r * 3 + 2]}, rd{d[r * 3], d[r * 3 + 1], d[r * 3 + 2]};
        bool ok;
        int32_t os = org_shape ? org_shape[r] : -1, ot = -1;
        if (os >= 0 && org_tri && org_tri[r] >= 0) {
            // file-order index -> leaf-order index of that shape's mesh (in the copy the mode traverses)
            const RtShape &sh = scene->shapes[os];
            const int32_t mesh = sh.type == RT_SHAPE_INSTANCE ? scene->protos[sh.proto].mesh : sh.mesh;
            const uint64_t base = mode == RT_MODE_REFERENCE_ORDER ? hs.flat.mesh_tri_base[(size_t)mesh] : hs.flat.mesh_tri_base_fast[(size_t)mesh];
            const bool own_copy = mode != RT_MODE_REFERENCE_ORDER && !hs.flat.blas_fast[(size_t)mesh].nodes.empty();
            const uint64_t count = own_copy ? hs.flat.blas_fast[(size_t)mesh].order.size() : scene->meshes[mesh].n_triangles;
            for (uint64_t k = 0; k < count; ++k)
                if (hs.flat.tris[(size_t)(base + k)].orig == org_tri[r]) ot = (int32_t)(base + k);
        }
        hit.t = (float)tmax[r];
        hit.shape = -1;
        hit.tri = -1;
        double st;
        if (self_hit_f64(hs.view, os, ro, rd, tmin[r], tmax[r], false, false, st)) {  // direction unknown here: always test
            hit.t = (float)st;
            hit.shape = os;
        }
        RayF ray;
        if (mode == RT_MODE_REFERENCE_ORDER) ray_from_f64<true>(ro, rd, ray);
        else ray_from_f64<false>(ro, rd, ray);
        if (mode == RT_MODE_REFERENCE_ORDER)
            ok = traverse<VecStack, false, false, true>(hs.view, ray, (float)tmin[r], hit, os, ot, stk, hs.view.stack_depth);
        else {
            FastTrav st;
            fast_begin(hs.view_fast, st, stk, ro, rd, hit.t, hit.shape, os, ot);
            bool overflow = false;
            while (fast_step<VecStack, false>(hs.view_fast, st, stk, (float)tmin[r], hs.view_fast.stack_depth, overflow)) {
            }
            ok = !overflow;
            hit.t = st.t;
            hit.shape = st.shape;
            hit.tri = st.tri;
        }
        if (!ok) return -2;
        t_out[r] = hit.t;
        shape_out[r] = hit.shape;
        tri_out[r] = tri_file_index(hs.view, hit.tri);
    }
    return 0;
}

// Structure of the flattened trees, for comparison with the oracle's own build:
// which = -1 scene-level (reference order), -2 the fast mode's scene-level tree, -3 - m the fast
// mode's tree of mesh m (no nodes if that mesh keeps the reference's), else mesh index.
// Same conventions as oracle_bvh_dump.
extern "C" int64_t hostsim_bvh_dump(const RtScene *scene, int32_t which, int64_t cap_nodes, int64_t cap_prims,
                                    int32_t *kinds_out, int32_t *prims_out, double *boxes_out, int64_t *n_prims_out,
                                    int32_t *stack_depth_out) {
    rt::FlatScene flat;
    if (!rt::flatten_scene(*scene, flat).empty()) return -2;
    const rt::Tree &t = which <= -3 ? flat.blas_fast[(size_t)(-3 - which)]
                                     : (which == -2 ? flat.tlas_fast : (which < 0 ? flat.tlas : flat.blas[(size_t)which]));
    if ((int64_t)t.nodes.size() > cap_nodes || (int64_t)t.order.size() > cap_prims) return -1;
    for (size_t i = 0; i < t.nodes.size(); ++i) {
        kinds_out[i] = t.nodes[i].count ? (int32_t)t.nodes[i].count : -1;
        std::memcpy(boxes_out + i * 6, t.nodes[i].bmin, 3 * sizeof(double));
        std::memcpy(boxes_out + i * 6 + 3, t.nodes[i].bmax, 3 * sizeof(double));
    }
    std::memcpy(prims_out, t.order.data(), t.order.size() * sizeof(int32_t));
    *n_prims_out = (int64_t)t.order.size();
    if (stack_depth_out) *stack_depth_out = which <= -2 ? flat.stack_depth_fast : flat.stack_depth;
    return (int64_t)t.nodes.size();
}

// For every group of 64 consecutive rays (camera rays if light < 0, else the shadow rays of that
// point light in lit order): the number of distinct nodes / triangles the group touches, against
// the per-ray maxima.  out = {waves, sum union nodes, sum max-lane nodes, sum mean-lane nodes,
// sum union tris, sum max-lane tris, sum mean-lane tris}.
extern "C" int hostsim_wave_union(const RtScene *scene, const RtRenderArgs *args, int light, double *out) {
    HostScene hs;
    if (!make_view(*scene, hs).empty()) return -1;
    const RtSceneView &S = hs.view_fast;
    const uint32_t n = args->sqrt_rays_per_pixel, spp = n * n;
    const uint32_t W = scene->image_width, H = scene->image_height;
    const uint32_t tw = args->tile_w ? args->tile_w : 32, th = args->tile_h ? args->tile_h : 32;
    const uint32_t tiles_x = (W + tw - 1) / tw, tiles_y = (H + th - 1) / th;
    std::vector<int32_t> tile_list;
    for (uint32_t k = 0; k < tiles_x * tiles_y; ++k) tile_list.push_back((int32_t)k);
    RtBatch B{};
    B.tile_list = tile_list.data();
    B.n_pixels = (uint32_t)tile_list.size() * tw * th;
    B.tile_w = tw; B.tile_h = th; B.tiles_x = tiles_x; B.width = W; B.height = H; B.n = n;
    if (!(args->flags & RT_RENDER_ROW_MAJOR_PIXELS)) rt_pixel_block(tw, th, spp, B.blk_w_log2, B.blk_h_log2);
    rt_batch_finish(B);
    const long n_samples = (long)B.n_pixels * spp;
    // rays: camera rays, or shadow rays from the camera rays' hit points
    struct R { D3 o, d; float tmin, tmax; int32_t os, ot; bool anyhit; uint32_t i, j; };
    std::vector<R> rays;
    VecStack stk; stk.v.assign((size_t)S.stack_depth + 2, 0);
    for (long ls = 0; ls < n_samples; ++ls) {
        uint32_t i, j;
        if (!batch_pixel(B, (uint32_t)(ls / spp), i, j)) continue;
        const uint32_t pq = (uint32_t)(ls % spp);
        D3 o, d;
        camera_ray(S.camera, i, j, ((double)(pq / n) + 0.5) / (double)n, ((double)(pq % n) + 0.5) / (double)n, o, d);
        if (light < 0) { rays.push_back(R{o, d, 1.0f, INFINITY, -1, -1, false, i, j}); continue; }
        FastTrav st; bool overflow = false;
        fast_begin(S, st, stk, o, d, INFINITY, -1, -1, -1);
        while (fast_step<VecStack, false>(S, st, stk, 1.0f, S.stack_depth, overflow)) {}
        if (st.shape < 0) continue;
        const int32_t st_ = S.shaders[S.shapes[st.shape].shader].type;
        if (st_ != RT_SHADER_LAMBERTIAN && st_ != RT_SHADER_BLINN_PHONG) continue;
        const D3 p = o + d * (double)st.t;
        const D3 lp = d3(S.lights[S.point_lights[light]].position);
        rays.push_back(R{p, lp - p, 1e-6f, 1.0f, st.shape, st.tri, true, i, j});
    }
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<int32_t> vn, vt, un, ut;
    // level of every node inside its own tree (the header node of a tree is level 0)
    std::vector<uint8_t> level(hs.flat.nodes.size(), 255);
    {
        std::vector<int32_t> roots{S.tlas_root};
        for (size_t k = 0; k < hs.flat.tprims.size(); ++k)
            if (hs.flat.tprims[k].type == RT_SHAPE_MESH ||
                (hs.flat.tprims[k].type == RT_SHAPE_INSTANCE && hs.flat.tprims[k].proto_type == RT_SHAPE_MESH))
                roots.push_back(hs.flat.tprims[k].blas_root);
        std::vector<std::pair<int32_t, int>> todo;
        for (int32_t r : roots) {
            todo.emplace_back(r, 0);
            while (!todo.empty()) {
                const auto [ref, lv] = todo.back();
                todo.pop_back();
                if (ref < 0) continue;
                const size_t k = (size_t)ref / sizeof(RtNode);
                if (level[k] != 255) continue;
                level[k] = (uint8_t)std::min(lv, 254);
                todo.emplace_back(hs.flat.nodes[k].left, lv + 1);
                todo.emplace_back(hs.flat.nodes[k].right, lv + 1);
            }
        }
    }
    // RT_WAVE_DUMP=<file>: one record of six uint32 per wave (union nodes, longest lane's nodes, union
    // triangles, longest lane's triangles, pixel of the first ray)
    FILE *dump = std::getenv("RT_WAVE_DUMP") ? std::fopen(std::getenv("RT_WAVE_DUMP"), "wb") : nullptr;
    for (size_t w0 = 0; w0 + 64 <= rays.size(); w0 += 64) {
        un.clear(); ut.clear();
        size_t maxn = 0, maxt = 0, sumn = 0, sumt = 0;
        for (size_t l = 0; l < 64; ++l) {
            const R &r = rays[w0 + l];
            vn.clear(); vt.clear();
            g_visit[0] = &vn; g_visit[1] = &vt;
            FastTrav st; bool overflow = false;
            fast_begin(S, st, stk, r.o, r.d, r.tmax, -1, r.os, r.ot);
            if (r.anyhit) { while (fast_step<VecStack, true>(S, st, stk, r.tmin, S.stack_depth, overflow)) {} }
            else { while (fast_step<VecStack, false>(S, st, stk, r.tmin, S.stack_depth, overflow)) {} }
            g_visit[0] = g_visit[1] = nullptr;
            maxn = std::max(maxn, vn.size()); maxt = std::max(maxt, vt.size());
            sumn += vn.size(); sumt += vt.size();
            un.insert(un.end(), vn.begin(), vn.end()); ut.insert(ut.end(), vt.begin(), vt.end());
        }
        // nodes of different instances of one mesh share indices; that is what a packet would share too
        std::sort(un.begin(), un.end()); un.erase(std::unique(un.begin(), un.end()), un.end());
        std::sort(ut.begin(), ut.end()); ut.erase(std::unique(ut.begin(), ut.end()), ut.end());
        acc[0] += 1; acc[1] += (double)un.size(); acc[2] += (double)maxn; acc[3] += (double)sumn / 64.0;
        acc[4] += (double)ut.size(); acc[5] += (double)maxt; acc[6] += (double)sumt / 64.0;
        for (int32_t ref : un) {
            const uint8_t lv = level[(size_t)ref / sizeof(RtNode)];
            acc[7] += lv < 6 ? 1.0 : 0.0;
            acc[8] += lv < 8 ? 1.0 : 0.0;
        }
        if (dump) {
            const uint32_t rec[6] = {(uint32_t)un.size(), (uint32_t)maxn, (uint32_t)ut.size(), (uint32_t)maxt, rays[w0].i, rays[w0].j};
            std::fwrite(rec, sizeof(rec), 1, dump);
        }
    }
    if (dump) std::fclose(dump);
    for (int k = 0; k < 9; ++k) out[k] = acc[k];
    return 0;
}
