// rt_tree.h -- one thread evaluates the whole ray tree of one sample, depth first.
//
// The staged pipeline (rt_pipeline.h) carries ONE ray per sample from bounce to bounce, which is
// all that the Lambertian / Blinn-Phong / perfect-mirror shaders need.  GGXMirrorShader
// (shader/ggx_mirror.rs:12-94) fans out: every hit spawns `samples` reflected rays, each of which
// is shaded recursively, and their colours are summed in order and divided by `samples`.  Scenes
// that use it are rendered through this file instead: the reference's recursion, unrolled into a
// loop with an explicit stack of the GGX hits that still wait for children.  It is the slow path
// by construction (the rays of neighbouring lanes part ways at the first rough bounce); its job
// is to produce the reference's image, sample by sample.
//
// Random numbers.  ggx_mirror.rs:33-34 draws two uniform numbers per reflected ray from
// rand::thread_rng(), which the operating system seeds: no two reference runs agree, and the order
// in which its recursion draws cannot be observed.  Here a ray's pair is keyed by the ray's PATH in
// its sample's tree: the camera ray has path 0; the k-th ray (k = 0, 1, ...) that a GGX hit spawns,
// found by a ray of path P, has path (P << 8) + k + 1 (64 bits, wrapping: beyond eight rough bounces
// the oldest drop out); a perfect mirror passes its path on.  The pair is Philox4x32-10(counter =
// (sample index low, high, path low, path high), key = seed): u1 from output words 0 and 1, u2 from
// words 2 and 3, 53 bits each.  (A path is never 0, the sample offsets of antialias.rs use counter
// words (.., 0, 0): the two streams never meet.)  A ray's numbers thus do not depend on what any other
// ray of the tree hit, and the tree may be traced depth first (this file) or level by level
// (the wavefront path of rt_backend.hip) with the same result.
#pragma once

#include <math.h>

#include "rt_pipeline.h"

#define RT_TREE_MAX_DEPTH 16  // GGX hits that can wait for children at once = recursion depth; deeper scenes are rejected

// math/coordinate_system.rs:53-62
RT_HD void create_coordinate_system(const D3 &normal, D3 &tangent, D3 &bitangent) {
    const D3 t0 = __builtin_fabs(normal.x) > 0.99 ? D3{0.0, 1.0, 0.0} : D3{1.0, 0.0, 0.0};
    bitangent = normalize(cross(normal, t0));
    tangent = normalize(cross(bitangent, normal));
}

// GGXMirrorShader::sample_ggx, ggx_mirror.rs:29-51
RT_HD uint64_t tree_child_path(uint64_t path, uint32_t k) { return (path << 8) + (uint64_t)k + 1ull; }

// The reflected ray number k of a GGX hit (ggx_mirror.rs:66-76): direction, normalised.
RT_HD D3 ggx_child_direction(const D3 &n, const D3 &incoming, double alpha, uint64_t sample_index, uint64_t child_path, uint64_t seed);

// (theta = atan(r) with r = sqrt(alpha^2 u2 / (1 - u2)) >= 0 is only ever used through its sine and cosine, which are
// r / sqrt(1 + r^2) and 1 / sqrt(1 + r^2): the same numbers to a unit in the last place without the arc tangent and
// two of the four trigonometric calls, which were most of what spawning a ray cost.)
RT_HD D3 sample_ggx(const D3 &normal, double alpha, double u1, double u2) {
    const double pi = 3.14159265358979323846;
    const double phi = 2.0 * pi * u1;
    const double r = sqrt(alpha * alpha * u2 / (1.0 - u2));
    const double cos_theta = 1.0 / sqrt(1.0 + r * r), sin_theta = r * cos_theta;
    double sin_phi, cos_phi;
    sincos(phi, &sin_phi, &cos_phi);  // (sincospi(2 u1) on the device, whose argument reduction is exact and short: no faster)
    const double x = sin_theta * cos_phi, y = sin_theta * sin_phi, z = cos_theta;
    D3 tangent, bitangent;
    create_coordinate_system(normal, tangent, bitangent);
    return tangent * x + bitangent * y + normal * z;
}

RT_HD D3 ggx_child_direction(const D3 &n, const D3 &incoming, double alpha, uint64_t sample_index, uint64_t child_path, uint64_t seed) {
    uint32_t x[4];
    rt_philox4x32_10((uint32_t)sample_index, (uint32_t)(sample_index >> 32), (uint32_t)child_path, (uint32_t)(child_path >> 32),
                     (uint32_t)seed, (uint32_t)(seed >> 32), x);
    const D3 m = sample_ggx(n, alpha, rt_unit_double(x[0], x[1]), rt_unit_double(x[2], x[3]));
    const D3 outgoing = incoming - m * (2.0 * dot(incoming, m));  // ggx_mirror.rs:69
    return normalize(outgoing);                                   // ggx_mirror.rs:74
}

struct TreeCounts {
    unsigned long long secondary, shadow, shaded;
};

// One ray against the scene.  FAST: the fast mode's traversal (any-hit for shadow rays);
// otherwise the reference-order traversal, closest hit always.  `self_t` / org_*: the re-hit of
// the primitive the ray starts on, decided in f64 by the caller, and that primitive (skipped here).
template <class Stack, bool FAST>
RT_HD bool tree_trace(const RtSceneView &S, Stack &stk, const D3 &o, const D3 &d, float tmin, float tmax, bool anyhit,
                      float self_t, int32_t org_shape, int32_t org_tri, float &t, int32_t &shape, int32_t &tri) {
    const bool own = self_t < tmax;
    const float t0 = own ? self_t : tmax;
    const int32_t shape0 = own ? org_shape : -1;
    if (FAST) {
        FastTrav st;
        fast_begin(S, st, stk, o, d, t0, shape0, org_shape, org_tri);
        bool overflow = false;
        if (anyhit) {
            if (shape0 < 0)
                while (fast_step<Stack, true>(S, st, stk, tmin, S.stack_depth, overflow)) {
                }
        } else {
            while (fast_step<Stack, false>(S, st, stk, tmin, S.stack_depth, overflow)) {
            }
        }
        t = st.t; shape = st.shape; tri = st.tri;
        return !overflow;
    }
    RayF ray;
    HitF hit;
    ray_from_f64<true>(o, d, ray);
    hit.t = t0;
    hit.shape = shape0;
    hit.tri = -1;
    const bool ok = traverse<Stack, false, false, true>(S, ray, tmin, hit, org_shape, org_tri, stk, S.stack_depth);
    t = hit.t; shape = hit.shape; tri = hit.tri;
    return ok;
}

// lambertian.rs:17-36 / blinn_phong.rs:23-47 with light/point.rs:30-40 inline: every point light's
// shadow ray is traced here, then the visible lights are summed in scene order.
template <class Stack, bool FAST>
RT_HD ColorF tree_shade_lights(const RtSceneView &S, Stack &stk, const RtShader &sh, const D3 &p, const D3 &n, const D3 &ray_dir,
                               int32_t shape, int32_t gtri, TreeCounts &counts, bool &ok) {
    ColorF color = {0.0f, 0.0f, 0.0f};
    counts.shaded += 1;
    for (int li = 0; li < S.n_lights; ++li) {
        const RtLight &light = S.lights[li];
        if (light.type == RT_LIGHT_AMBIENT) {  // ambient.rs:25-27
            add_light<false>(sh, light, n, n, ray_dir, color);
            continue;
        }
        const D3 dir = d3(light.position) - p;  // Ray::atob, ray.rs:15-20
        bool visible = true;
        if (!S.disable_shadows) {
            counts.shadow += 1;
            double st = 0.0;
            const bool self = self_hit_f64(S, shape, p, dir, RT_VERY_SMALL_NUMBER, 1.0, dot(n, dir) > 0.0, false, st);
            float t;
            int32_t bs, bt;
            ok = tree_trace<Stack, FAST>(S, stk, p, dir, (float)RT_VERY_SMALL_NUMBER, 1.0f, true,
                                         self ? (float)st : __builtin_inff(), shape, gtri, t, bs, bt) && ok;
            visible = bs < 0;
        }
        if (visible) add_light<false>(sh, light, n, dir, ray_dir, color);
    }
    return color;
}

struct TreeFrame {  // a GGX hit waiting for its reflected rays (ggx_mirror.rs:66-93)
    D3 p, n, incoming;
    ColorF acc;
    uint32_t k, samples;
    double alpha;
    uint64_t path;  // of the ray that found this hit
    int32_t shape, tri, depth;
};

// render.rs:64-74 for one sample, as a walk that can be taken one ray at a time: begin() takes the camera ray, every
// step() traces ONE ray of the sample's tree (and the shadow rays of the hit it finds, if that hit is lit) and returns
// true once the colour that render_pixel adds to the pixel's sum is known.  The order of the rays, of the random draws
// and of every sum is the order of the reference's recursion, whoever calls step() and however other samples' steps
// are interleaved with it; `path` is the current ray's path in the tree and `sample_index` the frame-wide sample index
// (together the generator's counter).
// (The stack of waiting GGX hits is the caller's array, not a member: an array that is indexed at run time lives in
// scratch memory, and as a member it would take the whole walk -- current ray, depth, counters -- there with it.)
struct TreeWalk {
    D3 o, d;
    uint64_t sample_index;
    int32_t top, depth, org_shape, org_tri;
    uint64_t path;
    float tmin, self_t;
    int32_t primary_shape, primary_tri;

    RT_HD void begin(const D3 &origin, const D3 &direction, uint64_t sample) {
        o = origin;
        d = direction;
        sample_index = sample;
        top = -1;
        path = 0;
        depth = 0;
        org_shape = org_tri = -1;
        tmin = 1.0f;
        self_t = __builtin_inff();
        primary_shape = primary_tri = -1;
    }

    template <class Stack, bool FAST>
    RT_HD bool step(const RtSceneView &S, Stack &stk, TreeFrame *frames, uint64_t seed, TreeCounts &counts, bool &ok, ColorF &out) {
        const ColorF background = {S.background[0], S.background[1], S.background[2]};
        ColorF ret = background;
        // ---- trace the current ray and shade its hit: ends in a colour (ret) or in another ray
        bool deliver = true;  // false: a GGX hit was pushed and `ret` holds nothing
        float tf;
        int32_t shape, gtri;
        ok = tree_trace<Stack, FAST>(S, stk, o, d, tmin, __builtin_inff(), false, self_t, org_shape, org_tri, tf, shape, gtri) && ok;
        if (depth == 0) {
            primary_shape = shape;
            primary_tri = gtri;
        }
        if (shape < 0) {
            ret = background;  // render.rs:73 / perfect_mirror.rs:30 / ggx_mirror.rs:86
        } else {
            double t = (double)tf;
            D3 n;
            exact_prim_hit(S, shape, gtri, o, d, (double)tmin, (double)__builtin_inff(), true, t, t, &n);
            const D3 p = o + d * t;
            const RtShader &sh = S.shaders[S.shapes[shape].shader];
            if (sh.type == RT_SHADER_LAMBERTIAN || sh.type == RT_SHADER_BLINN_PHONG) {
                ret = tree_shade_lights<Stack, FAST>(S, stk, sh, p, n, d, shape, gtri, counts, ok);
            } else if (sh.type == RT_SHADER_NORMAL) {  // normal.rs:10-16
                ret = ColorF{(1.0f + (float)n.x) / 2.0f, (1.0f + (float)n.y) / 2.0f, (1.0f + (float)n.z) / 2.0f};
            } else if (sh.type == RT_SHADER_PERFECT_MIRROR || sh.type == RT_SHADER_GGX_MIRROR) {
                if (depth >= S.recursion_depth) {  // perfect_mirror.rs:10-12, ggx_mirror.rs:56-58
                    ret = background;
                } else if (sh.type == RT_SHADER_PERFECT_MIRROR) {  // a tail call: its colour is the child's
                    const D3 out_dir = mirror_direction(d, n);      // the mirror ray (perfect_mirror.rs:18-27)
                    o = p;
                    d = out_dir;
                    depth += 1;
                    tmin = (float)RT_VERY_SMALL_NUMBER;
                    org_shape = shape;
                    org_tri = gtri;
                    double st;
                    self_t = self_hit_f64(S, shape, o, d, RT_VERY_SMALL_NUMBER, (double)__builtin_inff(), dot(n, d) > 0.0, false, st)
                                 ? (float)st
                                 : __builtin_inff();
                    counts.secondary += 1;
                    return false;
                } else {
                    top += 1;
                    TreeFrame &f = frames[top];
                    f.p = p; f.n = n; f.incoming = normalize(d);
                    f.acc = ColorF{0.0f, 0.0f, 0.0f};
                    f.k = 0; f.samples = sh.samples; f.alpha = sh.roughness;
                    f.shape = shape; f.tri = gtri; f.depth = depth;
                    f.path = path;
                    deliver = false;  // nothing to hand over yet: go and spawn its first reflected ray
                }
            } else {  // null.rs:7-9
                ret = ColorF{1.0f, 0.0f, 1.0f};
            }
        }
        // ---- a colour came back: hand it to the GGX hit that waits for it, or finish
        for (;;) {
            if (deliver) {
                if (top < 0) {
                    out = ret;
                    return true;
                }
                TreeFrame &w = frames[top];
                w.acc.r += ret.r; w.acc.g += ret.g; w.acc.b += ret.b;  // ggx_mirror.rs:89
                w.k += 1;
            }
            deliver = true;
            TreeFrame &f = frames[top];
            if (f.k >= f.samples) {  // ggx_mirror.rs:93: accumulated_color / self.samples as f32
                const float div = (float)f.samples;
                ret = ColorF{f.acc.r / div, f.acc.g / div, f.acc.b / div};
                top -= 1;
                continue;
            }
            path = tree_child_path(f.path, f.k);
            o = f.p;
            d = ggx_child_direction(f.n, f.incoming, f.alpha, sample_index, path, seed);
            depth = f.depth + 1;
            tmin = (float)RT_VERY_SMALL_NUMBER;
            org_shape = f.shape;
            org_tri = f.tri;
            double st;
            self_t = self_hit_f64(S, f.shape, o, d, RT_VERY_SMALL_NUMBER, (double)__builtin_inff(), dot(f.n, d) > 0.0, false, st)
                         ? (float)st
                         : __builtin_inff();
            counts.secondary += 1;
            return false;
        }
    }
};

// Where a finished sample's colour and ids go (k_resolve sums the colour slots).  Of the per-sample ids only the camera
// ray's hit is defined for a tree.
RT_HD void tree_store(const RtSceneView &S, const RtBuffers &Q, const RtBatch &B, const RtIdOut &ids, uint32_t ls, const ColorF &c,
                      int32_t primary_shape, int32_t primary_tri) {
    write_color(Q, (int32_t)ls, c.r, c.g, c.b);
    if (ids.primary_shape || ids.primary_tri || ids.final_shape || ids.final_tri || ids.bounces) {
        const size_t gs = global_sample(B, ls);
        if (ids.primary_shape) ids.primary_shape[gs] = primary_shape;
        if (ids.primary_tri) ids.primary_tri[gs] = tri_file_index(S, primary_tri);
        if (ids.final_shape) ids.final_shape[gs] = -1;
        if (ids.final_tri) ids.final_tri[gs] = -1;
        if (ids.bounces) ids.bounces[gs] = 0;
    }
}

// One local sample of a batch, start to finish: camera ray, tree, colour into the sample's slot.
template <class Stack, bool FAST>
RT_HD void tree_body(const RtSceneView &S, const RtBuffers &Q, const RtBatch &B, const RtIdOut &ids, uint32_t ls, Stack &stk,
                     TreeCounts &counts) {
    D3 o, d;
    if (!sample_camera_ray(S, B, ls, o, d)) return;
    TreeFrame frames[RT_TREE_MAX_DEPTH];
    TreeWalk walk;
    walk.begin(o, d, frame_sample(B, ls));
    bool ok = true;
    ColorF c;
    while (!walk.template step<Stack, FAST>(S, stk, frames, B.seed, counts, ok, c)) {
    }
    if (!ok) *Q.overflow = 1;
    tree_store(S, Q, B, ids, ls, c, walk.primary_shape, walk.primary_tri);
}
