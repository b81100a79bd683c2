// rt_pipeline.h -- the per-thread bodies of the wavefront pipeline's stages.
//
// The reference renders one sample with a recursion
//   render_pixel -> bvh.closest_hit -> shader.apply -> { light.illuminates -> bvh.closest_hit }*
//                                                   -> { PerfectMirror: bvh.closest_hit -> shader.apply ... }
// (render.rs:53-84, shader/*.rs, light/point.rs:30-40).  A mirror adds no colour
// of its own and does not attenuate, so a sample is a CHAIN of closest-hit rays
// that ends in the background or in one diffuse / Blinn-Phong hit; that hit casts
// one shadow ray per point light.  Here the chain is unrolled into stages that
// each run over a whole queue of rays:
//
//   raygen            sample -> camera ray                       -> ray queue[0]
//   trace (closest)   ray queue[b] -> hit records
//   shade             hit records -> colour (background / normal / null shader)
//                                  | mirror ray                   -> ray queue[b+1]
//                                  | lit hit                      -> shade queue
//   (repeat trace + shade while mirror rays remain, at most recursion_depth times)
//   trace (shadow)    shade queue x point lights -> blocker ids
//   accumulate        shade queue -> colour, lights summed in scene order
//   resolve           per pixel: samples summed in p-major order, / n*n  -> framebuffer
//
// Every stage body below is written for ONE thread and one queue index; the HIP
// kernels wrap them in grid-stride loops and do the queue-slot allocation with a
// wave-wide ballot.  The bodies are plain C++ so a host test build can run them
// in a for-loop.
#pragma once

#include "rt_core.h"

// Device buffers of one batch of samples (structure of arrays).
struct RtBuffers {
    // ray queues, ping-pong by bounce parity
    double *q_o[2][3];
    double *q_d[2][3];
    int32_t *q_sample[2];     // local sample index (colour slot)
    int32_t *q_org_shape[2];  // primitive the ray starts on, -1 for camera rays
    int32_t *q_org_tri[2];
    // hit records of the queue being processed
    float *h_t;
    int32_t *h_shape;
    int32_t *h_tri;
    // hits waiting for their lights
    double *s_p[3];  // hit point
    double *s_n[3];  // unit normal
    double *s_d[3];  // direction of the ray that hit (un-normalised)
    int32_t *s_sample;
    int32_t *s_shape;
    int32_t *s_tri;
    // [point light slot * capacity + shade entry]: blocker, -1 = light visible
    int32_t *occ_shape;
    int32_t *occ_tri;
    // colour of each local sample, 3 floats
    float *color;
    // device counters, see RT_CNT_*
    uint32_t *counters;
    uint32_t capacity;  // entries every queue can hold
};

enum {
    RT_CNT_RAYS0 = 0,        // [0 .. 31]: rays in queue of bounce b
    RT_CNT_SHADE = 32,       // shade-queue entries
    RT_CNT_OVERFLOW = 33,    // traversal-stack overflows (must stay 0)
    RT_CNT_SIZE = 40
};

// Which pixels a batch covers.  Local pixel lp of the batch lies in tile slot
// slot0 + lp / (tile_w*tile_h); tile_list maps a slot to the row-major tile index.
struct RtBatch {
    const int32_t *tile_list;
    uint32_t slot0;       // first tile slot of this batch
    uint32_t n_pixels;    // local pixels (tile slots x tile_w x tile_h)
    uint32_t tile_w, tile_h, tiles_x;
    uint32_t width, height;
    uint32_t n;           // sqrt_rays_per_pixel
    uint32_t compact_fb;  // 1: framebuffer is tile-major over the owned tiles, 0: full image layout
};

struct RtIdOut {  // RtRayIds, device pointers (any may be null)
    int32_t *primary_shape, *primary_tri, *final_shape, *final_tri, *bounces, *shadow_shape, *shadow_tri;
};

RT_HD bool batch_pixel(const RtBatch &B, uint32_t lp, uint32_t &i, uint32_t &j) {
    const uint32_t tp = B.tile_w * B.tile_h;
    const uint32_t slot = B.slot0 + lp / tp, r = lp % tp;
    const uint32_t tile = (uint32_t)B.tile_list[slot];
    i = (tile % B.tiles_x) * B.tile_w + r % B.tile_w;
    j = (tile / B.tiles_x) * B.tile_h + r / B.tile_w;
    return i < B.width && j < B.height;
}

// Global sample index of RtRayIds for a local sample.
RT_HD size_t global_sample(const RtBatch &B, uint32_t ls) {
    const uint32_t spp = B.n * B.n;
    uint32_t i, j;
    batch_pixel(B, ls / spp, i, j);
    return ((size_t)i + (size_t)j * B.width) * spp + ls % spp;
}

RT_HD int32_t tri_file_index(const RtSceneView &S, int32_t gtri) {
    return gtri < 0 ? -1 : reinterpret_cast<const RtTri *>(S.tris)[gtri].orig;
}

// ---- raygen: render.rs:64-68
RT_HD bool raygen_classify(const RtSceneView &S, const RtBatch &B, uint32_t ls, D3 &o, D3 &d) {
    const uint32_t spp = B.n * B.n;
    uint32_t i, j;
    if (!batch_pixel(B, ls / spp, i, j)) return false;
    const uint32_t k = ls % spp;
    camera_ray(S.camera, i, j, k / B.n, k % B.n, B.n, o, d);
    return true;
}

RT_HD void queue_write_ray(const RtBuffers &Q, int which, uint32_t slot, const D3 &o, const D3 &d, int32_t sample,
                           int32_t org_shape, int32_t org_tri) {
    Q.q_o[which][0][slot] = o.x; Q.q_o[which][1][slot] = o.y; Q.q_o[which][2][slot] = o.z;
    Q.q_d[which][0][slot] = d.x; Q.q_d[which][1][slot] = d.y; Q.q_d[which][2][slot] = d.z;
    Q.q_sample[which][slot] = sample;
    Q.q_org_shape[which][slot] = org_shape;
    Q.q_org_tri[which][slot] = org_tri;
}

RT_HD void queue_read_ray(const RtBuffers &Q, int which, uint32_t slot, D3 &o, D3 &d) {
    o = D3{Q.q_o[which][0][slot], Q.q_o[which][1][slot], Q.q_o[which][2][slot]};
    d = D3{Q.q_d[which][0][slot], Q.q_d[which][1][slot], Q.q_d[which][2][slot]};
}

// ---- trace (closest): render.rs:70 / perfect_mirror.rs:27.
// Hit::new gives camera rays t_min = 1.0 (hit_struct.rs:20); mirror rays get
// VERY_SMALL_NUMBER (perfect_mirror.rs:25).
template <class Stack, bool ORDERED, bool EXACT>
RT_HD void trace_closest_body(const RtSceneView &S, const RtBuffers &Q, int bounce, uint32_t idx, Stack &stk) {
    const int which = bounce & 1;
    D3 o, d;
    queue_read_ray(Q, which, idx, o, d);
    const double tmin = bounce == 0 ? 1.0 : RT_VERY_SMALL_NUMBER;
    HitF hit;
    const bool ok = trace_ray<Stack, ORDERED, false, EXACT>(S, o, d, tmin, (double)__builtin_inff(),
                                                            Q.q_org_shape[which][idx], Q.q_org_tri[which][idx], stk,
                                                            S.stack_depth, hit);
    if (!ok) Q.counters[RT_CNT_OVERFLOW] = 1;
    Q.h_t[idx] = hit.t;
    Q.h_shape[idx] = hit.shape;
    Q.h_tri[idx] = hit.tri;
}

// ---- shade
enum { RT_ACT_NONE = 0, RT_ACT_MIRROR = 1, RT_ACT_LIGHTS = 2 };

struct ShadeTmp {
    D3 p, n, d;  // hit point, normal, direction of the incoming ray (MIRROR: of the outgoing ray)
    int32_t sample, shape, tri;
};

RT_HD void write_color(const RtBuffers &Q, int32_t sample, float r, float g, float b) {
    float *c = Q.color + (size_t)sample * 3;
    c[0] = r; c[1] = g; c[2] = b;
}

// Decides what happens to queue entry idx of bounce `bounce` (shader/*.rs apply()).
RT_HD int shade_classify(const RtSceneView &S, const RtBuffers &Q, const RtBatch &B, const RtIdOut &ids, int bounce,
                         uint32_t idx, ShadeTmp &tmp) {
    const int which = bounce & 1;
    const int32_t sample = Q.q_sample[which][idx];
    const int32_t shape = Q.h_shape[idx], gtri = Q.h_tri[idx];
    const bool want_ids = ids.primary_shape || ids.primary_tri || ids.final_shape || ids.final_tri || ids.bounces;
    const size_t gs = want_ids ? global_sample(B, (uint32_t)sample) : 0;
    const int32_t ftri = tri_file_index(S, gtri);
    if (bounce == 0) {
        if (ids.primary_shape) ids.primary_shape[gs] = shape;
        if (ids.primary_tri) ids.primary_tri[gs] = ftri;
    }
    if (ids.final_shape) ids.final_shape[gs] = shape;
    if (ids.final_tri) ids.final_tri[gs] = ftri;
    if (ids.bounces) ids.bounces[gs] = bounce;
    if (shape < 0) {  // render.rs:73 / perfect_mirror.rs:30
        write_color(Q, sample, S.background[0], S.background[1], S.background[2]);
        return RT_ACT_NONE;
    }
    D3 o, d;
    queue_read_ray(Q, which, idx, o, d);
    const double tmin = bounce == 0 ? 1.0 : RT_VERY_SMALL_NUMBER;
    double t = (double)Q.h_t[idx];
    D3 n;
    exact_prim_hit(S, shape, gtri, o, d, tmin, (double)__builtin_inff(), true, t, t, &n);
    const D3 p = o + d * t;  // Hit::hit_point, hit_struct.rs:41-43
    const RtShader &sh = S.shaders[S.shapes[shape].shader];
    tmp.sample = sample;
    tmp.shape = shape;
    tmp.tri = gtri;
    tmp.p = p;
    tmp.n = n;
    switch (sh.type) {
    case RT_SHADER_LAMBERTIAN:
    case RT_SHADER_BLINN_PHONG:
        tmp.d = d;
        return RT_ACT_LIGHTS;
    case RT_SHADER_PERFECT_MIRROR:  // perfect_mirror.rs:9-32
        if (bounce >= S.recursion_depth) {
            write_color(Q, sample, S.background[0], S.background[1], S.background[2]);
            return RT_ACT_NONE;
        }
        tmp.d = mirror_direction(d, n);
        return RT_ACT_MIRROR;
    case RT_SHADER_NORMAL:  // normal.rs:10-16
        write_color(Q, sample, (1.0f + (float)n.x) / 2.0f, (1.0f + (float)n.y) / 2.0f, (1.0f + (float)n.z) / 2.0f);
        return RT_ACT_NONE;
    default:  // null.rs:7-9
        write_color(Q, sample, 1.0f, 0.0f, 1.0f);
        return RT_ACT_NONE;
    }
}

RT_HD void shade_commit_mirror(const RtBuffers &Q, int bounce, uint32_t slot, const ShadeTmp &tmp) {
    queue_write_ray(Q, (bounce + 1) & 1, slot, tmp.p, tmp.d, tmp.sample, tmp.shape, tmp.tri);
}

RT_HD void shade_commit_lights(const RtBuffers &Q, uint32_t slot, const ShadeTmp &tmp) {
    Q.s_p[0][slot] = tmp.p.x; Q.s_p[1][slot] = tmp.p.y; Q.s_p[2][slot] = tmp.p.z;
    Q.s_n[0][slot] = tmp.n.x; Q.s_n[1][slot] = tmp.n.y; Q.s_n[2][slot] = tmp.n.z;
    Q.s_d[0][slot] = tmp.d.x; Q.s_d[1][slot] = tmp.d.y; Q.s_d[2][slot] = tmp.d.z;
    Q.s_sample[slot] = tmp.sample;
    Q.s_shape[slot] = tmp.shape;
    Q.s_tri[slot] = tmp.tri;
}

// ---- trace (shadow): light/point.rs:30-37.  Work item = (point light slot, shade entry).
template <class Stack, bool ORDERED, bool ANYHIT, bool EXACT>
RT_HD void trace_shadow_body(const RtSceneView &S, const RtBuffers &Q, uint32_t light_slot, uint32_t entry,
                             Stack &stk) {
    const RtLight &light = S.lights[S.point_lights[light_slot]];
    const D3 p = D3{Q.s_p[0][entry], Q.s_p[1][entry], Q.s_p[2][entry]};
    const D3 dir = d3(light.position) - p;  // Ray::atob, ray.rs:15-20
    HitF hit;
    // Hit::to_light: t = 1.0, t_min = VERY_SMALL_NUMBER (hit_struct.rs:29-39)
    const bool ok = trace_ray<Stack, ORDERED, ANYHIT, EXACT>(S, p, dir, RT_VERY_SMALL_NUMBER, 1.0, Q.s_shape[entry],
                                                             Q.s_tri[entry], stk, S.stack_depth, hit);
    if (!ok) Q.counters[RT_CNT_OVERFLOW] = 1;
    const size_t k = (size_t)light_slot * Q.capacity + entry;
    Q.occ_shape[k] = hit.shape;
    Q.occ_tri[k] = hit.tri;
}

// ---- accumulate: lambertian.rs:17-36 / blinn_phong.rs:23-47
RT_HD void accumulate_body(const RtSceneView &S, const RtBuffers &Q, const RtBatch &B, const RtIdOut &ids,
                           uint32_t entry) {
    const D3 p = D3{Q.s_p[0][entry], Q.s_p[1][entry], Q.s_p[2][entry]};
    const D3 n = D3{Q.s_n[0][entry], Q.s_n[1][entry], Q.s_n[2][entry]};
    const D3 d = D3{Q.s_d[0][entry], Q.s_d[1][entry], Q.s_d[2][entry]};
    const int32_t sample = Q.s_sample[entry];
    const RtShader &sh = S.shaders[S.shapes[Q.s_shape[entry]].shader];
    const bool want_ids = ids.shadow_shape || ids.shadow_tri;
    const size_t gs = want_ids ? global_sample(B, (uint32_t)sample) : 0;
    ColorF color = {0.0f, 0.0f, 0.0f};
    int slot = 0;
    for (int li = 0; li < S.n_lights; ++li) {
        const RtLight &light = S.lights[li];
        if (light.type == RT_LIGHT_AMBIENT) {  // ambient.rs:25-27
            add_light(sh, light, n, n, d, color);
            continue;
        }
        bool visible = true;
        if (!S.disable_shadows) {
            const size_t k = (size_t)slot * Q.capacity + entry;
            const int32_t blocker = Q.occ_shape[k];
            visible = blocker < 0;
            if (ids.shadow_shape) ids.shadow_shape[gs * (size_t)S.n_point_lights + slot] = blocker;
            if (ids.shadow_tri)
                ids.shadow_tri[gs * (size_t)S.n_point_lights + slot] = tri_file_index(S, Q.occ_tri[k]);
        }
        if (visible) add_light(sh, light, n, d3(light.position) - p, d, color);
        ++slot;
    }
    write_color(Q, sample, color.r, color.g, color.b);
}

// ---- resolve: render.rs:63-83, framebuffer.rs:21-31
RT_HD void resolve_body(const RtBuffers &Q, const RtBatch &B, uint32_t lp, float *fb) {
    uint32_t i, j;
    if (!batch_pixel(B, lp, i, j)) return;
    const uint32_t spp = B.n * B.n;
    const float *c = Q.color + (size_t)lp * spp * 3;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t k = 0; k < spp; ++k) {  // p-major, q-minor: k = p*n + q
        r += c[k * 3 + 0];
        g += c[k * 3 + 1];
        b += c[k * 3 + 2];
    }
    const float div = (float)(uint16_t)(B.n * B.n);
    float *px;
    if (B.compact_fb) {
        const uint32_t tp = B.tile_w * B.tile_h;
        px = fb + ((size_t)(B.slot0 + lp / tp) * tp + lp % tp) * 3;
    } else {
        px = fb + ((size_t)i + (size_t)j * B.width) * 3;
    }
    px[0] = r / div;
    px[1] = g / div;
    px[2] = b / div;
}
