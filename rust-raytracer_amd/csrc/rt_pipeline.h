// rt_pipeline.h -- the per-thread bodies of the wavefront pipeline's stages.
//
// The reference renders one sample with a recursion
//   render_pixel -> bvh.closest_hit -> shader.apply -> { light.illuminates -> bvh.closest_hit }*
//                                                   -> { PerfectMirror: bvh.closest_hit -> shader.apply ... }
// (render.rs:53-84, shader/*.rs, light/point.rs:30-40).  A mirror adds no colour
// of its own and does not attenuate, so a sample is a CHAIN of closest-hit rays
// that ends in the background or in one diffuse / Blinn-Phong hit; that hit casts
// one shadow ray per point light.  Here the chain is unrolled into stages that
// each run over many rays at once:
//
//   trace  (bounce 0)   sample -> camera ray -> closest hit          (hit record per sample)
//   shade  (bounce 0)   hit -> colour (background / normal / null shader)
//                           | mirror ray  -> compacted ray queue
//                           | lit hit     -> that sample's slot of the "lit" table
//   trace / shade (bounce b >= 1)   the same over the ray queue, while mirror rays remain
//   trace  (shadow)     lit table x point lights -> blocker ids
//   accumulate          lit table -> colour, lights summed in scene order
//   resolve             per pixel: samples summed in p-major order, / n*n  -> framebuffer
//
// Camera rays are never stored: trace and shade both rebuild them from the
// sample index.  A sample ends in at most one lit hit, so the lit table is
// indexed by sample and needs no compaction; neighbouring lanes stay neighbouring
// samples.  Only the mirror-ray queue is compacted (wave-wide ballot).
//
// Every stage body below is written for ONE thread; the HIP kernels wrap them.
// The bodies are plain C++ so a host test build can run them in a for-loop.
#pragma once

#include "rt_core.h"

// Device buffers of one batch of samples (structure of arrays).
// Accesses to the per-batch work buffers (written once by one stage, read once by a later one):
// on the device they carry the non-temporal hint, so that these hundreds of megabytes per frame
// pass by the L2 lines that hold the trees and triangles, which every wave keeps coming back to.
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_STREAM_ST(lvalue, value) __builtin_nontemporal_store((value), &(lvalue))
#define RT_STREAM_LD(lvalue) __builtin_nontemporal_load(&(lvalue))
#else
#define RT_STREAM_ST(lvalue, value) ((lvalue) = (value))
#define RT_STREAM_LD(lvalue) (lvalue)
#endif

struct RtBuffers {
    // mirror-ray queues, ping-pong by bounce parity: bounce b reads q[b & 1], writes q[(b + 1) & 1]
    double *q_o[2][3];
    double *q_d[2][3];
    int32_t *q_sample[2];     // local sample index (colour / lit-table slot)
    int32_t *q_org_shape[2];  // primitive the ray starts on
    int32_t *q_org_tri[2];
    float *q_self_t[2];       // f64 re-hit of the origin primitive, as fp32 t; +inf = none
    // hit records: indexed by sample at bounce 0, by queue slot afterwards
    float *h_t;
    int32_t *h_shape;
    int32_t *h_tri;
    // lit table, indexed by local sample; l_shape < 0 = this sample has no lit hit
    double *l_p[3];  // hit point
    double *l_n[3];  // unit normal
    double *l_d[3];  // direction of the ray that hit (un-normalised)
    int32_t *l_shape;
    int32_t *l_tri;
    uint32_t *l_self;  // bit k: the hit's own primitive blocks point light k (f64 test in the shade stage)
    // [point light slot * capacity + sample]: blocker, -1 = light visible
    int32_t *occ_shape;
    int32_t *occ_tri;
    float *occ_t;     // t of the origin primitive's re-hit, valid where occ_shape >= 0 after the shade stage
    // colour of each local sample, 3 floats
    float *color;
    // device counters, see RT_CNT_*
    uint32_t *counters;
    uint32_t *overflow;      // set to 1 by a traversal whose stack would have overflowed (must stay 0); never
                             // cleared by the device, so that it survives until the host looks
    uint32_t *block_counts;  // lit hits per chunk of RT_LIT_CHUNK samples (added up by the shade stage), then
                             // their exclusive prefix sums (not with RtBatch::direct_lit: they stay counts); lies
                             // behind the counters and is cleared with them
    uint32_t *mirror_counts; // RtBatch::chunked_queue: mirror rays per chunk of RT_LIT_CHUNK samples that the camera rays left in
                             // the chunk's stretch of queue 1 (slots from chunk * RT_LIT_CHUNK on); lies behind block_counts
    uint32_t *counters_next; // the other of the two sets of counters and chunk counts: batches take turns with them,
                             // and a batch's first kernel zeroes the set of the batch behind it where it can
    uint32_t *lit_list;      // samples that have a lit hit: ascending and without gaps, counters[RT_CNT_LIT] of them;
                             // with RtBatch::direct_lit those of chunk c in the order they were found, from
                             // c * RT_LIT_CHUNK on, block_counts[c] of them
    uint32_t capacity;       // samples a batch can hold
    uint32_t lit_f32;        // set per frame: the lit table's vectors are stored as f32 (fast mode), see lit_store_vec
    // Ray trees traced level by level (scenes with a GGXMirror shader, rt_backend.hip): the rays of one level of all
    // the batch's trees are the "samples" of a pass through the stages above, and a ray whose chain ends on a GGX
    // surface leaves that hit here, in its own slot, for the pass that spawns the next level.  Null otherwise.
    double *g_p[3];          // hit point
    double *g_n[3];          // unit normal
    double *g_in[3];         // normalised direction of the ray that hit (ggx_mirror.rs:62)
    int32_t *g_shape;
    int32_t *g_tri;
    int32_t *g_mark;         // (shader index << 8) | depth of the hit; < 0: the slot's ray did not end on a GGX surface
};

#define RT_LIT_CHUNK 1024u

enum {
    RT_CNT_RAYS0 = 0,      // [1 .. 31]: rays in the queue of bounce b (b >= 1)
    RT_CNT_LIT = 32,       // lit hits of the batch
    RT_CNT_UNUSED = 33,
    // ray-tree path (rt_tree.h): 64-bit counts, two dwords each
    RT_CNT_TREE_SECONDARY = 34,
    RT_CNT_TREE_SHADOW = 36,
    RT_CNT_TREE_SHADED = 38,
    RT_CNT_SIZE = 40
};

// Which pixels a batch covers.  Local pixel lp of the batch lies in tile slot
// slot0 + lp / (tile_w*tile_h); tile_list maps a slot to the row-major tile index.
struct RtBatch {
    const int32_t *tile_list;  // null: every tile is owned, slot k is tile k
    const int32_t *tile_place; // compact framebuffer only: the tile of slot k is stored at place tile_place[k] (null: at place k)
    uint32_t slot0;       // first tile slot of this batch
    uint32_t n_pixels;    // local pixels (tile slots x tile_w x tile_h)
    uint32_t tile_w, tile_h, tiles_x;
    uint32_t width, height;  // of the frame (they key the sample index of the stochastic sample offsets)
    // The pixels that are rendered: the window (win_x0, win_y0, win_w, win_h) of the frame, which the tiles are
    // laid over from its lower left corner; outputs (framebuffer, ids) are indexed inside the window.
    // win_w = 0 on the way into rt_batch_finish: the whole frame.
    uint32_t win_x0, win_y0, win_w, win_h;
    uint32_t n;           // sqrt_rays_per_pixel
    uint32_t compact_fb;  // 1: framebuffer is tile-major over the owned tiles, 0: full image layout
    uint32_t fused_shadows;  // 1: shadow rays and light sums run in one kernel that reads l_self, not the occ_* arrays
    uint32_t light_sums_f32; // 1 (fast mode): add_light<true>; 0 (reference-order mode): the reference's f64 arithmetic
    uint32_t chunked_queue;  // 1: the camera rays' mirror hits go into a stretch of queue 1 per chunk of samples (RtBuffers::mirror_counts)
    uint32_t direct_lit;     // 1: the kernels that find lit hits list them themselves, chunk by chunk (RtBuffers::lit_list)
    int32_t aa_method;    // RT_AA_*
    uint64_t seed;        // key of the sample-offset generator
    // Order of the pixels inside a tile: in blocks of (1 << blk_w_log2) x (1 << blk_h_log2) pixels,
    // row-major inside a block and from block to block.  The host picks the block so that it holds
    // about 64 samples: the 64 consecutive samples of a wavefront then cover a compact patch of the
    // image instead of a strip one pixel high, and the nodes their rays visit overlap more.
    // 0 / 0 = plain row-major order.
    uint32_t blk_w_log2, blk_h_log2;
    // divisors of the index arithmetic below, filled by rt_batch_finish
    RtDiv div_tile_px;      // tile_w * tile_h
    RtDiv div_blocks_row;   // tile_w >> blk_w_log2
    RtDiv div_tiles_x;
    RtDiv div_spp;          // n * n
    RtDiv div_n;
};

// Call once all other fields are set.
inline void rt_batch_finish(RtBatch &B) {
    if (B.win_w == 0 || B.win_h == 0) {
        B.win_x0 = B.win_y0 = 0;
        B.win_w = B.width;
        B.win_h = B.height;
    }
    B.div_tile_px = rt_div_make(B.tile_w * B.tile_h);
    B.div_blocks_row = rt_div_make(B.tile_w >> B.blk_w_log2);
    B.div_tiles_x = rt_div_make(B.tiles_x);
    B.div_spp = rt_div_make(B.n * B.n);
    B.div_n = rt_div_make(B.n);
}

struct RtIdOut {  // RtRayIds, device pointers (any may be null)
    int32_t *primary_shape, *primary_tri, *final_shape, *final_tri, *bounces, *shadow_shape, *shadow_tri;
};

// Pixel (i, j) of the FRAME that local pixel lp of the batch is; false for the padding pixels of a tile that
// sticks out of the rendered window.
RT_HD bool batch_pixel(const RtBatch &B, uint32_t lp, uint32_t &i, uint32_t &j) {
    const uint32_t in_batch = rt_div(lp, B.div_tile_px), r = lp - in_batch * B.div_tile_px.d;
    const uint32_t tile = B.tile_list ? (uint32_t)B.tile_list[B.slot0 + in_batch] : B.slot0 + in_batch;
    const uint32_t bshift = B.blk_w_log2 + B.blk_h_log2;
    const uint32_t blk = r >> bshift, in_blk = r & ((1u << bshift) - 1u);
    const uint32_t blk_y = rt_div(blk, B.div_blocks_row), blk_x = blk - blk_y * B.div_blocks_row.d;
    const uint32_t x = (blk_x << B.blk_w_log2) + (in_blk & ((1u << B.blk_w_log2) - 1u));
    const uint32_t y = (blk_y << B.blk_h_log2) + (in_blk >> B.blk_w_log2);
    const uint32_t tile_y = rt_div(tile, B.div_tiles_x), tile_x = tile - tile_y * B.div_tiles_x.d;
    const uint32_t wi = tile_x * B.tile_w + x, wj = tile_y * B.tile_h + y;  // inside the window
    i = B.win_x0 + wi;
    j = B.win_y0 + wj;
    return wi < B.win_w && wj < B.win_h;
}

// Index of frame pixel (i, j) in the outputs, which cover the rendered window.
RT_HD size_t output_pixel(const RtBatch &B, uint32_t i, uint32_t j) {
    return (size_t)(i - B.win_x0) + (size_t)(j - B.win_y0) * B.win_w;
}

// Sample index of RtRayIds (inside the rendered window) for a local sample.
RT_HD size_t global_sample(const RtBatch &B, uint32_t ls) {
    const uint32_t lp = rt_div(ls, B.div_spp);
    uint32_t i, j;
    batch_pixel(B, lp, i, j);
    return output_pixel(B, i, j) * B.div_spp.d + (ls - lp * B.div_spp.d);
}

// Index of a local sample in the whole FRAME, (i + j * width) * n*n + p * n + q: what keys the counter-based random
// numbers, so that a pixel does not depend on the window or part it is rendered in.
RT_HD uint64_t frame_sample(const RtBatch &B, uint32_t ls) {
    const uint32_t lp = rt_div(ls, B.div_spp);
    uint32_t i, j;
    batch_pixel(B, lp, i, j);
    return ((uint64_t)i + (uint64_t)j * B.width) * B.div_spp.d + (ls - lp * B.div_spp.d);
}

RT_HD int32_t tri_file_index(const RtSceneView &S, int32_t gtri) {
    return gtri < 0 ? -1 : reinterpret_cast<const RtTri *>(S.tris)[gtri].orig;
}

// Camera ray of a local sample (render.rs:64-68); false for the padding pixels of
// a tile that sticks out of the image.
RT_HD bool sample_plane_point(const RtSceneView &S, const RtBatch &B, uint32_t ls, double &u, double &v) {
    const uint32_t spp = B.div_spp.d;
    const uint32_t lp = rt_div(ls, B.div_spp);
    uint32_t i, j;
    if (!batch_pixel(B, lp, i, j)) return false;
    const uint32_t k = ls - lp * spp;
    const uint32_t p = rt_div(k, B.div_n);
    double di, dj;
    sample_offsets(B.aa_method, B.seed, ((uint64_t)i + (uint64_t)j * B.width) * spp + k, p, k - p * B.n, B.n, di, dj);
    camera_plane_point(S.camera, i, j, di, dj, u, v);
    return true;
}
RT_HD bool sample_camera_ray(const RtSceneView &S, const RtBatch &B, uint32_t ls, D3 &o, D3 &d) {
    double u, v;
    if (!sample_plane_point(S, B, ls, u, v)) return false;
    camera_ray_through(S.camera, u, v, o, d);
    return true;
}

RT_HD void queue_read_ray(const RtBuffers &Q, int which, uint32_t slot, D3 &o, D3 &d) {
    o = D3{RT_STREAM_LD(Q.q_o[which][0][slot]), RT_STREAM_LD(Q.q_o[which][1][slot]), RT_STREAM_LD(Q.q_o[which][2][slot])};
    d = D3{RT_STREAM_LD(Q.q_d[which][0][slot]), RT_STREAM_LD(Q.q_d[which][1][slot]), RT_STREAM_LD(Q.q_d[which][2][slot])};
}

// Point, normal and ray direction of a lit hit in the lit table.  In reference-order mode the vectors are
// kept as computed (f64).  In fast mode their consumers round them to f32 first thing -- the shadow rays'
// tree search takes an f32 origin, add_light<true> f32 vectors -- so they are stored rounded, in the
// front half of the same arrays: 12 bytes each instead of 24 (RtBuffers::lit_f32).
RT_HD void lit_store_vec(double *const (&arr)[3], uint32_t sample, const D3 &v, bool as_f32) {
    if (as_f32) {
        RT_STREAM_ST(reinterpret_cast<float *>(arr[0])[sample], (float)v.x);
        RT_STREAM_ST(reinterpret_cast<float *>(arr[1])[sample], (float)v.y);
        RT_STREAM_ST(reinterpret_cast<float *>(arr[2])[sample], (float)v.z);
    } else {
        RT_STREAM_ST(arr[0][sample], v.x); RT_STREAM_ST(arr[1][sample], v.y); RT_STREAM_ST(arr[2][sample], v.z);
    }
}
RT_HD D3 lit_load_vec(double *const (&arr)[3], uint32_t sample, bool as_f32) {
    if (as_f32)
        return D3{(double)RT_STREAM_LD(reinterpret_cast<const float *>(arr[0])[sample]),
                  (double)RT_STREAM_LD(reinterpret_cast<const float *>(arr[1])[sample]),
                  (double)RT_STREAM_LD(reinterpret_cast<const float *>(arr[2])[sample])};
    return D3{RT_STREAM_LD(arr[0][sample]), RT_STREAM_LD(arr[1][sample]), RT_STREAM_LD(arr[2][sample])};
}

// ---- trace (closest): render.rs:70 / perfect_mirror.rs:27.
// Hit::new gives camera rays t_min = 1.0 (hit_struct.rs:20); mirror rays get
// VERY_SMALL_NUMBER (perfect_mirror.rs:25).  idx = local sample (bounce 0) or queue slot.

struct RaySetup {
    D3 o, d;            // the f64 ray
    float t0;           // initial search bound (or the origin primitive's own re-hit)
    int32_t shape0;     // -1, or the origin primitive when t0 is its re-hit
    int32_t org_shape, org_tri;
};

RT_HD float closest_tmin(int bounce) { return bounce == 0 ? 1.0f : (float)RT_VERY_SMALL_NUMBER; }

// False for the padding samples of a tile that sticks out of the image.
RT_HD bool closest_setup(const RtSceneView &S, const RtBuffers &Q, const RtBatch &B, int bounce, uint32_t idx,
                         RaySetup &r) {
    r.t0 = __builtin_inff();
    r.shape0 = -1;
    r.org_shape = -1;
    r.org_tri = -1;
    if (bounce == 0) return sample_camera_ray(S, B, idx, r.o, r.d);
    const int which = bounce & 1;
    queue_read_ray(Q, which, idx, r.o, r.d);
    r.org_shape = Q.q_org_shape[which][idx];
    r.org_tri = Q.q_org_tri[which][idx];
    const float self_t = Q.q_self_t[which][idx];
    if (self_t < r.t0) {
        r.t0 = self_t;
        r.shape0 = r.org_shape;
    }
    return true;
}

RT_HD void closest_store(const RtBuffers &Q, uint32_t idx, float t, int32_t shape, int32_t tri) {
    RT_STREAM_ST(Q.h_t[idx], t);
    RT_STREAM_ST(Q.h_shape[idx], shape);
    RT_STREAM_ST(Q.h_tri[idx], tri);
}

// Reference-shaped traversal of one ray (both modes can use it; the fast mode's kernels
// use the resumable form instead).
template <class Stack, bool ORDERED, bool EXACT>
RT_HD void trace_closest_body(const RtSceneView &S, const RtBuffers &Q, const RtBatch &B, int bounce, uint32_t idx,
                              Stack &stk) {
    RayF ray;
    HitF hit;
    int32_t org_shape, org_tri;
    {
        RaySetup r;
        if (!closest_setup(S, Q, B, bounce, idx, r)) return;
        ray_from_f64<EXACT>(r.o, r.d, ray);
        hit.t = r.t0;
        hit.shape = r.shape0;
        hit.tri = -1;
        org_shape = r.org_shape;
        org_tri = r.org_tri;
    }
    if (!traverse<Stack, ORDERED, false, EXACT>(S, ray, closest_tmin(bounce), hit, org_shape, org_tri, stk, S.stack_depth))
        *Q.overflow = 1;
    closest_store(Q, idx, hit.t, hit.shape, hit.tri);
}

// Fast mode, one ray per lane, start to finish (mirror bounces; the packet kernels replace it for coherent rays).
template <class Stack>
RT_HD void trace_closest_fast_body(const RtSceneView &S, const RtBuffers &Q, const RtBatch &B, int bounce,
                                   uint32_t idx, Stack &stk) {
    FastTrav st;
    {
        RaySetup r;
        if (!closest_setup(S, Q, B, bounce, idx, r)) return;
        fast_begin(S, st, stk, r.o, r.d, r.t0, r.shape0, r.org_shape, r.org_tri);
    }
    bool overflow = false;
    const float tmin = closest_tmin(bounce);
    while (fast_step<Stack, false>(S, st, stk, tmin, S.stack_depth, overflow)) {
    }
    if (overflow) *Q.overflow = 1;
    closest_store(Q, idx, st.t, st.shape, st.tri);
}

// ---- shade
enum { RT_ACT_NONE = 0, RT_ACT_MIRROR = 1, RT_ACT_LIT = 2, RT_ACT_GGX = 3 };

struct MirrorTmp {
    D3 p, d;  // origin and direction of the reflected ray
    int32_t sample, shape, tri;
    float self_t;
};

RT_HD void write_color(const RtBuffers &Q, int32_t sample, float r, float g, float b) {
    float *c = Q.color + (size_t)sample * 3;
    RT_STREAM_ST(c[0], r); RT_STREAM_ST(c[1], g); RT_STREAM_ST(c[2], b);
}

// Decides what happens to one traced ray (shader/*.rs apply()).  Colours that need
// no further ray are written here; a lit hit goes into the sample's slot of the lit
// table together with the origin primitive's own shadow test; a mirror ray is
// returned for the caller to append to the next queue.
// (o, d) is the traced ray of `sample` and (shape, gtri, hit_t) its hit record as the trace stage leaves it.
RT_HD int shade_hit(const RtSceneView &S, const RtBuffers &Q, const RtBatch &B, const RtIdOut &ids, int bounce,
                    int32_t sample, const D3 &o, const D3 &d, int32_t shape, int32_t gtri, float hit_t, MirrorTmp &tmp) {
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
    const double tmin = bounce == 0 ? 1.0 : RT_VERY_SMALL_NUMBER;
    double t = (double)hit_t;
    D3 n;
    exact_prim_hit(S, shape, gtri, o, d, tmin, (double)__builtin_inff(), true, t, t, &n);
    const D3 p = o + d * t;  // Hit::hit_point, hit_struct.rs:41-43
    // Fast mode only: p is known to lie ON the primitive (self_hit_f64's box_leaves).  Not so when the ray
    // started inside a box: the reference then reports the hit at t_min (bbox.rs:124), a point inside it.
    const bool on_surface = Q.lit_f32 != 0u && t > tmin;
    const RtShader &sh = S.shaders[S.shapes[shape].shader];
    switch (sh.type) {
    case RT_SHADER_LAMBERTIAN:
    case RT_SHADER_BLINN_PHONG: {
        lit_store_vec(Q.l_p, (uint32_t)sample, p, Q.lit_f32 != 0u);
        lit_store_vec(Q.l_n, (uint32_t)sample, n, Q.lit_f32 != 0u);
        if (sh.type == RT_SHADER_BLINN_PHONG) lit_store_vec(Q.l_d, (uint32_t)sample, d, Q.lit_f32 != 0u);  // only its highlight looks at the ray
        RT_STREAM_ST(Q.l_shape[sample], shape);
        RT_STREAM_ST(Q.l_tri[sample], gtri);
        uint32_t self_mask = 0;
        if (!S.disable_shadows) {
            // Hit::to_light (hit_struct.rs:29-39) against the primitive the shadow ray starts on
            for (int ls = 0; ls < S.n_point_lights; ++ls) {
                const D3 dir = d3(S.lights[S.point_lights[ls]].position) - p;  // Ray::atob, ray.rs:15-20
                double st = 0.0;
                const bool self = self_hit_f64(S, shape, p, dir, RT_VERY_SMALL_NUMBER, 1.0, dot(n, dir) > 0.0,
                                               on_surface && leaves_clearly(n, dir), st);
                if (B.fused_shadows) {
                    self_mask |= self ? 1u << (ls & 31) : 0u;
                    continue;
                }
                const size_t k = (size_t)ls * Q.capacity + (size_t)sample;
                RT_STREAM_ST(Q.occ_shape[k], self ? shape : -1);
                if (self) {  // rare; the other two arrays are only read where occ_shape says "blocked"
                    Q.occ_t[k] = (float)st;
                    Q.occ_tri[k] = -1;
                }
            }
        }
        if (B.fused_shadows) RT_STREAM_ST(Q.l_self[sample], self_mask);
        return RT_ACT_LIT;
    }
    case RT_SHADER_PERFECT_MIRROR: {  // perfect_mirror.rs:9-32
        if (bounce >= S.recursion_depth) {
            write_color(Q, sample, S.background[0], S.background[1], S.background[2]);
            return RT_ACT_NONE;
        }
        tmp.p = p;
        tmp.d = mirror_direction(d, n);
        tmp.sample = sample;
        tmp.shape = shape;
        tmp.tri = gtri;
        double st;
        tmp.self_t = self_hit_f64(S, shape, tmp.p, tmp.d, RT_VERY_SMALL_NUMBER, (double)__builtin_inff(),
                                  dot(n, tmp.d) > 0.0, on_surface && leaves_clearly(n, tmp.d), st)
                         ? (float)st
                         : __builtin_inff();
        return RT_ACT_MIRROR;
    }
    case RT_SHADER_GGX_MIRROR: {  // ggx_mirror.rs:54-94, the part in front of the loop over the reflected rays
        if (bounce >= S.recursion_depth || Q.g_mark == nullptr) {
            write_color(Q, sample, S.background[0], S.background[1], S.background[2]);
            return RT_ACT_NONE;
        }
        const D3 incoming = normalize(d);
        for (int k = 0; k < 3; ++k) {
            RT_STREAM_ST(Q.g_p[k][sample], k == 0 ? p.x : k == 1 ? p.y : p.z);
            RT_STREAM_ST(Q.g_n[k][sample], k == 0 ? n.x : k == 1 ? n.y : n.z);
            RT_STREAM_ST(Q.g_in[k][sample], k == 0 ? incoming.x : k == 1 ? incoming.y : incoming.z);
        }
        RT_STREAM_ST(Q.g_shape[sample], shape);
        RT_STREAM_ST(Q.g_tri[sample], gtri);
        RT_STREAM_ST(Q.g_mark[sample], (int32_t)(((uint32_t)S.shapes[shape].shader << 8) | (uint32_t)bounce));
        return RT_ACT_GGX;
    }
    case RT_SHADER_NORMAL:  // normal.rs:10-16
        write_color(Q, sample, (1.0f + (float)n.x) / 2.0f, (1.0f + (float)n.y) / 2.0f, (1.0f + (float)n.z) / 2.0f);
        return RT_ACT_NONE;
    default:  // null.rs:7-9
        write_color(Q, sample, 1.0f, 0.0f, 1.0f);
        return RT_ACT_NONE;
    }
}

// The same for a hit record that lies in the batch's hit arrays, the ray computed (bounce 0) or read again.
RT_HD int shade_classify(const RtSceneView &S, const RtBuffers &Q, const RtBatch &B, const RtIdOut &ids, int bounce,
                         uint32_t idx, MirrorTmp &tmp) {
    // the hit record first: its loads are then in flight while the camera ray is computed again
    const int32_t shape = RT_STREAM_LD(Q.h_shape[idx]), gtri = RT_STREAM_LD(Q.h_tri[idx]);
    const float hit_t = RT_STREAM_LD(Q.h_t[idx]);
    D3 o, d;
    int32_t sample;
    if (bounce == 0) {
        sample = (int32_t)idx;
        Q.l_shape[sample] = -1;  // every sample passes here exactly once: this clears the lit table
        if (!sample_camera_ray(S, B, idx, o, d)) return RT_ACT_NONE;
    } else {
        const int which = bounce & 1;
        sample = Q.q_sample[which][idx];
        queue_read_ray(Q, which, idx, o, d);
    }
    return shade_hit(S, Q, B, ids, bounce, sample, o, d, shape, gtri, hit_t, tmp);
}

// Puts a reflected ray into slot `slot` of mirror-ray queue `which`.
RT_HD void queue_write_ray(const RtBuffers &Q, int which, uint32_t slot, const MirrorTmp &tmp) {
    RT_STREAM_ST(Q.q_o[which][0][slot], tmp.p.x); RT_STREAM_ST(Q.q_o[which][1][slot], tmp.p.y); RT_STREAM_ST(Q.q_o[which][2][slot], tmp.p.z);
    RT_STREAM_ST(Q.q_d[which][0][slot], tmp.d.x); RT_STREAM_ST(Q.q_d[which][1][slot], tmp.d.y); RT_STREAM_ST(Q.q_d[which][2][slot], tmp.d.z);
    RT_STREAM_ST(Q.q_sample[which][slot], tmp.sample);
    RT_STREAM_ST(Q.q_org_shape[which][slot], tmp.shape);
    RT_STREAM_ST(Q.q_org_tri[which][slot], tmp.tri);
    RT_STREAM_ST(Q.q_self_t[which][slot], tmp.self_t);
}
RT_HD void shade_commit_mirror(const RtBuffers &Q, int bounce, uint32_t slot, const MirrorTmp &tmp) {
    queue_write_ray(Q, (bounce + 1) & 1, slot, tmp);
}

// ---- trace (shadow): light/point.rs:30-37.  Work item = (point light slot, local sample).

// False when there is nothing to trace (no lit hit, or already known to be blocked).
template <bool ANYHIT>
RT_HD bool shadow_setup(const RtSceneView &S, const RtBuffers &Q, uint32_t light_slot, uint32_t sample, RaySetup &r) {
    r.org_shape = Q.l_shape[sample];
    if (r.org_shape < 0) return false;
    const size_t k = (size_t)light_slot * Q.capacity + sample;
    r.shape0 = Q.occ_shape[k];
    r.t0 = r.shape0 >= 0 ? Q.occ_t[k] : 1.0f;  // the origin primitive's own re-hit, or Hit::to_light's 1.0
    if (ANYHIT && r.shape0 >= 0) return false;
    r.org_tri = Q.l_tri[sample];
    const RtLight &light = S.lights[S.point_lights[light_slot]];
    r.o = lit_load_vec(Q.l_p, sample, Q.lit_f32 != 0u);
    r.d = d3(light.position) - r.o;  // Ray::atob, ray.rs:15-20
    return true;
}

// The same for an entry of the lit list (whose samples all hold a lit hit): every field is
// fetched before any is looked at, so the loads are in flight together instead of one
// round trip to memory after the other.
RT_HD bool shadow_setup_lit(const RtSceneView &S, const RtBuffers &Q, uint32_t light_slot, uint32_t sample, RaySetup &r) {
    const size_t k = (size_t)light_slot * Q.capacity + sample;
    const int32_t org_shape = RT_STREAM_LD(Q.l_shape[sample]), org_tri = RT_STREAM_LD(Q.l_tri[sample]), shape0 = RT_STREAM_LD(Q.occ_shape[k]);
    const float t0 = 1.0f;  // Hit::to_light; a sample whose own primitive blocks the light is not traced at all
    const D3 p = lit_load_vec(Q.l_p, sample, Q.lit_f32 != 0u);
    const RtLight &light = S.lights[S.point_lights[light_slot]];
    r.org_shape = org_shape;
    r.org_tri = org_tri;
    r.t0 = t0;
    r.shape0 = shape0;
    r.o = p;
    r.d = d3(light.position) - r.o;
    return org_shape >= 0 && shape0 < 0;
}

RT_HD void shadow_store(const RtBuffers &Q, uint32_t light_slot, uint32_t sample, int32_t shape, int32_t tri) {
    const size_t k = (size_t)light_slot * Q.capacity + sample;
    RT_STREAM_ST(Q.occ_shape[k], shape);
    if (shape >= 0) Q.occ_tri[k] = tri;  // only read where occ_shape says "blocked"
}

template <class Stack, bool ORDERED, bool ANYHIT, bool EXACT>
RT_HD void trace_shadow_body(const RtSceneView &S, const RtBuffers &Q, uint32_t light_slot, uint32_t sample,
                             Stack &stk) {
    RayF ray;
    HitF hit;
    int32_t org_shape, org_tri;
    {
        RaySetup r;
        if (!shadow_setup<ANYHIT>(S, Q, light_slot, sample, r)) return;
        ray_from_f64<EXACT>(r.o, r.d, ray);
        hit.t = r.t0;
        hit.shape = r.shape0;
        hit.tri = -1;
        org_shape = r.org_shape;
        org_tri = r.org_tri;
    }
    if (!traverse<Stack, ORDERED, ANYHIT, EXACT>(S, ray, (float)RT_VERY_SMALL_NUMBER, hit, org_shape, org_tri, stk,
                                                 S.stack_depth))
        *Q.overflow = 1;
    shadow_store(Q, light_slot, sample, hit.shape, hit.tri);
}

template <class Stack>
RT_HD void trace_shadow_fast_body(const RtSceneView &S, const RtBuffers &Q, uint32_t light_slot, uint32_t sample,
                                  Stack &stk) {
    FastTrav st;
    {
        RaySetup r;
        if (!shadow_setup<true>(S, Q, light_slot, sample, r)) return;
        fast_begin(S, st, stk, r.o, r.d, r.t0, r.shape0, r.org_shape, r.org_tri);
    }
    bool overflow = false;
    while (fast_step<Stack, true>(S, st, stk, (float)RT_VERY_SMALL_NUMBER, S.stack_depth, overflow)) {
    }
    if (overflow) *Q.overflow = 1;
    shadow_store(Q, light_slot, sample, st.shape, st.tri);
}

// ---- accumulate: lambertian.rs:17-36 / blinn_phong.rs:23-47.  Returns true if the
// sample had a lit hit (callers normally only pass such samples).
// The same sum for a lit sample whose shadow rays were traced by the caller: bit k of `blocked` = point
// light k (in scene order among the point lights) is hidden.
// `p`: the hit point as the lit table holds it (the caller has it at hand).
template <bool F32>
RT_HD void accumulate_masked(const RtSceneView &S, const RtBuffers &Q, uint32_t sample, int32_t shape, const D3 &p, uint32_t blocked) {
    // The view direction is asked for whether or not the shader turns out to look at it (only a Blinn-Phong hit has
    // stored one; what lies in a Lambertian hit's slot is never used): asked for once the shader's type is known, it
    // was a fourth memory round trip in a row -- list entry, shape, shader, direction -- at the end of every wave.
    const D3 n = lit_load_vec(Q.l_n, sample, F32);
    const D3 d = lit_load_vec(Q.l_d, sample, F32);
    const RtShader &sh = S.shaders[S.shapes[shape].shader];
    ColorF color = {0.0f, 0.0f, 0.0f};
    int slot = 0;
    for (int li = 0; li < S.n_lights; ++li) {
        const RtLight &light = S.lights[li];
        if (light.type == RT_LIGHT_AMBIENT) {  // ambient.rs:25-27
            add_light<F32>(sh, light, n, n, d, color);
            continue;
        }
        if (((blocked >> (slot & 31)) & 1u) == 0u) add_light<F32>(sh, light, n, d3(light.position) - p, d, color);
        ++slot;
    }
    write_color(Q, (int32_t)sample, color.r, color.g, color.b);
}

RT_HD bool accumulate_body(const RtSceneView &S, const RtBuffers &Q, const RtBatch &B, const RtIdOut &ids,
                           uint32_t sample) {
    const int32_t shape = Q.l_shape[sample];
    if (shape < 0) return false;
    const D3 p = lit_load_vec(Q.l_p, sample, Q.lit_f32 != 0u);
    const RtShader &sh = S.shaders[S.shapes[shape].shader];
    const D3 n = lit_load_vec(Q.l_n, sample, Q.lit_f32 != 0u);
    const D3 d = sh.type == RT_SHADER_BLINN_PHONG ? lit_load_vec(Q.l_d, sample, Q.lit_f32 != 0u) : D3{0.0, 0.0, -1.0};
    const bool want_ids = ids.shadow_shape || ids.shadow_tri;
    const size_t gs = want_ids ? global_sample(B, sample) : 0;
    ColorF color = {0.0f, 0.0f, 0.0f};
    int slot = 0;
    for (int li = 0; li < S.n_lights; ++li) {
        const RtLight &light = S.lights[li];
        if (light.type == RT_LIGHT_AMBIENT) {  // ambient.rs:25-27
            if (B.light_sums_f32) add_light<true>(sh, light, n, n, d, color);
            else add_light<false>(sh, light, n, n, d, color);
            continue;
        }
        bool visible = true;
        if (!S.disable_shadows) {
            const size_t k = (size_t)slot * Q.capacity + sample;
            const int32_t blocker = RT_STREAM_LD(Q.occ_shape[k]);
            visible = blocker < 0;
            if (ids.shadow_shape) ids.shadow_shape[gs * (size_t)S.n_point_lights + slot] = blocker;
            if (ids.shadow_tri)
                ids.shadow_tri[gs * (size_t)S.n_point_lights + slot] = blocker >= 0 ? tri_file_index(S, Q.occ_tri[k]) : -1;
        }
        if (visible) {
            if (B.light_sums_f32) add_light<true>(sh, light, n, d3(light.position) - p, d, color);
            else add_light<false>(sh, light, n, d3(light.position) - p, d, color);
        }
        ++slot;
    }
    write_color(Q, (int32_t)sample, color.r, color.g, color.b);
    return true;
}

// ---- resolve: render.rs:63-83, framebuffer.rs:21-31
// Where pixel lp of the batch lies in the framebuffer (false: a padding pixel of a tile that sticks out).
RT_HD bool resolve_address(const RtBatch &B, uint32_t lp, float *fb, float *&px) {
    uint32_t i, j;
    if (!batch_pixel(B, lp, i, j)) return false;
    if (B.compact_fb) {
        const uint32_t tp = B.tile_w * B.tile_h;  // tiles are stored row-major whatever the order they were worked in
        const uint32_t slot = B.slot0 + rt_div(lp, B.div_tile_px);
        px = fb + ((size_t)(B.tile_place ? (uint32_t)B.tile_place[slot] : slot) * tp + (size_t)((j - B.win_y0) % B.tile_h) * B.tile_w + (i - B.win_x0) % B.tile_w) * 3;
    } else {
        px = fb + output_pixel(B, i, j) * 3;
    }
    return true;
}

RT_HD void resolve_body(const RtBuffers &Q, const RtBatch &B, uint32_t lp, float *fb) {
    float *px;
    if (!resolve_address(B, lp, fb, px)) return;
    const uint32_t spp = B.n * B.n;
    const float *c = Q.color + (size_t)lp * spp * 3;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t k = 0; k < spp; ++k) {  // p-major, q-minor: k = p*n + q
        r += c[k * 3 + 0];
        g += c[k * 3 + 1];
        b += c[k * 3 + 2];
    }
    const float div = (float)(uint16_t)(B.n * B.n);
    px[0] = r / div;
    px[1] = g / div;
    px[2] = b / div;
}
