/* rt_oracle.h -- C interface of the CPU oracle (test infrastructure only; see
 * rt_oracle.cpp).  Loaded by the test harness through ctypes. */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#include "../include/rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-sample outputs, same indexing and meaning as RtRayIds in rt_hip.h:
 * s = (i + j*width)*n*n + p*n + q.  Any pointer may be NULL. */
typedef struct OracleOutputs {
    int32_t *primary_shape;
    int32_t *primary_tri;
    int32_t *final_shape;
    int32_t *final_tri;
    int32_t *bounces;
    int32_t *shadow_shape; /* n_point_lights per sample: blocker, -1 visible, -2 no ray */
    int32_t *shadow_tri;
} OracleOutputs;

/* Index 0 = camera rays, 1 = mirror rays, 2 = shadow rays. */
typedef struct OracleCounters {
    uint64_t rays[3];
    uint64_t node_tests[3];     /* BVH nodes entered (one box test each), bvh.rs:95 */
    uint64_t sphere_tests[3];
    uint64_t triangle_tests[3];
    uint64_t box_tests[3];
    uint64_t instance_visits[3];
    uint64_t shaded_hits;
} OracleCounters;

/* geometry_bits: 64 = the reference CLI's arithmetic; 32 = the same formulas in f32.
 * threads: OpenMP threads over pixel columns (<= 0: runtime default). */
int oracle_render(const RtScene *scene, uint16_t sqrt_rays_per_pixel, int geometry_bits, int threads,
                  float *fb_rgb, const OracleOutputs *outputs, OracleCounters *counters);

/* The same with antialias.rs's three methods (0 normal, 1 jittered, 2 random).  The uniform
 * numbers of the two stochastic methods come from Philox4x32-10 keyed by `seed` and counted by
 * the frame-wide sample index (the reference's own generator is OS-seeded). */
int oracle_render_aa(const RtScene *scene, uint16_t sqrt_rays_per_pixel, int antialias_method, uint64_t seed,
                     int geometry_bits, int threads, float *fb_rgb, const OracleOutputs *outputs,
                     OracleCounters *counters);

/* The same for the pixels [x0, x0 + width) x [y0, y0 + height) only (width = height = 0: the whole
 * frame): fb_rgb holds width*height*3 floats and the per-sample outputs width*height*n*n entries, both
 * indexed inside the window (pixel (i - x0) + (j - y0) * width).  A pixel's value is the one it has in
 * the whole frame; the counters cover the window's rays. */
int oracle_render_window(const RtScene *scene, uint16_t sqrt_rays_per_pixel, int antialias_method, uint64_t seed,
                         int geometry_bits, int threads, uint32_t x0, uint32_t y0, uint32_t width, uint32_t height,
                         float *fb_rgb, const OracleOutputs *outputs, OracleCounters *counters);

void oracle_philox(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);

void oracle_sample_offsets(int method, uint64_t seed, uint64_t sample, uint16_t n, uint16_t p, uint16_t q, double *di,
                           double *dj);

int oracle_trace_rays(const RtScene *scene, uint64_t n, const double *origins, const double *directions,
                      const double *tmin, const double *tmax, int geometry_bits, double *t_out,
                      int32_t *shape_out, int32_t *tri_out, double *normal_out);

uint32_t oracle_point_light_count(const RtScene *scene);

int oracle_bbox_hit(const double mn[3], const double mx[3], const double o[3], const double d[3],
                    double tmin, double tmax, double *t_out);

int oracle_primitive_hit(int kind, const double p0[3], const double p1[3], const double p2[3], double radius,
                         const double o[3], const double d[3], double tmin, double tmax, double *t_out,
                         double normal_out[3]);

void oracle_camera_basis(const double view[3], double u_out[3], double v_out[3], double w_out[3]);

int64_t oracle_bvh_dump(const RtScene *scene, int32_t which, int64_t cap_nodes, int64_t cap_prims,
                        int32_t *kinds_out, int32_t *prims_out, double *boxes_out, int64_t *n_prims_out);

#ifdef __cplusplus
}
#endif
#endif
