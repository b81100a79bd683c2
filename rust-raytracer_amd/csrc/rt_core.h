// rt_core.h -- per-ray device functions of the render backend: fp32 BVH
// traversal and primitive intersection, the f64 "exact" primitive routines used
// once per ray, camera rays and shading.
//
// Precision split (what runs in which type, and why):
//   * Everything that runs once per BVH node or per candidate primitive is fp32:
//     the node test, the sphere / triangle / box candidate tests and the transform
//     of the ray into an instance's space while searching.  The reference-order
//     mode follows the reference's formulas operation by operation
//     (geometry/bbox.rs:76-125, sphere.rs:58-89, triangle.rs:66-118); the fast
//     mode rearranges them (see FastTrav below).
//   * Everything that runs once per RAY is f64 and is the reference's own
//     arithmetic: camera ray generation, the hit point / normal of the winning
//     primitive (including its instance transform), mirror and shadow ray set-up,
//     and the test of a secondary ray against the primitive it starts on.  That
//     costs a few percent and keeps fp32 error from accumulating along a chain of
//     bounces: each new ray starts from the same f64 point the reference would use.
//   * A secondary ray starts ON a primitive; whether it re-hits that primitive is
//     decided by t ~ 1e-16 against t_min = 1e-6 in the reference.  fp32 cannot
//     resolve that, so traversal skips the origin primitive and the caller tests
//     it separately in f64 (exact_prim_hit) -- closest-hit is order independent.
//
// The functions are written so that they also compile as ordinary C++ for a
// host-side test build (no HIP intrinsics in this file).
#pragma once

#include <stdint.h>

#include "../../include/rt_scene.h"
#include "rt_device_types.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
typedef float4 rt_f4;
#else
#define RT_HD inline
struct alignas(16) rt_f4 {
    float x, y, z, w;
};
#endif

#define RT_VERY_SMALL_NUMBER 1e-6 /* prelude.rs:42 */

// Work counters of the host test build (nodes fetched, primitives tested); nothing on the device.
#ifndef RT_TALLY
#define RT_TALLY(what)
#endif
#ifndef RT_VISIT
#define RT_VISIT(kind, index)
#endif

// Everything a kernel needs to know about the uploaded scene; passed by value.
struct RtSceneView {
    const rt_f4 *nodes;   // 4 per RtNode
    const rt_f4 *tprims;  // 4 per RtTPrim
    const rt_f4 *tris;    // 3 per RtTri
    const rt_f4 *tri_xf;  // 3 per RtTriXf, same order as tris
    const rt_f4 *tprim_xf;  // 3 per RtTPrim: rows of the instance's inverse transform (fp32), zeros otherwise
    const RtShape *shapes;
    const RtShape *protos;
    const double *verts64;          // 12 doubles per mesh triangle, same order as tris: a, b, c, unit normal
    const uint64_t *mesh_tri_base;  // first triangle of each mesh in tris / tri_xf / verts64
    const RtShader *shaders;
    const RtLight *lights;
    const int32_t *point_lights;    // indices into lights[] of the point lights, in order
    RtCamera camera;
    float background[3];
    int32_t n_lights, n_point_lights;
    int32_t tlas_root;
    int32_t recursion_depth;
    int32_t disable_shadows;
    int32_t stack_depth;
};

// ---------------------------------------------------------------- small helpers

RT_HD int32_t rt_as_int(float f) { return __builtin_bit_cast(int32_t, f); }
RT_HD float rt_as_float(int32_t i) { return __builtin_bit_cast(float, i); }
RT_HD float rt_fmax(float a, float b) { return __builtin_fmaxf(a, b); }  // NaN-ignoring, like f64::max
RT_HD float rt_fmin(float a, float b) { return __builtin_fminf(a, b); }
RT_HD double rt_dmax(double a, double b) { return __builtin_fmax(a, b); }
RT_HD double rt_dmin(double a, double b) { return __builtin_fmin(a, b); }

struct D3 {
    double x, y, z;
};
RT_HD D3 d3(double x, double y, double z) { return D3{x, y, z}; }
RT_HD D3 d3(const RtVec3 &v) { return D3{v.x, v.y, v.z}; }
RT_HD D3 operator+(const D3 &a, const D3 &b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD D3 operator-(const D3 &a, const D3 &b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD D3 operator-(const D3 &a) { return D3{-a.x, -a.y, -a.z}; }
RT_HD D3 operator*(const D3 &a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
RT_HD D3 operator/(const D3 &a, double s) { return D3{a.x / s, a.y / s, a.z / s}; }
RT_HD double dot(const D3 &a, const D3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD D3 cross(const D3 &a, const D3 &b) {
    return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RT_HD D3 normalize(const D3 &a) { return a / __builtin_sqrt(dot(a, a)); }

// Matrix3::from_columns(&[c0, c1, c2]).determinant() -- the expression of the
// reference's matrix library, used by triangle.rs:76-105.
RT_HD double det3_columns(const D3 &c0, const D3 &c1, const D3 &c2) {
    const double minor_m12_m23 = c1.y * c2.z - c1.z * c2.y;
    const double minor_m11_m23 = c0.y * c2.z - c0.z * c2.y;
    const double minor_m11_m22 = c0.y * c1.z - c0.z * c1.y;
    return c0.x * minor_m12_m23 - c1.x * minor_m11_m23 + c2.x * minor_m11_m22;
}
RT_HD float det3_columns_f(float c0x, float c0y, float c0z, float c1x, float c1y, float c1z, float c2x,
                           float c2y, float c2z) {
    const float minor_m12_m23 = c1y * c2z - c1z * c2y;
    const float minor_m11_m23 = c0y * c2z - c0z * c2y;
    const float minor_m11_m22 = c0y * c1z - c0z * c1y;
    return c0x * minor_m12_m23 - c1x * minor_m11_m23 + c2x * minor_m11_m22;
}

// Matrix4::transform_vector / transform_point on a row-major 4x4 (instance.rs:78-79, 89).
RT_HD D3 transform_vector(const double *m, const D3 &v) {
    D3 r;
    r.x = (m[0] * v.x + m[1] * v.y) + m[2] * v.z;
    r.y = (m[4] * v.x + m[5] * v.y) + m[6] * v.z;
    r.z = (m[8] * v.x + m[9] * v.y) + m[10] * v.z;
    return r;
}
RT_HD D3 transform_point(const double *m, const D3 &p) {
    D3 r = transform_vector(m, p);
    r = D3{r.x + m[3], r.y + m[7], r.z + m[11]};
    const double n = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (n != 0.0 && n != 1.0) r = r / n;  // x / 1.0 is x, bit for bit: an affine matrix costs no division
    return r;
}

// ---------------------------------------------------------------- f64 "exact" primitives

// BBox::hit, geometry/bbox.rs:76-125, sequential form.
// The reference swaps a slab's two distances when 1 / d < 0 (bbox.rs:84).  That quotient is negative exactly
// when d carries a sign bit and is not NaN (1 / -0.0 = -inf included), so it is read off d: the same
// decisions, three divisions less per call.
RT_HD bool rt_inv_negative(double d) { return __builtin_signbit(d) && d == d; }

RT_HD bool slab_f64(const D3 &mn, const D3 &mx, const D3 &o, const D3 &d, double tmin, double tmax, double &t_out) {
    double tn = (mn.x - o.x) / d.x, tf = (mx.x - o.x) / d.x;
    if (rt_inv_negative(d.x)) { double s = tn; tn = tf; tf = s; }
    if (tn >= tf || tn >= tmax || tf <= tmin) return false;
    tmax = rt_dmin(tmax, tf);
    tmin = rt_dmax(tmin, tn);
    tn = (mn.y - o.y) / d.y; tf = (mx.y - o.y) / d.y;
    if (rt_inv_negative(d.y)) { double s = tn; tn = tf; tf = s; }
    if (tn >= tf || tn >= tmax || tf <= tmin) return false;
    tmax = rt_dmin(tmax, tf);
    tmin = rt_dmax(tmin, tn);
    tn = (mn.z - o.z) / d.z; tf = (mx.z - o.z) / d.z;
    if (rt_inv_negative(d.z)) { double s = tn; tn = tf; tf = s; }
    if (tn >= tf || tn >= tmax || tf <= tmin) return false;
    t_out = rt_dmax(tn, tmin);
    return true;
}

// Sphere::closest_hit, sphere.rs:58-84 (the t selection).  `forced` returns the
// nearest admissible root even when the f64 test itself says "miss".
RT_HD bool sphere_f64(const D3 &c, double r, const D3 &o, const D3 &d, double tmin, double tmax, bool forced,
                      double &t_out) {
    const D3 oc = o - c;
    const double cd = dot(oc, d);
    double disc = cd * cd - dot(d, d) * (dot(oc, oc) - r * r);
    if (disc < 0.0) {
        if (!forced) return false;
        disc = 0.0;
    }
    const double num = -dot(oc, d), den = dot(d, d);
    const double t1 = (num - __builtin_sqrt(disc)) / den;
    const double t2 = (num + __builtin_sqrt(disc)) / den;
    const bool in1 = tmin <= t1 && t1 < tmax, in2 = tmin <= t2 && t2 < tmax;
    if (in1 && in2) t_out = rt_dmin(t1, t2);
    else if (in1) t_out = t1;
    else if (in2) t_out = t2;
    else {
        if (!forced) return false;
        t_out = t1 >= tmin ? t1 : t2;
    }
    return true;
}

// Triangle::closest_hit, triangle.rs:66-110.
RT_HD bool triangle_f64(const D3 &a, const D3 &b, const D3 &c, const D3 &o, const D3 &d, double tmin, double tmax,
                        bool forced, double &t_out) {
    const D3 ab = a - b, ac = a - c, ao = a - o;
    const double det_a = det3_columns(ab, ac, d);
    if (__builtin_fabs(det_a) < 2.220446049250313e-16 && !forced) return false;
    const double t = det3_columns(ab, ac, ao) / det_a;
    t_out = t;
    if (forced) return true;
    if (t < tmin || t > tmax) return false;
    const double gamma = det3_columns(ab, ao, d) / det_a;
    if (gamma < 0.0 || gamma > 1.0) return false;
    const double beta = det3_columns(ao, ac, d) / det_a;
    if (beta < 0.0 || beta > 1.0 - gamma) return false;
    return true;
}

// Cuboid::normal, cuboid.rs:24-39 (before normalisation).
RT_HD D3 cuboid_normal(const D3 &mn, const D3 &mx, const D3 &p) {
    const D3 centroid = (mn + mx) * 0.5;  // bbox.rs:18
    const D3 extent = mx - mn;            // bbox.rs:19
    const D3 pc = p - centroid;
    const double dx = __builtin_fabs(pc.x / extent.x), dy = __builtin_fabs(pc.y / extent.y),
                 dz = __builtin_fabs(pc.z / extent.z);
    if (dx > dy && dx > dz) return D3{pc.x > 0.0 ? 1.0 : -1.0, 0.0, 0.0};
    if (dy > dz) return D3{0.0, pc.y > 0.0 ? 1.0 : -1.0, 0.0};
    return D3{0.0, 0.0, pc.z > 0.0 ? 1.0 : -1.0};
}

// Intersects ONE identified primitive with a ray, entirely in f64 with the
// reference's formulas, instance transform included (instance.rs:75-94).
//   shape_id : index into RtScene.shapes;  gtri : global leaf-order triangle index or -1.
//   forced   : the caller already knows (from fp32 traversal) that this primitive
//              is the winner and only wants t and the normal.
// Returns true on a hit; t_out and, if non-null, the world-space unit normal.
RT_HD bool exact_prim_hit(const RtSceneView &S, int32_t shape_id, int32_t gtri, D3 o, D3 d, double tmin, double tmax,
                          bool forced, double t_fallback, double &t_out, D3 *normal_out) {
    const RtShape *sh = &S.shapes[shape_id];
    const RtShape *inst = nullptr;
    if (sh->type == RT_SHAPE_INSTANCE) {
        inst = sh;
        o = transform_point(inst->inv_transform, o);
        d = transform_vector(inst->inv_transform, d);
        sh = &S.protos[inst->proto];
    }
    D3 n = D3{0.0, 0.0, 0.0};
    bool got = false;
    double t = t_fallback;
    switch (sh->type) {
    case RT_SHAPE_SPHERE: {
        const D3 c = d3(sh->p0);
        got = sphere_f64(c, sh->radius, o, d, tmin, tmax, forced, t);
        if (got && normal_out) n = normalize((o + d * t) - c);  // sphere.rs:86
        break;
    }
    case RT_SHAPE_BOX: {
        const D3 mn = d3(sh->p0), mx = d3(sh->p1);
        got = slab_f64(mn, mx, o, d, tmin, tmax, t);
        if (!got && forced) {
            t = t_fallback;
            got = true;
        }
        // cuboid.rs:66 normalises this axis vector: a division of +-1 and 0 by sqrt(1.0) = 1.0, which changes no bit
        if (got && normal_out) n = cuboid_normal(mn, mx, o + d * t);
        break;
    }
    case RT_SHAPE_TRIANGLE: {
        const D3 a = d3(sh->p0), b = d3(sh->p1), c = d3(sh->p2);
        got = triangle_f64(a, b, c, o, d, tmin, tmax, forced, t);
        if (got && normal_out) n = normalize(cross(b - a, c - a));  // triangle.rs:22
        break;
    }
    case RT_SHAPE_MESH: {
        if (gtri < 0) break;
        const double *v = S.verts64 + (size_t)gtri * 12;
        const D3 a = d3(v[0], v[1], v[2]), b = d3(v[3], v[4], v[5]), c = d3(v[6], v[7], v[8]);
        got = triangle_f64(a, b, c, o, d, tmin, tmax, forced, t);
        if (got && normal_out) n = d3(v[9], v[10], v[11]);  // computed once per triangle, as Triangle::new does
        break;
    }
    default: break;
    }
    if (!got) return false;
    t_out = t;
    if (normal_out) {
        if (inst) n = normalize(transform_vector(inst->normal_matrix, n));  // instance.rs:89-90
        *normal_out = n;
    }
    return true;
}

// ---------------------------------------------------------------- fp32 traversal

struct RayF {
    float ox, oy, oz;      // origin
    float dx, dy, dz;      // direction, un-normalised like the reference's rays
    float ix, iy, iz;      // 1 / d (only used by the fast node test)
    uint32_t mx, my, mz;   // all ones where 1/d < 0, the reference's swap condition (bbox.rs:84), else 0
};

RT_HD uint32_t rt_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
RT_HD float rt_float(uint32_t u) { return __builtin_bit_cast(float, u); }
// m ? a : b for an all-ones / all-zero mask: one v_bfi_b32, and no per-lane boolean for the
// compiler to keep (and merge at every divergent join) in scalar mask registers.
RT_HD float rt_select(uint32_t m, float a, float b) { return rt_float((m & rt_bits(a)) | (~m & rt_bits(b))); }

// 1/x for the culling-only node test of the fast mode.
RT_HD float rt_rcp_approx(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// EXACT: correctly rounded reciprocals (they are not used by the exact node test, which
// divides, but keep the two modes' RayF identical in meaning).
template <bool EXACT>
RT_HD void ray_finish(RayF &r) {
    if (EXACT) {
        r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
    } else {
        r.ix = rt_rcp_approx(r.dx); r.iy = rt_rcp_approx(r.dy); r.iz = rt_rcp_approx(r.dz);
    }
    // 1/d < 0  <=>  d < 0 or d == -0.0: the sign bit (a NaN direction poisons everything anyway)
    r.mx = (uint32_t)((int32_t)rt_bits(r.dx) >> 31);
    r.my = (uint32_t)((int32_t)rt_bits(r.dy) >> 31);
    r.mz = (uint32_t)((int32_t)rt_bits(r.dz) >> 31);
}

// The f64 ray of the pipeline, rounded once to fp32 for traversal.
template <bool EXACT>
RT_HD void ray_from_f64(const D3 &o, const D3 &d, RayF &r) {
    r.ox = (float)o.x; r.oy = (float)o.y; r.oz = (float)o.z;
    r.dx = (float)d.x; r.dy = (float)d.y; r.dz = (float)d.z;
    ray_finish<EXACT>(r);
}

// The ray in an instance's object space (instance.rs:77-80): origin as a point,
// direction as a vector, NOT renormalised, so t means the same in both spaces.
// fp32 here; the f64 form of the same transform is applied to the winning hit
// in exact_prim_hit.
template <bool EXACT>
RT_HD void ray_to_instance(const RtSceneView &S, uint32_t tprim_slot, float wox, float woy, float woz, float wdx,
                           float wdy, float wdz, RayF &out) {
    const rt_f4 *m = S.tprim_xf + (size_t)tprim_slot * 3;
    const rt_f4 r0 = m[0], r1 = m[1], r2 = m[2];
    out.ox = ((r0.x * wox + r0.y * woy) + r0.z * woz) + r0.w;
    out.oy = ((r1.x * wox + r1.y * woy) + r1.z * woz) + r1.w;
    out.oz = ((r2.x * wox + r2.y * woy) + r2.z * woz) + r2.w;
    out.dx = (r0.x * wdx + r0.y * wdy) + r0.z * wdz;
    out.dy = (r1.x * wdx + r1.y * wdy) + r1.z * wdz;
    out.dz = (r2.x * wdx + r2.y * wdy) + r2.z * wdz;
    ray_finish<EXACT>(out);
}

// BBox::hit (bbox.rs:76-125) in closed form.  The reference's three per-axis
// stages reject exactly when  max(tn) >= min(tf)  or  max(tn) >= tmax  or
// min(tf) <= tmin, with NaN (0/0) comparing false and ignored by min/max, and
// it returns max(tn_z, tmin') = max(tmin, max(tn)).  EXACT uses true divisions
// like the reference; otherwise the per-ray reciprocal.
template <bool EXACT>
RT_HD bool slab_f32(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, const RayF &r, float tmin,
                    float tmax, float &entry) {
    const float ax = mnx - r.ox, ay = mny - r.oy, az = mnz - r.oz;
    const float bx = mxx - r.ox, by = mxy - r.oy, bz = mxz - r.oz;
    float t0x, t1x, t0y, t1y, t0z, t1z;
    if (EXACT) {
        t0x = ax / r.dx; t1x = bx / r.dx;
        t0y = ay / r.dy; t1y = by / r.dy;
        t0z = az / r.dz; t1z = bz / r.dz;
    } else {
        t0x = ax * r.ix; t1x = bx * r.ix;
        t0y = ay * r.iy; t1y = by * r.iy;
        t0z = az * r.iz; t1z = bz * r.iz;
    }
    const float tnx = rt_select(r.mx, t1x, t0x), tfx = rt_select(r.mx, t0x, t1x);
    const float tny = rt_select(r.my, t1y, t0y), tfy = rt_select(r.my, t0y, t1y);
    const float tnz = rt_select(r.mz, t1z, t0z), tfz = rt_select(r.mz, t0z, t1z);
    // per-axis emptiness (tn_k >= tf_k) is part of max(tn) >= min(tf)
    const float e = rt_fmax(rt_fmax(tnx, tny), tnz);
    const float x = rt_fmin(rt_fmin(tfx, tfy), tfz);
    if (e >= x || e >= tmax || x <= tmin) return false;
    entry = rt_fmax(e, tmin);
    return true;
}

struct HitF {
    float t;        // in: upper bound of the search interval; out: t of the hit
    int32_t shape;  // index into RtScene.shapes, -1 = nothing
    int32_t tri;    // global leaf-order triangle index, -1 when not inside a mesh
};

// Sphere::closest_hit (sphere.rs:58-84) in fp32.
//
// The reference's discriminant  (oc.d)^2 - (d.d)(oc.oc - r^2)  subtracts two
// numbers of size (|oc||d|)^2 to get one of size (r|d|)^2; in fp32 that loses
// the silhouette of a small distant sphere (1e-4 of its samples at |oc|/r = 30).
// By Lagrange's identity the same quantity is  (d.d) r^2 - |oc x d|^2, whose
// fp32 error is |oc|/(2r) times smaller, so hit/miss agrees with the f64
// reference far more often.  Roots are then formed as the reference forms them.
RT_HD bool sphere_f32(float cx, float cy, float cz, float rad, const RayF &r, float tmin, float tmax, float &t_out) {
    const float ocx = r.ox - cx, ocy = r.oy - cy, ocz = r.oz - cz;
    const float cd = ocx * r.dx + ocy * r.dy + ocz * r.dz;
    const float dd = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
    const float kx = ocy * r.dz - ocz * r.dy, ky = ocz * r.dx - ocx * r.dz, kz = ocx * r.dy - ocy * r.dx;
    const float disc = dd * (rad * rad) - (kx * kx + ky * ky + kz * kz);
    if (disc < 0.0f) return false;
    const float num = -cd;
    const float sq = __builtin_sqrtf(disc);
    const float t1 = (num - sq) / dd;
    const float t2 = (num + sq) / dd;
    const bool in1 = tmin <= t1 && t1 < tmax, in2 = tmin <= t2 && t2 < tmax;
    if (in1 && in2) t_out = rt_fmin(t1, t2);
    else if (in1) t_out = t1;
    else if (in2) t_out = t2;
    else return false;
    return true;
}

// Triangle::closest_hit (triangle.rs:66-110) in fp32, with a-b and a-c precomputed.
RT_HD bool triangle_f32(float ax, float ay, float az, float abx, float aby, float abz, float acx, float acy,
                        float acz, const RayF &r, float tmin, float tmax, float &t_out) {
    const float aox = ax - r.ox, aoy = ay - r.oy, aoz = az - r.oz;
    const float det_a = det3_columns_f(abx, aby, abz, acx, acy, acz, r.dx, r.dy, r.dz);
    if (__builtin_fabsf(det_a) < 2.220446049250313e-16f) return false;
    const float t = det3_columns_f(abx, aby, abz, acx, acy, acz, aox, aoy, aoz) / det_a;
    if (t < tmin || t > tmax) return false;
    const float gamma = det3_columns_f(abx, aby, abz, aox, aoy, aoz, r.dx, r.dy, r.dz) / det_a;
    if (gamma < 0.0f || gamma > 1.0f) return false;
    const float beta = det3_columns_f(aox, aoy, aoz, acx, acy, acz, r.dx, r.dy, r.dz) / det_a;
    if (beta < 0.0f || beta > 1.0f - gamma) return false;
    t_out = t;
    return true;
}

// One sphere / box / triangle payload (RtTPrim::f) against a ray.
RT_HD bool payload_hit_f32(int32_t type, const rt_f4 &f0, const rt_f4 &f1, const rt_f4 &f2, const RayF &r,
                           float tmin, float tmax, float &t_out) {
    if (type == RT_SHAPE_SPHERE) return sphere_f32(f0.x, f0.y, f0.z, f0.w, r, tmin, tmax, t_out);
    if (type == RT_SHAPE_BOX)  // Cuboid::closest_hit, cuboid.rs:63-73: the box test IS the hit
        return slab_f32<true>(f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, r, tmin, tmax, t_out);
    if (type == RT_SHAPE_TRIANGLE)
        return triangle_f32(f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, r, tmin, tmax, t_out);
    return false;
}

// BVH::closest_hit (geometry/bvh.rs:93-125, 140-142) over the scene-level tree and,
// nested, the trees of meshes and instanced meshes (mesh.rs:91-93, instance.rs:75-94).
//
//   ORDERED  visit the nearer child first (else left-then-right like the reference)
//   ANYHIT   return at the first accepted primitive (shadow rays)
//   EXACT    true divisions in the node test (else multiply by 1/d)
//
// `ray` is the world-space ray.  `hit` comes in holding the search bound (and, if
// the caller already found that the ray re-hits the primitive it starts on, that
// hit).  The primitive (org_shape, org_tri) is skipped here: fp32 cannot decide a
// re-hit at t ~ 0, so its owner tests it in f64 (self_hit_f64).
//
// Loop shape: lanes first walk inner nodes until every lane of the wave holds a
// leaf (or is done), then all leaves are processed together, so the two bodies
// are not interleaved lane by lane.  Returns false only on stack overflow.
template <class Stack, bool ORDERED, bool ANYHIT, bool EXACT>
RT_HD bool traverse(const RtSceneView &S, RayF ray, float tmin, HitF &hit, int32_t org_shape, int32_t org_tri,
                    Stack &stk, int max_sp) {
    const float wox = ray.ox, woy = ray.oy, woz = ray.oz, wdx = ray.dx, wdy = ray.dy, wdz = ray.dz;
    int32_t cur = S.tlas_root;
    int sp = 0;
    int32_t cur_shape = -1;  // >= 0 while inside the tree of a mesh / instanced mesh
    for (;;) {
        while (cur >= 0) {
            RT_TALLY(node);
            const rt_f4 *n = reinterpret_cast<const rt_f4 *>(reinterpret_cast<const char *>(S.nodes) + (uint32_t)cur);
            const rt_f4 a = n[0], b = n[1], c = n[2], l = n[3];
            const int32_t lref = rt_as_int(l.x), rref = rt_as_int(l.y);
            float el = 0.0f, er = 0.0f;
            // RtNode: a = left (min.x, max.x, min.y, max.y), b = right likewise, c = z planes (l min, l max, r min, r max)
            const bool hl = slab_f32<EXACT>(a.x, a.z, c.x, a.y, a.w, c.y, ray, tmin, hit.t, el);
            const bool hr =
                rref != RT_REF_NONE && slab_f32<EXACT>(b.x, b.z, c.z, b.y, b.w, c.w, ray, tmin, hit.t, er);
            if (hl && hr) {
                const bool swap = ORDERED && er < el;
                if (sp >= max_sp) return false;
                stk.set(sp++, swap ? lref : rref);
                cur = swap ? rref : lref;
            } else if (hl) {
                cur = lref;
            } else if (hr) {
                cur = rref;
            } else {
                cur = sp > 0 ? stk.get(--sp) : RT_REF_NONE;
            }
        }
        if (cur == RT_REF_NONE) break;
        if (cur == RT_REF_EXIT_BLAS) {
            // instance.rs:84 "reset the ray"
            ray.ox = wox; ray.oy = woy; ray.oz = woz; ray.dx = wdx; ray.dy = wdy; ray.dz = wdz;
            ray_finish<EXACT>(ray);
            cur_shape = -1;
        } else {
            const uint32_t v = (uint32_t)~cur;
            const uint32_t cnt = v & 7u, first = v >> 3;
            if (cur_shape >= 0) {
                for (uint32_t k = 0; k < cnt; ++k) {
                    const int32_t gtri = (int32_t)(first + k);
                    if (cur_shape == org_shape && gtri == org_tri) continue;
                    RT_TALLY(tri);
                    const rt_f4 *tp = S.tris + (size_t)gtri * 3;
                    const rt_f4 f0 = tp[0], f1 = tp[1], f2 = tp[2];
                    float t;
                    if (triangle_f32(f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, ray, tmin, hit.t, t)) {
                        hit.t = t;
                        hit.shape = cur_shape;
                        hit.tri = gtri;
                        if (ANYHIT) return true;
                    }
                }
            } else {
                bool entered = false;
                for (uint32_t k = 0; k < cnt && !entered; ++k) {
                    const rt_f4 *pp = S.tprims + (size_t)(first + k) * 4;
                    const rt_f4 h = pp[0];
                    const int32_t type = rt_as_int(h.x), shape_id = rt_as_int(h.y), blas_root = rt_as_int(h.z),
                                  proto_type = rt_as_int(h.w);
                    const bool is_inst = type == RT_SHAPE_INSTANCE;
                    RT_TALLY(tprim);
                    if (type == RT_SHAPE_MESH || (is_inst && proto_type == RT_SHAPE_MESH)) {
                        if (!EXACT && is_inst) {
                            // fast mode: skip the transform when the ray misses the instance's world box
                            // (the reference has no such test; it cannot change a hit, only save work)
                            const rt_f4 f0 = pp[1], f1 = pp[2];
                            float e;
                            if (!slab_f32<false>(f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, ray, tmin, hit.t, e)) continue;
                        }
                        // descend into the mesh; the rest of this leaf waits on the stack
                        if (sp + 2 > max_sp) return false;
                        if (k + 1 < cnt) stk.set(sp++, rt_make_leaf_ref(first + k + 1, cnt - k - 1));
                        stk.set(sp++, RT_REF_EXIT_BLAS);
                        if (is_inst) ray_to_instance<EXACT>(S, first + k, wox, woy, woz, wdx, wdy, wdz, ray);
                        cur_shape = shape_id;
                        cur = blas_root;
                        entered = true;
                    } else if (shape_id != org_shape) {
                        const rt_f4 f0 = pp[1], f1 = pp[2], f2 = pp[3];
                        RayF lr = ray;
                        if (is_inst) ray_to_instance<EXACT>(S, first + k, wox, woy, woz, wdx, wdy, wdz, lr);
                        float t;
                        if (payload_hit_f32(is_inst ? proto_type : type, f0, f1, f2, lr, tmin, hit.t, t)) {
                            hit.t = t;
                            hit.shape = shape_id;
                            hit.tri = -1;
                            if (ANYHIT) return true;
                        }
                    }
                }
                if (entered) continue;
            }
        }
        if (sp == 0) break;
        cur = stk.get(--sp);
    }
    return true;
}

// ---------------------------------------------------------------- fast-mode traversal
//
// Same primitives, same closest hit as traverse<> above (ties aside; the mesh trees are the same,
// the scene-level tree is the fast mode's own), but the arithmetic is arranged for the GPU
// rather than to mirror the reference line by line:
//   * node test: p * (1/d) - o * (1/d) as packed multiply-adds, with min/max per axis instead of
//     a sign-selected swap.  1/d is clamped to +-2^62 so that 0 * (1/0) is 0, not NaN: a ray lying
//     in a box face's plane then treats that axis as unconstrained on the entry side like the
//     reference does.  The test only culls; primitives decide hits.
//   * triangle test: an affine map per triangle (RtTriXf) gives the plane distance and both
//     barycentrics as dot products, with one reciprocal instead of triangle.rs:76-105's four
//     determinants and three divisions.
// It is written as begin / step: one step walks inner nodes to the next leaf and processes it.

// Pairs of floats as the packed fp32 instructions (v_pk_fma_f32: two multiply-adds per lane per
// instruction) take them: an aligned register pair on the device, a plain struct on the host.
RT_HD float rt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#if defined(__HIP_DEVICE_COMPILE__)
typedef float rt_f2 __attribute__((ext_vector_type(2)));
RT_HD rt_f2 rt_fma2(rt_f2 a, rt_f2 b, rt_f2 c) { return __builtin_elementwise_fma(a, b, c); }
RT_HD rt_f2 rt_pair(float x, float y) { return rt_f2{x, y}; }
RT_HD rt_f2 rt_mul2(rt_f2 a, rt_f2 b) { return a * b; }
RT_HD rt_f2 rt_lo2(rt_f2 v) { return __builtin_shufflevector(v, v, 0, 0); }
RT_HD rt_f2 rt_hi2(rt_f2 v) { return __builtin_shufflevector(v, v, 1, 1); }
#else
struct rt_f2 { float x, y; };
RT_HD rt_f2 rt_fma2(rt_f2 a, rt_f2 b, rt_f2 c) { return rt_f2{rt_fma(a.x, b.x, c.x), rt_fma(a.y, b.y, c.y)}; }
RT_HD rt_f2 rt_pair(float x, float y) { return rt_f2{x, y}; }
RT_HD rt_f2 rt_mul2(rt_f2 a, rt_f2 b) { return rt_f2{a.x * b.x, a.y * b.y}; }
RT_HD rt_f2 rt_lo2(rt_f2 v) { return rt_f2{v.x, v.x}; }
RT_HD rt_f2 rt_hi2(rt_f2 v) { return rt_f2{v.y, v.y}; }
#endif

struct FastTrav {
    rt_f2 px, py, pz;              // current-space ray, per axis (origin, direction): the operand shape of fast_triangle
    rt_f2 ixy;                     // clamped 1/d, x and y
    rt_f2 nxy;                     // -o * (1/d), x and y: a plane p is crossed at t = p * (1/d) + n
    rt_f2 zin;                     // (1/d.z, -o.z * (1/d.z))
    float t;                       // search bound / closest hit so far
    int32_t shape, tri;            // closest hit so far
    int32_t cur, sp, cur_shape;    // traversal position
    int32_t tos;                   // top of the traversal stack, kept in a register (RT_REF_NONE = empty)
    int32_t org_shape, org_tri;    // primitive the ray starts on (skipped)
};


// 1/d limited to +-2^62.  The limit only matters for a ray parallel to a slab (d = 0): the
// plane distances become +-huge with the sign of (p - o), which is all the slab logic needs,
// and they stay finite, so that p * (1/d) - o * (1/d) never meets inf - inf.  p == o gives
// exactly 0 (the limit is a power of two): a ray lying in a box face's plane treats that axis
// as unconstrained on the entry side, like the reference's 0/0 = NaN comparisons do.
RT_HD float rt_inv_clamped(float d) {
    const float big = 4611686018427387904.0f;
    const float r = rt_rcp_approx(d);
    return rt_fmin(rt_fmax(r, -big), big);
}

RT_HD void fast_set_ray(FastTrav &s, float ox, float oy, float oz, float dx, float dy, float dz) {
    s.px = rt_pair(ox, dx); s.py = rt_pair(oy, dy); s.pz = rt_pair(oz, dz);
    const float ix = rt_inv_clamped(dx), iy = rt_inv_clamped(dy);
    s.ixy = rt_pair(ix, iy);
    s.nxy = rt_pair(-(ox * ix), -(oy * iy));
    const float iz = rt_inv_clamped(dz);
    s.zin = rt_pair(iz, -(oz * iz));
}

// The world ray (wx, wy, wz: per-axis (origin, direction) pairs) in the space of an instance: through the
// rows r0..r2 of its inverse transform (instance.rs:77-80), the translation reaching the origin only.
// Each row meets the ray as two packed multiply-adds and a packed multiplication over the pairs, so origin
// and direction advance together; every traversal path of the fast mode goes through this one function,
// so they all see the same local ray.
RT_HD void fast_set_ray_in_instance(FastTrav &s, const rt_f4 &r0, const rt_f4 &r1, const rt_f4 &r2, const rt_f2 &wx, const rt_f2 &wy,
                                    const rt_f2 &wz) {
    const rt_f2 x = rt_fma2(rt_pair(r0.x, r0.x), wx, rt_fma2(rt_pair(r0.y, r0.y), wy, rt_mul2(rt_pair(r0.z, r0.z), wz)));
    const rt_f2 y = rt_fma2(rt_pair(r1.x, r1.x), wx, rt_fma2(rt_pair(r1.y, r1.y), wy, rt_mul2(rt_pair(r1.z, r1.z), wz)));
    const rt_f2 z = rt_fma2(rt_pair(r2.x, r2.x), wx, rt_fma2(rt_pair(r2.y, r2.y), wy, rt_mul2(rt_pair(r2.z, r2.z), wz)));
    fast_set_ray(s, x.x + r0.w, y.x + r1.w, z.x + r2.w, x.y, y.y, z.y);
}

// The stack keeps its top entry in a register: a pop hands that out at once and starts the
// LDS read of the entry below, whose latency is then hidden behind the next node.
template <class Stack>
RT_HD void fast_push(FastTrav &s, Stack &stk, int32_t ref) {
    stk.set(s.sp, s.tos);
    s.sp += 1;
    s.tos = ref;
}
// (Row 0 always holds the RT_REF_NONE that the first push moved there, so popping the last
// entry reloads the empty marker without a special case; a pop of the empty stack returns
// RT_REF_NONE, which ends the traversal, and what it leaves in `tos` is never used.)
template <class Stack>
RT_HD int32_t fast_pop(FastTrav &s, Stack &stk) {
    const int32_t r = s.tos;
    s.sp = s.sp > 0 ? s.sp - 1 : 0;
    s.tos = stk.get(s.sp);
    return r;
}

// Aux slots 0..5 of the stack object hold the world-space ray while a lane is inside an instance.
template <class Stack>
RT_HD void fast_begin(const RtSceneView &S, FastTrav &s, Stack &stk, const D3 &o, const D3 &d, float t0,
                      int32_t shape0, int32_t org_shape, int32_t org_tri) {
    fast_set_ray(s, (float)o.x, (float)o.y, (float)o.z, (float)d.x, (float)d.y, (float)d.z);
    stk.set_aux(0, s.px.x); stk.set_aux(1, s.py.x); stk.set_aux(2, s.pz.x);
    stk.set_aux(3, s.px.y); stk.set_aux(4, s.py.y); stk.set_aux(5, s.pz.y);
    s.t = t0;
    s.shape = shape0;
    s.tri = -1;
    s.cur = S.tlas_root;
    s.sp = 0;
    s.tos = RT_REF_NONE;
    s.cur_shape = -1;
    s.org_shape = org_shape;
    s.org_tri = org_tri;
}

// Entry distance of the ray into a box within (tmin, s.t), or a miss.  The test only culls
// (primitives decide hits); a box of zero thickness gives entry == exit and is never entered,
// like the reference's (bbox.rs:89,104,119).
RT_HD bool fast_box(const FastTrav &s, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float tmin,
                    float &entry) {
    const float t0x = rt_fma(mnx, s.ixy.x, s.nxy.x), t1x = rt_fma(mxx, s.ixy.x, s.nxy.x);
    const float t0y = rt_fma(mny, s.ixy.y, s.nxy.y), t1y = rt_fma(mxy, s.ixy.y, s.nxy.y);
    const float t0z = rt_fma(mnz, s.zin.x, s.zin.y), t1z = rt_fma(mxz, s.zin.x, s.zin.y);
    const float e = rt_fmax(rt_fmax(rt_fmax(rt_fmin(t0x, t1x), rt_fmin(t0y, t1y)), rt_fmin(t0z, t1z)), tmin);
    const float x = rt_fmin(rt_fmin(rt_fmin(rt_fmax(t0x, t1x), rt_fmax(t0y, t1y)), rt_fmax(t0z, t1z)), s.t);
    entry = e;
    return e < x;
}

// Both child boxes of a node at once (see rt_f2 above): a = left (min.x, max.x, min.y, max.y),
// b = right likewise, c = (left min.z, max.z, right min.z, max.z).
RT_HD void fast_box2(const FastTrav &s, const rt_f4 &a, const rt_f4 &b, const rt_f4 &c, float tmin, float &el, float &er,
                     bool &hl, bool &hr) {
    const rt_f2 ixx = rt_lo2(s.ixy), nxx = rt_lo2(s.nxy), iyy = rt_hi2(s.ixy), nyy = rt_hi2(s.nxy);
    const rt_f2 izz = rt_lo2(s.zin), nzz = rt_hi2(s.zin);
    const rt_f2 lx = rt_fma2(rt_pair(a.x, a.y), ixx, nxx), ly = rt_fma2(rt_pair(a.z, a.w), iyy, nyy);
    const rt_f2 rx = rt_fma2(rt_pair(b.x, b.y), ixx, nxx), ry = rt_fma2(rt_pair(b.z, b.w), iyy, nyy);
    const rt_f2 lz = rt_fma2(rt_pair(c.x, c.y), izz, nzz), rz = rt_fma2(rt_pair(c.z, c.w), izz, nzz);
    el = rt_fmax(rt_fmax(rt_fmax(rt_fmin(lx.x, lx.y), rt_fmin(ly.x, ly.y)), rt_fmin(lz.x, lz.y)), tmin);
    const float xl = rt_fmin(rt_fmin(rt_fmin(rt_fmax(lx.x, lx.y), rt_fmax(ly.x, ly.y)), rt_fmax(lz.x, lz.y)), s.t);
    er = rt_fmax(rt_fmax(rt_fmax(rt_fmin(rx.x, rx.y), rt_fmin(ry.x, ry.y)), rt_fmin(rz.x, rz.y)), tmin);
    const float xr = rt_fmin(rt_fmin(rt_fmin(rt_fmax(rx.x, rx.y), rt_fmax(ry.x, ry.y)), rt_fmax(rz.x, rz.y)), s.t);
    hl = el < xl;
    hr = er < xr;
}

// The same for a ray whose direction signs are known at compile time: OCT bit k set = the ray runs
// towards smaller coordinates on axis k (its 1/d there is negative), so it enters a slab through
// the max plane -- the second element of the pair -- and leaves through the min plane.
template <int OCT>
RT_HD void fast_box2_signed(const FastTrav &s, const rt_f4 &a, const rt_f4 &b, const rt_f4 &c, float tmin, bool &hl, bool &hr) {
    const rt_f2 ixx = rt_lo2(s.ixy), nxx = rt_lo2(s.nxy), iyy = rt_hi2(s.ixy), nyy = rt_hi2(s.nxy);
    const rt_f2 izz = rt_lo2(s.zin), nzz = rt_hi2(s.zin);
    const rt_f2 lx = rt_fma2(rt_pair(a.x, a.y), ixx, nxx), ly = rt_fma2(rt_pair(a.z, a.w), iyy, nyy);
    const rt_f2 rx = rt_fma2(rt_pair(b.x, b.y), ixx, nxx), ry = rt_fma2(rt_pair(b.z, b.w), iyy, nyy);
    const rt_f2 lz = rt_fma2(rt_pair(c.x, c.y), izz, nzz), rz = rt_fma2(rt_pair(c.z, c.w), izz, nzz);
    constexpr bool nx = (OCT & 1) != 0, ny = (OCT & 2) != 0, nz = (OCT & 4) != 0;
    const float el = rt_fmax(rt_fmax(rt_fmax(nx ? lx.y : lx.x, ny ? ly.y : ly.x), nz ? lz.y : lz.x), tmin);
    const float xl = rt_fmin(rt_fmin(rt_fmin(nx ? lx.x : lx.y, ny ? ly.x : ly.y), nz ? lz.x : lz.y), s.t);
    const float er = rt_fmax(rt_fmax(rt_fmax(nx ? rx.y : rx.x, ny ? ry.y : ry.x), nz ? rz.y : rz.x), tmin);
    const float xr = rt_fmin(rt_fmin(rt_fmin(nx ? rx.x : rx.y, ny ? ry.x : ry.y), nz ? rz.x : rz.y), s.t);
    hl = el < xl;
    hr = er < xr;
}

// One box, given as per-axis (min, max) plane pairs, for a ray of known direction signs (fast_box's test
// with the entry and exit plane of each slab picked at compile time): 3 packed multiply-adds, 2 max, 2 min
// and a compare.
template <int OCT>
RT_HD bool fast_box_signed(const FastTrav &s, const rt_f2 &bx, const rt_f2 &by, const rt_f2 &bz, float tmin) {
    const rt_f2 tx = rt_fma2(bx, rt_lo2(s.ixy), rt_lo2(s.nxy)), ty = rt_fma2(by, rt_hi2(s.ixy), rt_hi2(s.nxy));
    const rt_f2 tz = rt_fma2(bz, rt_lo2(s.zin), rt_hi2(s.zin));
    constexpr bool nx = (OCT & 1) != 0, ny = (OCT & 2) != 0, nz = (OCT & 4) != 0;
    const float e = rt_fmax(rt_fmax(rt_fmax(nx ? tx.y : tx.x, ny ? ty.y : ty.x), nz ? tz.y : tz.x), tmin);
    const float x = rt_fmin(rt_fmin(rt_fmin(nx ? tx.x : tx.y, ny ? ty.x : ty.y), nz ? tz.x : tz.y), s.t);
    return e < x;
}

// Triangle::closest_hit (triangle.rs:66-110) through the triangle's affine map (RtTriXf):
// t from the height row, then beta and gamma of the point at t.  Same acceptance rules:
// t in [tmin, bound] closed, gamma in [0, 1], beta in [0, 1 - gamma]; a ray parallel to the
// plane gives t = inf or NaN and fails the range test, as |det A| < epsilon does there.
//
// uv0 = (u.x, v.x, u.y, v.y), uv1 = (u.z, v.z, u.w, v.w) (RtTriXf::uv), n = the height row.  gamma <= 1
// needs no test of its own: beta >= 0 and beta <= 1 - gamma imply it (1 - gamma is exact near 1).
RT_HD bool fast_triangle(const FastTrav &s, const rt_f4 &uv0, const rt_f4 &uv1, const rt_f4 &n, float tmin, float &t_out) {
    // each row of the map against origin and direction at once: (row . o, row . d); the row's
    // coefficients are wave-uniform (scalar registers), so the broadcast costs nothing
    const rt_f2 hn = rt_fma2(rt_pair(n.x, n.x), s.px, rt_fma2(rt_pair(n.y, n.y), s.py, rt_mul2(rt_pair(n.z, n.z), s.pz)));
    const float t = -(hn.x + n.w) * rt_rcp_approx(hn.y);
    if (!(t >= tmin && t <= s.t)) return false;
    const rt_f2 bu = rt_fma2(rt_pair(uv0.x, uv0.x), s.px, rt_fma2(rt_pair(uv0.z, uv0.z), s.py, rt_mul2(rt_pair(uv1.x, uv1.x), s.pz)));
    const rt_f2 gv = rt_fma2(rt_pair(uv0.y, uv0.y), s.px, rt_fma2(rt_pair(uv0.w, uv0.w), s.py, rt_mul2(rt_pair(uv1.y, uv1.y), s.pz)));
    const float beta = rt_fma(t, bu.y, bu.x) + uv1.z, gamma = rt_fma(t, gv.y, gv.x) + uv1.w;
    if (!(gamma >= 0.0f)) return false;
    if (!(beta >= 0.0f && beta <= 1.0f - gamma)) return false;
    t_out = t;
    return true;
}

// One scene-level sphere / box / triangle payload (RtTPrim::f) against the ray held in `r`
// (already in the primitive's space), fast-mode arithmetic: same acceptance rules as
// sphere.rs:58-84, cuboid.rs:63-73 and triangle.rs:66-110, with the box decided by the culling
// slab test (entry distance = the hit) and one reciprocal per primitive.
RT_HD bool fast_payload_hit(const FastTrav &r, int32_t type, const rt_f4 &f0, const rt_f4 &f1, const rt_f4 &f2,
                            float tmin, float &t_out) {
    if (type == RT_SHAPE_SPHERE) {
        const float ocx = r.px.x - f0.x, ocy = r.py.x - f0.y, ocz = r.pz.x - f0.z;
        const float cd = ocx * r.px.y + ocy * r.py.y + ocz * r.pz.y;
        const float dd = r.px.y * r.px.y + r.py.y * r.py.y + r.pz.y * r.pz.y;
        const float kx = ocy * r.pz.y - ocz * r.py.y, ky = ocz * r.px.y - ocx * r.pz.y, kz = ocx * r.py.y - ocy * r.px.y;
        const float disc = dd * (f0.w * f0.w) - (kx * kx + ky * ky + kz * kz);  // see sphere_f32
        if (!(disc >= 0.0f)) return false;
        const float sq = __builtin_sqrtf(disc), inv = 1.0f / dd;
        const float t1 = (-cd - sq) * inv, t2 = (-cd + sq) * inv;
        const bool in1 = tmin <= t1 && t1 < r.t, in2 = tmin <= t2 && t2 < r.t;
        t_out = in1 ? t1 : t2;  // t1 <= t2
        return in1 || in2;
    }
    if (type == RT_SHAPE_BOX) return fast_box(r, f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, tmin, t_out);
    if (type == RT_SHAPE_TRIANGLE) {
        const float abx = f0.w, aby = f1.x, abz = f1.y, acx = f1.z, acy = f1.w, acz = f2.x;
        // c = ac x d ; det A = ab . c ; q = ab x ao ; see triangle_f32 for the determinants
        const float cx = acy * r.pz.y - acz * r.py.y, cy = acz * r.px.y - acx * r.pz.y, cz = acx * r.py.y - acy * r.px.y;
        const float det_a = abx * cx + aby * cy + abz * cz;
        if (__builtin_fabsf(det_a) < 2.220446049250313e-16f) return false;
        const float inv_det = 1.0f / det_a;
        const float aox = f0.x - r.px.x, aoy = f0.y - r.py.x, aoz = f0.z - r.pz.x;
        const float qx = aby * aoz - abz * aoy, qy = abz * aox - abx * aoz, qz = abx * aoy - aby * aox;
        const float t = -(acx * qx + acy * qy + acz * qz) * inv_det;
        if (!(t >= tmin && t <= r.t)) return false;
        const float gamma = (r.px.y * qx + r.py.y * qy + r.pz.y * qz) * inv_det;
        if (!(gamma >= 0.0f && gamma <= 1.0f)) return false;
        const float beta = (aox * cx + aoy * cy + aoz * cz) * inv_det;
        if (!(beta >= 0.0f && beta <= 1.0f - gamma)) return false;
        t_out = t;
        return true;
    }
    return false;
}

// The ray of `w` (world space) in the space of scene-level primitive `slot`: through the
// instance's inverse transform (instance.rs:77-80) if it is one, unchanged otherwise.  The
// search bound and origin primitive are carried over.
RT_HD void fast_local_ray(const RtSceneView &S, uint32_t slot, bool is_inst, float wox, float woy, float woz, float wdx,
                          float wdy, float wdz, const FastTrav &w, FastTrav &out) {
    if (is_inst) {
        const rt_f4 *m = S.tprim_xf + (size_t)slot * 3;
        fast_set_ray_in_instance(out, m[0], m[1], m[2], rt_pair(wox, wdx), rt_pair(woy, wdy), rt_pair(woz, wdz));
    } else {
        fast_set_ray(out, wox, woy, woz, wdx, wdy, wdz);
    }
    out.t = w.t;
}

// One outer iteration: walk inner nodes until a leaf (or the end), then process that leaf.
// Returns false when the ray is finished (stack empty, any-hit satisfied, or overflow flagged).
template <class Stack, bool ANYHIT>
RT_HD bool fast_step(const RtSceneView &S, FastTrav &s, Stack &stk, float tmin, int max_sp, bool &overflow) {
    const char *node_base = reinterpret_cast<const char *>(S.nodes);
    int32_t cur = s.cur;
    while (cur >= 0) {
        RT_TALLY(node);
        RT_VISIT(0, cur);
        const rt_f4 *n = reinterpret_cast<const rt_f4 *>(node_base + (uint32_t)cur);
        // a = left (min.x, max.x, min.y, max.y), b = right likewise, c = z planes (l min, l max, r min, r max)
        const rt_f4 a = n[0], b = n[1], c = n[2], l = n[3];
        const int32_t lref = rt_as_int(l.x), rref = rt_as_int(l.y);
        float el, er;
        bool hl, hr;
        fast_box2(s, a, b, c, tmin, el, er, hl, hr);
        hr = hr && rref != RT_REF_NONE;
        if (hl && hr) {
            const bool swap = er < el;
            if (s.sp >= max_sp) { overflow = true; return false; }
            fast_push(s, stk, swap ? lref : rref);
            cur = swap ? rref : lref;
        } else if (hl) {
            cur = lref;
        } else if (hr) {
            cur = rref;
        } else {
            cur = fast_pop(s, stk);
        }
    }
    if (cur == RT_REF_NONE) { s.cur = cur; return false; }
    bool entered = false;
    if (cur == RT_REF_EXIT_BLAS) {
        fast_set_ray(s, stk.get_aux(0), stk.get_aux(1), stk.get_aux(2), stk.get_aux(3), stk.get_aux(4), stk.get_aux(5));
        s.cur_shape = -1;
    } else {
        const uint32_t v = (uint32_t)~cur;
        const uint32_t cnt = v & 7u, first = v >> 3;
        if (s.cur_shape >= 0) {
            const char *tri_base = reinterpret_cast<const char *>(S.tri_xf);
            for (uint32_t k = 0; k < cnt; ++k) {
                const int32_t gtri = (int32_t)(first + k);
                if (s.cur_shape == s.org_shape && gtri == s.org_tri) continue;
                RT_TALLY(tri);
                RT_VISIT(1, gtri);
                const rt_f4 *tp = reinterpret_cast<const rt_f4 *>(tri_base + (uint32_t)gtri * 48u);
                float t;
                if (fast_triangle(s, tp[0], tp[1], tp[2], tmin, t)) {
                    s.t = t;
                    s.shape = s.cur_shape;
                    s.tri = gtri;
                    if (ANYHIT) { s.cur = RT_REF_NONE; return false; }
                }
            }
        } else {
            for (uint32_t k = 0; k < cnt && !entered; ++k) {
                const rt_f4 *pp = S.tprims + (size_t)(first + k) * 4;
                const rt_f4 h = pp[0];
                const int32_t type = rt_as_int(h.x), shape_id = rt_as_int(h.y), blas_root = rt_as_int(h.z),
                              proto_type = rt_as_int(h.w);
                const bool is_inst = type == RT_SHAPE_INSTANCE;
                RT_TALLY(tprim);
                if (type == RT_SHAPE_MESH || (is_inst && proto_type == RT_SHAPE_MESH)) {
                    if (is_inst) {
                        // skip the transform when the ray misses the instance's world box (the reference
                        // has no such test; it cannot change a hit, only save work)
                        const rt_f4 f0 = pp[1], f1 = pp[2];
                        float e;
                        if (!fast_box(s, f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, tmin, e)) continue;
                    }
                    if (s.sp + 2 > max_sp) { overflow = true; return false; }
                    if (k + 1 < cnt) fast_push(s, stk, rt_make_leaf_ref(first + k + 1, cnt - k - 1));
                    fast_push(s, stk, RT_REF_EXIT_BLAS);
                    if (is_inst) {
                        const rt_f4 *m = S.tprim_xf + (size_t)(first + k) * 3;
                        const rt_f2 wx = s.px, wy = s.py, wz = s.pz;
                        fast_set_ray_in_instance(s, m[0], m[1], m[2], wx, wy, wz);
                    }
                    s.cur_shape = shape_id;
                    cur = blas_root;
                    entered = true;
                } else if (shape_id != s.org_shape) {
                    // sphere / box / triangle at scene level, possibly instanced
                    FastTrav lr;
                    fast_local_ray(S, first + k, is_inst, s.px.x, s.py.x, s.pz.x, s.px.y, s.py.y, s.pz.y, s, lr);
                    float t;
                    if (fast_payload_hit(lr, is_inst ? proto_type : type, pp[1], pp[2], pp[3], tmin, t)) {
                        s.t = t;
                        s.shape = shape_id;
                        s.tri = -1;
                        if (ANYHIT) { s.cur = RT_REF_NONE; return false; }
                    }
                }
            }
        }
    }
    if (!entered) cur = fast_pop(s, stk);
    s.cur = cur;
    return cur != RT_REF_NONE;
}

// The f64 test of a secondary ray against the primitive it starts on, with the
// reference's own arithmetic.  A triangle can never be re-hit from a point in its
// own plane (t = 0 < t_min), so only spheres and boxes are tested.
//
// `outward`: the ray leaves the surface on the side of the normal.  A sphere's hit point lies
// on the sphere, so an outward ray cannot meet it again (its second root is rounding noise
// around t = 0, far below t_min) and the test is skipped.  A box is always tested: for a ray
// that started inside it the reference reports the "hit" at t_min, a point INSIDE the box
// (cuboid.rs:63-73 with bbox.rs:124), from which every direction hits the box again.
//
// `box_leaves`: the caller has found the ray to leave the surface by a clear margin (leaves_clearly below)
// and the mode allows conclusions in place of arithmetic.  The hit point lies on a face of the box to within
// a few units in the last place (2e-15 for coordinates of order 10), the box is convex, so such a ray is out of
// it again at t <= 2e-15 / (n . d): below t_min = 1e-6 whenever the cosine exceeds 2e-9, and the reference's
// slab test -- two divisions per axis, once per light -- would report no hit.
RT_HD bool self_hit_f64(const RtSceneView &S, int32_t org_shape, const D3 &o, const D3 &d, double tmin, double tmax,
                        bool outward, bool box_leaves, double &t_out) {
    if (org_shape < 0) return false;
    const RtShape *sh = &S.shapes[org_shape];
    const int32_t pt = sh->type == RT_SHAPE_INSTANCE ? S.protos[sh->proto].type : sh->type;
    if (pt != RT_SHAPE_SPHERE && pt != RT_SHAPE_BOX) return false;
    if (pt == RT_SHAPE_SPHERE && outward) return false;
    if (pt == RT_SHAPE_BOX && box_leaves) return false;
    return exact_prim_hit(S, org_shape, -1, o, d, tmin, tmax, false, 0.0, t_out, nullptr);
}

// The ray from a surface point along `d` leaves on the side of the unit normal `n` with a cosine above 1e-6.
RT_HD bool leaves_clearly(const D3 &n, const D3 &d) {
    const double nd = dot(n, d);
    return nd > 0.0 && nd * nd > 1e-12 * dot(d, d);
}

// ---------------------------------------------------------------- camera

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11): ten rounds of two 32x32 -> 64 bit multiplications, key bumped by the Weyl constants
// between rounds.
RT_HD void rt_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int round = 0; round < 10; ++round) {
        const uint64_t m0 = (uint64_t)0xD2511F53u * c0, m1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(m1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(m0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)m1;
        c3 = (uint32_t)m0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 53 random bits -> [0, 1), like rand's Standard distribution for f64
RT_HD double rt_unit_double(uint32_t hi, uint32_t lo) {
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

// antialias.rs:26-57: offsets of sample (p, q) of pixel (i, j) inside the pixel.  `sample` is the
// frame-wide sample index, which is the generator's counter for the two stochastic methods.
RT_HD void sample_offsets(int32_t method, uint64_t seed, uint64_t sample, uint32_t p, uint32_t q, uint32_t n, double &di,
                          double &dj) {
    if (method == 0) {  // normal, antialias.rs:38-43
        di = ((double)p + 0.5) / (double)n;
        dj = ((double)q + 0.5) / (double)n;
        return;
    }
    uint32_t x[4];
    rt_philox4x32_10((uint32_t)sample, (uint32_t)(sample >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
    const double r1 = rt_unit_double(x[0], x[1]), r2 = rt_unit_double(x[2], x[3]);
    if (method == 1) {  // jittered, antialias.rs:45-50
        di = ((double)p + r1) / (double)n;
        dj = ((double)q + r2) / (double)n;
    } else {  // random, antialias.rs:52-57
        di = r1;
        dj = r2;
    }
}

// camera/mod.rs:52-56 + perspective.rs:20-28 / orthographic.rs:19-26, in two steps: the sample's point
// (u, v) on the image plane, and the ray through it.
RT_HD void camera_plane_point(const RtCamera &cam, uint32_t i, uint32_t j, double di, double dj, double &u, double &v) {
    u = cam.left + (cam.right - cam.left) * ((double)i + di) / (double)cam.pixels_x;
    v = cam.bottom + (cam.top - cam.bottom) * ((double)j + dj) / (double)cam.pixels_y;
}
RT_HD void camera_ray_through(const RtCamera &cam, double u, double v, D3 &o, D3 &d) {
    if (cam.type == RT_CAMERA_PERSPECTIVE) {
        o = d3(cam.position);
        d = (d3(cam.u) * u + d3(cam.v) * v) - d3(cam.w) * cam.focal_length;
    } else {
        o = d3(cam.position) + D3{u, v, 0.0};
        d = D3{0.0, 0.0, -1.0};
    }
}
RT_HD void camera_ray(const RtCamera &cam, uint32_t i, uint32_t j, double di, double dj, D3 &o, D3 &d) {
    double u, v;
    camera_plane_point(cam, i, j, di, dj, u, v);
    camera_ray_through(cam, u, v, o, d);
}

// ---------------------------------------------------------------- shading

struct ColorF {
    float r, g, b;
};

// 1/sqrt(x) and x^p (x >= 0) for the fast mode's light sums: the hardware's reciprocal square root and
// base-2 logarithm / exponential (one unit in the last place each), where the correctly rounded sqrt,
// division and powf cost 20, 20 and over 100 instructions.  x^p = 2^(p log2 x) is then off by about
// |p log2 x| * 1e-7, relatively: 1e-6 for a highlight that is still visible, in a colour that is quantised
// to 1 / 255.  x = 0 gives 2^-inf = 0; p = 0 gives 1 for x > 0 and, by the explicit test, for x = 0 as powf does.
RT_HD float rt_rsqrt_approx(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / __builtin_sqrtf(x);
#endif
}
RT_HD float rt_pow_approx(float x, float p) {
    if (p == 0.0f) return 1.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x));
#else
    return __builtin_exp2f(p * __builtin_log2f(x));
#endif
}

// One light's contribution (lambertian.rs:29-34, blinn_phong.rs:35-45).
// `stl` is the un-normalised surface-to-light vector the light returned.
//
// F32 = false: the reference's arithmetic (f64 normalisations and dot products, the cosine rounded to
// f32 where the reference rounds it).  F32 = true (fast mode): the same formulas on the f32-rounded
// vectors, with cos = (n . l) / |l| in place of n . (l / |l|), 1 / |.| from rt_rsqrt_approx and the
// highlight's power from rt_pow_approx.  Hit point, normal and directions are still the f64 ones; only
// these few products change, by parts in 1e6 of a colour that is quantised to 8 bits (measured on the
// 1000 x 1000 frames: PSNR against the f64 oracle unchanged to three digits).
template <bool F32>
RT_HD void add_light(const RtShader &sh, const RtLight &light, const D3 &normal, const D3 &stl, const D3 &ray_dir,
                     ColorF &color) {
    float k, s = 0.0f;
    if (F32) {
        const float nx = (float)normal.x, ny = (float)normal.y, nz = (float)normal.z;
        const float lx = (float)stl.x, ly = (float)stl.y, lz = (float)stl.z;
        const float inv_l = rt_rsqrt_approx(lx * lx + ly * ly + lz * lz);
        k = rt_fmax((nx * lx + ny * ly + nz * lz) * inv_l, 0.0f);
        if (sh.type == RT_SHADER_BLINN_PHONG) {
            const float dx = (float)ray_dir.x, dy = (float)ray_dir.y, dz = (float)ray_dir.z;
            const float inv_d = rt_rsqrt_approx(dx * dx + dy * dy + dz * dz);
            const float hx = lx * inv_l - dx * inv_d, hy = ly * inv_l - dy * inv_d, hz = lz * inv_l - dz * inv_d;
            const float inv_h = rt_rsqrt_approx(hx * hx + hy * hy + hz * hz);
            s = rt_pow_approx(rt_fmax((nx * hx + ny * hy + nz * hz) * inv_h, 0.0f), sh.shininess);
        }
    } else {
        const D3 stol_normal = normalize(stl);
        const double cos_incidence = dot(normal, stol_normal);
        k = (float)rt_dmax(cos_incidence, 0.0);
        if (sh.type == RT_SHADER_BLINN_PHONG) {
            const D3 half_vector = normalize((-normalize(ray_dir)) + stol_normal);
            s = __builtin_powf((float)rt_dmax(dot(normal, half_vector), 0.0), sh.shininess);
        }
    }
    color.r += (sh.diffuse[0] * light.intensity[0]) * k;
    color.g += (sh.diffuse[1] * light.intensity[1]) * k;
    color.b += (sh.diffuse[2] * light.intensity[2]) * k;
    if (sh.type == RT_SHADER_BLINN_PHONG) {
        color.r += (sh.specular[0] * light.intensity[0]) * s;
        color.g += (sh.specular[1] * light.intensity[1]) * s;
        color.b += (sh.specular[2] * light.intensity[2]) * s;
    }
}

// perfect_mirror.rs:15-16
RT_HD D3 mirror_direction(const D3 &ray_dir, const D3 &normal) {
    const D3 incoming = normalize(ray_dir);
    return normal * (2.0 * -dot(incoming, normal)) + incoming;
}
