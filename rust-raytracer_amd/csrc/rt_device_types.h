// rt_device_types.h -- fp32 traversal records as they lie in HBM, shared by the
// host flattening code and the kernels.
#pragma once

#include <stdint.h>

// A BVH node holds the boxes of BOTH its children, so one 64-byte fetch decides
// which of them to enter.  The tree itself (which shapes go left and right at
// every level) is the reference's (geometry/bvh.rs:36-87); only the storage
// differs from its Box<BVHNode> pointers.
//
// Child reference encoding:
//   ref >= 0                inner node index into the node array
//   ref <  0, != NONE       leaf: v = ~ref, count = v & 7 (1..4), first = v >> 3
//                           (first indexes the top-level primitive table, or the
//                           triangle table while inside a mesh)
//   ref == RT_REF_NONE      no child (only in a tree's header node)
//
// Plane order: the x and y planes of both boxes come first, in (min.x, min.y) / (max.x, max.y)
// pairs, the four z planes last.  A ray then needs its (1/d.x, 1/d.y) pair for the first four
// register pairs of the record and 1/d.z for the last two, which is the operand shape of the
// packed fp32 multiply-add (two planes per instruction).
struct RtNode {
    float lmin_xy[2], lmax_xy[2];
    float rmin_xy[2], rmax_xy[2];
    float lz[2];  // left box min.z, max.z
    float rz[2];  // right box min.z, max.z
    int32_t left, right;
    int32_t axis;  // the axis the two children were split on: the left child holds the smaller centroids
    int32_t _pad;
};
static_assert(sizeof(RtNode) == 64, "RtNode must be 64 bytes");

#define RT_REF_NONE ((int32_t)0x80000000)
#define RT_REF_EXIT_BLAS ((int32_t)0x80000001) /* traversal-stack marker, never stored in a node */

#if defined(__HIPCC__)
#define RT_TYPES_INL __host__ __device__ inline
#else
#define RT_TYPES_INL inline
#endif

RT_TYPES_INL int32_t rt_make_leaf_ref(uint32_t first, uint32_t count) {
    return ~(int32_t)((first << 3) | count);
}

// Owner of tile k = ty * tiles_x + tx in a frame split over `parts` devices.  Tiles are dealt
// round-robin in row-major order, so part sizes differ by at most one tile; when a tile row holds
// a whole number of rounds the deal is shifted by one per row, which keeps a part from owning
// whole columns of the image.
RT_TYPES_INL uint32_t rt_tile_owner(uint32_t k, uint32_t ty, uint32_t tiles_x, uint32_t parts) {
    if (parts <= 1) return 0;
    return (k + (tiles_x % parts == 0 ? ty : 0u)) % parts;
}

// Top-level primitive, 64 bytes, stored in leaf order of the scene-level tree.
//   SPHERE    f[0..2] center, f[3] radius
//   BOX       f[0..2] min, f[3..5] max
//   TRIANGLE  f[0..2] a, f[3..5] a-b, f[6..8] a-c
//   MESH      blas_root = header node of the mesh's tree, tri_base = first triangle
//   INSTANCE  proto_type + the prototype's payload as above (object space)
struct RtTPrim {
    int32_t type;       // RT_SHAPE_*
    int32_t shape_id;   // index into RtScene.shapes
    int32_t blas_root;  // MESH / INSTANCE of a mesh
    int32_t proto_type; // INSTANCE: RT_SHAPE_* of the prototype
    float f[12];
};
static_assert(sizeof(RtTPrim) == 64, "RtTPrim must be 64 bytes");

// The same triangle as an affine map, 48 bytes, same order as RtTri: for a point p,
//   beta(p)  = u . (p, 1)     gamma(p) = v . (p, 1)     height(p) = n . (p, 1)
// are its coordinates in the frame (b - a, c - a, (b - a) x (c - a)) at a, i.e. the solution of
// triangle.rs:76-105's linear system without the per-ray determinants.  A degenerate
// triangle (the reference's det A == 0, never hit) stores NaN rows.
struct RtTriXf {
    float u[4], v[4], n[4];
};
static_assert(sizeof(RtTriXf) == 48, "RtTriXf must be 48 bytes");

// Mesh triangle, 48 bytes, stored in leaf order of its mesh's tree.
struct RtTri {
    float a[3];
    float ab[3];  // a - b
    float ac[3];  // a - c
    int32_t orig; // index of this triangle in the mesh's file order
    int32_t mesh; // index into RtScene.meshes
    int32_t _pad;
};
static_assert(sizeof(RtTri) == 48, "RtTri must be 48 bytes");
