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
//   ref >= 0                inner node: BYTE offset of its record in the node array (index * 64),
//                           which a scalar load takes as its offset operand without arithmetic
//   ref <  0, != NONE       leaf: v = ~ref, count = v & 7 (1..4), first = v >> 3
//                           (first indexes the top-level primitive table, or the
//                           triangle table while inside a mesh)
//   ref == RT_REF_NONE      no child (only in a tree's header node)
//
// Plane order: per box and axis one (min, max) pair.  The packed fp32 multiply-add turns such a
// pair into the ray's two plane distances on that axis in one instruction, and for a ray whose
// direction sign on the axis is known the first or second of the two IS the entry distance: no
// min / max needed to sort them.
struct RtNode {
    float lx[2], ly[2];  // left box: (min.x, max.x), (min.y, max.y)
    float rx[2], ry[2];  // right box
    float lz[2];         // left box (min.z, max.z)
    float rz[2];         // right box (min.z, max.z)
    int32_t left, right;
    // bit o set: a ray of direction octant o (bit k of o = it runs towards smaller coordinates on
    // axis k) comes to the RIGHT child first, i.e. (o >> axis) & 1
    int32_t right_first_mask;
    int32_t axis;  // the axis the two children were split on: the left child holds the smaller centroids
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

// Owner of tile (tx, ty) in a frame split over `parts` devices: (tx + skew * ty) % parts, a
// lattice.  A part's tiles then lie on a regular oblique grid, and how evenly that grid samples
// the image decides how evenly the cost of a frame (which varies smoothly over the image) is
// shared.  rt_tile_skew picks the skew for a tile grid: first the skews that give the parts the
// most equal tile counts, among those the one whose lattice has the longest shortest vector
// (skew 1 would put a part's tiles on diagonals 'parts' tiles apart; for 8 parts skew 3 spaces
// them 2.8 tiles apart in every direction), then the smallest.
RT_TYPES_INL uint32_t rt_tile_owner(uint32_t tx, uint32_t ty, uint32_t parts, uint32_t skew) {
    return parts <= 1 ? 0u : (uint32_t)(((uint64_t)skew * ty + tx) % parts);
}

inline uint32_t rt_tile_skew(uint32_t tiles_x, uint32_t tiles_y, uint32_t parts) {
    if (parts <= 1) return 0;
    static thread_local uint32_t memo_key[3] = {0, 0, 0}, memo_val = 0;
    if (memo_key[0] == tiles_x && memo_key[1] == tiles_y && memo_key[2] == parts) return memo_val;
    uint32_t best = 1, best_max = ~0u, best_len = 0;
    for (uint32_t s = 1; s < parts; ++s) {
        // tile counts: every part gets tiles_x / parts per row, the first tiles_x % parts owners of a row one more
        uint32_t most = 0;
        for (uint32_t p = 0; p < parts; ++p) {
            uint32_t extra = 0;
            for (uint32_t ty = 0; ty < tiles_y; ++ty) {
                const uint32_t first = (uint32_t)(((uint64_t)s * ty) % parts);  // owner of tx = 0 (and of tx = q * parts)
                extra += (p + parts - first) % parts < tiles_x % parts ? 1u : 0u;
            }
            most = extra > most ? extra : most;
        }
        // shortest non-zero (a, b) with a + s * b = 0 (mod parts)
        uint32_t len = parts * parts;  // (parts, 0) is always in the lattice
        for (uint32_t b = 0; b <= parts; ++b)
            for (uint32_t a = 0; a < parts; ++a) {  // a and a - parts are the two candidates of smallest |a|
                if ((a + (uint64_t)s * b) % parts != 0 || (a == 0 && b == 0)) continue;
                const uint32_t m = a < parts - a ? a : parts - a;
                const uint32_t l = m * m + b * b;
                len = l < len ? l : len;
            }
        if (most < best_max || (most == best_max && len > best_len)) {
            best = s; best_max = most; best_len = len;
        }
    }
    memo_key[0] = tiles_x; memo_key[1] = tiles_y; memo_key[2] = parts; memo_val = best;
    return best;
}

// The split of a frame over `parts` devices, optionally with a smaller share for part 0 (the device
// that also receives the other parts' tiles and restores the image layout).  `part0_weight` = w in
// 1..7 gives part 0 a share w/8 of what each other part gets: the tiles are dealt to w + 8 * (parts - 1)
// LANES on the lattice above, part 0 owns w of them spaced evenly through the residues (so its tiles
// stay spread over the image), and the remaining lanes go to parts 1, 2, ... in turn, eight each.
// 0 or 8 = equal shares: one lane per part.
struct RtTileMap {
    uint32_t parts, lanes, skew;
    uint32_t lead, step;  // part 0 owns lanes 0, step, 2 * step, ... (lead of them); lead = 0: lane = part
};

inline RtTileMap rt_tile_map(uint32_t tiles_x, uint32_t tiles_y, uint32_t parts, uint32_t part0_weight) {
    RtTileMap m;
    m.parts = parts < 1 ? 1 : parts;
    m.lead = (m.parts >= 2 && part0_weight >= 1 && part0_weight <= 7) ? part0_weight : 0;
    m.lanes = m.lead ? m.lead + 8u * (m.parts - 1u) : m.parts;
    m.step = m.lead ? m.lanes / m.lead : 1u;
    m.skew = rt_tile_skew(tiles_x, tiles_y, m.lanes);
    return m;
}

inline uint32_t rt_tile_map_owner(const RtTileMap &m, uint32_t tx, uint32_t ty) {
    const uint32_t lane = rt_tile_owner(tx, ty, m.lanes, m.skew);
    if (!m.lead) return lane;
    const uint32_t q = lane / m.step;
    if (lane % m.step == 0 && q < m.lead) return 0;
    const uint32_t part0_lanes_below = q + 1 < m.lead ? q + 1 : m.lead;  // of lanes 0 .. lane
    return 1u + (lane - part0_lanes_below) % (m.parts - 1u);
}

// Division of a dividend below 2^31 by a divisor fixed on the host, as one 32 x 32 -> 64 bit
// multiplication and a shift.  The index arithmetic of a sample (sample -> pixel -> tile) divides
// by five such numbers; a hardware integer division costs a wave about 25 instructions each, in
// kernels whose remaining work per sample is a few hundred.
// With l = ceil(log2 d) and m = floor(2^(31 + l) / d) + 1 (which fits 32 bits), floor(n * m / 2^(31 + l))
// = floor(n / d) for every n < 2^31 (Granlund & Montgomery, "Division by invariant integers using
// multiplication", PLDI'94, theorem 4.2 with N = 31).
struct RtDiv {
    uint32_t d, m, sh;
};

inline RtDiv rt_div_make(uint32_t d) {
    RtDiv v;
    v.d = d < 1 ? 1 : d;
    uint32_t l = 0;
    while ((1ull << l) < v.d) ++l;
    v.sh = 31 + l;
    v.m = (uint32_t)((1ull << v.sh) / v.d + 1ull);
    return v;
}

RT_TYPES_INL uint32_t rt_div(uint32_t n, const RtDiv &v) { return (uint32_t)(((uint64_t)n * v.m) >> v.sh); }
RT_TYPES_INL uint32_t rt_mod(uint32_t n, const RtDiv &v) { return n - rt_div(n, v) * v.d; }

// Pixel block of the in-tile order (RtBatch::blk_*_log2): the power-of-two block, as square as
// possible (wider than high if not), that holds about 64 samples and divides the tile.
inline void rt_pixel_block(uint32_t tile_w, uint32_t tile_h, uint32_t samples_per_pixel, uint32_t &w_log2, uint32_t &h_log2) {
    uint32_t px_log2 = 0;
    while ((2u << px_log2) * samples_per_pixel <= 96u) ++px_log2;  // 9 samples per pixel: 8 pixels, 4 x 2
    w_log2 = (px_log2 + 1) / 2;
    h_log2 = px_log2 / 2;
    while (w_log2 > 0 && tile_w % (1u << w_log2) != 0) --w_log2;
    while (h_log2 > 0 && tile_h % (1u << h_log2) != 0) --h_log2;
}

// Top-level primitive, 64 bytes, stored in leaf order of the scene-level tree.
//   SPHERE    f[0..2] center, f[3] radius
//   BOX       f[0..2] min, f[3..5] max
//   TRIANGLE  f[0..2] a, f[3..5] a-b, f[6..8] a-c
//   MESH      blas_root = header node of the mesh's tree, tri_base = first triangle
//   INSTANCE  proto_type + the prototype's payload as above (object space); of a mesh: f[0..2] min, f[3..5] max of
//             the instance's world box, f[6..11] the same box as (min, max) pairs per axis
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
// The beta and gamma rows are stored interleaved, (u.x, v.x), (u.y, v.y), (u.z, v.z), (u.w, v.w): each
// pair is one operand of a packed multiply-add that advances both barycentrics at once.
struct RtTriXf {
    float uv[8];
    float n[4];
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
