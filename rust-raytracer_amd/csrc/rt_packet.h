// rt_packet.h -- wave-packet traversal for coherent rays (gfx950 device code only).
//
// The 64 rays of a wavefront walk the tree TOGETHER: the current node, the stack and the
// position inside a leaf are wave-uniform (scalar registers, one small LDS stack per wave),
// node and triangle records are fetched once per wave through the scalar cache instead of once
// per lane through the vector memory pipeline, and every branch is a scalar branch.  A lane
// only owns its ray, its closest hit so far and its vote on where to go next:
//
//   * a child is entered if ANY lane's ray enters its box (wave ballot); the child on the
//     side most of the wave's rays come from (the node's right_first_mask against the direction
//     octant of the wave) is visited first, the other one is pushed;
//   * every lane tests every primitive of a visited leaf; a lane whose ray does not reach the
//     leaf's box simply finds no hit there;
//   * an any-hit lane that has found its blocker sets its search bound to -inf, so all its
//     further box and primitive tests fail, and the wave stops when no lane is left.
//
// This is exactly the per-lane traversal's result (closest hit is order independent, box
// tests only cull); what changes is the cost.  Camera rays and shadow rays of neighbouring
// samples touch nearly the same nodes (measured on the 1000x1000 bunny frame: a wave's union
// is 14 nodes for camera rays against 12.6 for its longest lane, 21..25 against 17..21 for
// shadow rays), so the packet executes about as many steps as its slowest lane did alone,
// without the divergence, the per-lane stack traffic and 4 KiB of vector loads per step.
// Incoherent rays (mirror bounces) keep the per-lane kernel.
#pragma once

#include "rt_core.h"

// Uniform stack of one wave: `rows` dwords of LDS.  The top entry lives in a register (every
// lane holds the same value), so a pop starts the LDS read of the entry below without waiting.
struct WaveStack {
    int32_t *base;
    int32_t sp;
    int32_t high;  // most entries ever held: checked against the rows once, at the end
    int32_t tos;   // same in all lanes
    __device__ __forceinline__ void init(int32_t *b) { base = b; sp = 0; high = 0; tos = RT_REF_NONE; }
    __device__ __forceinline__ void push(int32_t ref) {
        base[sp] = tos;  // all lanes store the same value to the same address
        sp += 1;
        high = high > sp ? high : sp;
        tos = ref;
    }
    // Row 0 holds the RT_REF_NONE moved there by the first push; popping an empty stack
    // returns RT_REF_NONE (end of traversal).
    __device__ __forceinline__ int32_t pop() {
        const int32_t r = __builtin_amdgcn_readfirstlane(tos);
        sp = sp > 0 ? sp - 1 : 0;
        tos = base[sp];
        return r;
    }
};

// Direction signs of the wave's live rays.  Bits 0..2: bit k set if most of them run towards smaller
// coordinates along axis k (such a wave reaches a node's RIGHT child first: the builder puts the
// smaller centroids left).  Bit 3: the rays do NOT all agree on all three axes.  "Runs towards
// smaller coordinates" is read off the clamped 1/d the box test multiplies with, so that it is
// also defined for d = +-0.
// `live`: the ballot of the lanes that hold a ray (taken once per traversal; combining it with the sign tests
// as lane masks in scalar registers keeps each test at one vector compare).
__device__ __forceinline__ uint32_t packet_direction_bits(const FastTrav &s, unsigned long long live) {
    const int n = __popcll(live);
    const int nx = __popcll(__ballot(s.ixy.x < 0.0f) & live);
    const int ny = __popcll(__ballot(s.ixy.y < 0.0f) & live);
    const int nz = __popcll(__ballot(s.zin.x < 0.0f) & live);
    const int half = n >> 1;
    const bool mixed = (nx != 0 && nx != n) || (ny != 0 && ny != n) || (nz != 0 && nz != n);
    return (nx > half ? 1u : 0u) | (ny > half ? 2u : 0u) | (nz > half ? 4u : 0u) | (mixed ? 8u : 0u);
}

// Record fetches of the walk, spelled out as scalar loads.  The address is the same in all lanes and
// nothing writes the records, so a scalar load is what the compiler emits too -- as long as the walk
// is all there is to the kernel.  Inside a loop over lights, with other memory traffic around it, it
// turns part of every record fetch (the first quarter of a node, the whole triangle) into per-lane
// vector loads of one and the same address, to save itself register copies; the walk is then 2-3x
// slower.  Each helper issues the loads of one record and waits for them (the box or triangle test
// that follows needs them at once anyway).
#if defined(__HIP_DEVICE_COMPILE__)
typedef float rt_f8v __attribute__((ext_vector_type(8)));
typedef float rt_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rt_f4 rt_f4_of(rt_f4v v) { return rt_f4{v[0], v[1], v[2], v[3]}; }

// 64-byte node record at base + off
__device__ __forceinline__ void rt_fetch_node(const void *base, uint32_t off, rt_f4 &a, rt_f4 &b, rt_f4 &c, rt_f4 &l) {
    rt_f8v lo, hi;
    asm("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx8 %1, %2, %3 offset:0x20\n\ts_waitcnt lgkmcnt(0)"
        : "=&s"(lo), "=&s"(hi)
        : "s"(base), "s"(off));
    a = rt_f4{lo[0], lo[1], lo[2], lo[3]};
    b = rt_f4{lo[4], lo[5], lo[6], lo[7]};
    c = rt_f4{hi[0], hi[1], hi[2], hi[3]};
    l = rt_f4{hi[4], hi[5], hi[6], hi[7]};
}

// 48-byte record (RtTriXf) at base + off
__device__ __forceinline__ void rt_fetch_48(const void *base, uint32_t off, rt_f4 &r0, rt_f4 &r1, rt_f4 &r2) {
    rt_f8v lo;
    rt_f4v hi;
    asm("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %3 offset:0x20\n\ts_waitcnt lgkmcnt(0)"
        : "=&s"(lo), "=&s"(hi)
        : "s"(base), "s"(off));
    r0 = rt_f4{lo[0], lo[1], lo[2], lo[3]};
    r1 = rt_f4{lo[4], lo[5], lo[6], lo[7]};
    r2 = rt_f4_of(hi);
}
#else  // the host pass of the compiler only parses these
__device__ inline void rt_fetch_node(const void *base, uint32_t off, rt_f4 &a, rt_f4 &b, rt_f4 &c, rt_f4 &l) {
    const rt_f4 *n = reinterpret_cast<const rt_f4 *>(reinterpret_cast<const char *>(base) + off);
    a = n[0]; b = n[1]; c = n[2]; l = n[3];
}
__device__ inline void rt_fetch_48(const void *base, uint32_t off, rt_f4 &r0, rt_f4 &r1, rt_f4 &r2) {
    const rt_f4 *n = reinterpret_cast<const rt_f4 *>(reinterpret_cast<const char *>(base) + off);
    r0 = n[0]; r1 = n[1]; r2 = n[2];
}
#endif

// Inner nodes until a leaf (or the end) comes up.  OCT 0..7: all live rays of the wave have these
// direction signs (bit k = negative on axis k), which turns the twelve min / max that sort each
// slab's two plane distances into register choices made at compile time; OCT 8: mixed signs, the
// general test.
// Any-hit packets (shadow rays) visit the left child first instead of the nearer one: which blocker a
// ray finds first does not matter to "blocked or not", most shadow rays of a frame are not blocked at
// all and visit the same nodes in any order, and the walk saves the order test (bunny frame: shadow
// kernel -5 %).  Closest-hit packets keep the nearer child first: an early hit shortens every later test.
#ifndef RT_ANYHIT_ORDERED
#define RT_ANYHIT_ORDERED 0
#endif
template <int OCT, int R, bool ORDERED>
__device__ __forceinline__ int32_t packet_walk(const RtSceneView &S, const FastTrav (&s)[R], float tmin, WaveStack &stk, int32_t cur,
                                               uint32_t dir_bits) {
    const char *node_base = reinterpret_cast<const char *>(S.nodes);
    while (cur >= 0) {
        rt_f4 a, b, c, l;
        rt_fetch_node(node_base, (uint32_t)cur, a, b, c, l);
        bool hl = false, hr = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            bool l_in, r_in;  // a header's absent right child has an empty box
            if (OCT < 8) {
                fast_box2_signed<OCT & 7>(s[r], a, b, c, tmin, l_in, r_in);
            } else {
                float el, er;
                fast_box2(s[r], a, b, c, tmin, el, er, l_in, r_in);
            }
            hl = hl || l_in;
            hr = hr || r_in;
        }
        const unsigned long long ml = __ballot(hl), mr = __ballot(hr);
        const int32_t ref_l = rt_as_int(l.x), ref_r = rt_as_int(l.y);
        const uint32_t order = (uint32_t)rt_as_int(l.z);  // RtNode::right_first_mask
        const bool right_first = ORDERED && ((order >> (OCT < 8 ? (uint32_t)OCT : (dir_bits & 7u))) & 1u) != 0u;
        // two mirrored copies instead of selecting near / far operands: the branch is one scalar
        // instruction, the selects were six
        if (right_first) {
            if (mr != 0ull) {
                if (ml != 0ull) stk.push(ref_l);
                cur = ref_r;
            } else if (ml != 0ull) {
                cur = ref_l;
            } else {
                cur = stk.pop();
            }
        } else {
            if (ml != 0ull) {
                if (mr != 0ull) stk.push(ref_r);
                cur = ref_l;
            } else if (mr != 0ull) {
                cur = ref_r;
            } else {
                cur = stk.pop();
            }
        }
    }
    return cur;
}

// The same loop for one ray per lane and a known octant, written out in gfx950 assembly.  What the
// compiler makes of the C++ above is fine in its vector half (6 packed multiply-adds, 8 min / max, 2
// compares) but spends about 26 scalar instructions and 3 to 4 taken branches per node on the decision
// that follows (uniform conditions widened to lane masks and narrowed again, copies between the arms
// of its structured control flow), as much issue time per wave as the box test itself.  Here a node
// costs 16 vector instructions, 9 to 16 scalar ones and 2 branches, and the broadcast operands of the
// packed multiply-adds are picked with op_sel instead of being held twice (12 registers less).
//
// Registers: s[48:63] the node record (s[48:51] left x / y plane pairs, s[52:55] right, s[56:59] the z
// pairs, s60 / s61 the child references, s62 RtNode::right_first_mask), s[64:67] the two ballots,
// s[68:73] and v[48:61] scratch; all named in the clobber list.  The LDS stack is WaveStack's: row
// `sp` receives the old top on a push, and the register copy of the top is refilled by a read that
// is only waited for at the next record fetch -- or at the way out, the one place where the
// compiler's code could otherwise see it unfinished.
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_WALK_ORDERED(OCT)                                        \
    "s_bitcmp1_b32 s62, " OCT "\n\t"                                \
    "s_cselect_b32 s48, s61, s60\n\t"                               \
    "s_cselect_b32 s49, s60, s61\n\t"                               \
    "s_cselect_b64 s[50:51], s[56:57], s[54:55]\n\t"                \
    "s_cselect_b64 s[52:53], s[54:55], s[56:57]\n\t"
// The order of a node's children as five macro arguments: the instructions that establish it, then the
// registers that hold the first and second child's reference and ballot.  Left first: the record's own
// registers and the compares' results serve as they are.
#define RT_ORDER_OCT(OCT) RT_WALK_ORDERED(OCT), "s48", "s49", "s[50:51]", "s[52:53]"
#define RT_ORDER_LEFT "", "s60", "s61", "s[54:55]", "s[56:57]"
// E_* / X_*: the register of each plane pair that holds the entry / exit distance (left and right box)
#define RT_WALK_ASM_(E_LX, X_LX, E_RX, X_RX, E_LY, X_LY, E_RY, X_RY, E_LZ, X_LZ, E_RZ, X_RZ, ORDER, FIRST, SECOND, FIRST_BAL, SECOND_BAL)                      \
    "1:\n\t"                                                                                                             \
    "s_load_dwordx8 s[48:55], %[nb], %[cur]\n\t"                                                                         \
    "s_load_dwordx8 s[56:63], %[nb], %[cur] offset:0x20\n\t"                                                             \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                           \
    "v_pk_fma_f32 v[48:49], s[48:49], %[ixy], %[nxy] op_sel_hi:[1,0,0]\n\t"                                              \
    "v_pk_fma_f32 v[50:51], s[50:51], %[ixy], %[nxy] op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"                               \
    "v_pk_fma_f32 v[52:53], s[56:57], %[zin], %[zin] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"                               \
    "v_pk_fma_f32 v[54:55], s[52:53], %[ixy], %[nxy] op_sel_hi:[1,0,0]\n\t"                                              \
    "v_pk_fma_f32 v[56:57], s[54:55], %[ixy], %[nxy] op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"                               \
    "v_pk_fma_f32 v[58:59], s[58:59], %[zin], %[zin] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"                               \
    "v_max_f32 v60, " E_LX ", " E_LY "\n\t"                                                                              \
    "v_min_f32 v61, " X_LX ", " X_LY "\n\t"                                                                              \
    "v_max3_f32 v60, v60, " E_LZ ", %[tmin]\n\t"                                                                         \
    "v_min3_f32 v61, v61, " X_LZ ", %[t]\n\t"                                                                            \
    "v_cmp_lt_f32_e64 s[54:55], v60, v61\n\t"                                                                            \
    "v_max_f32 v60, " E_RX ", " E_RY "\n\t"                                                                              \
    "v_min_f32 v61, " X_RX ", " X_RY "\n\t"                                                                              \
    "v_max3_f32 v60, v60, " E_RZ ", %[tmin]\n\t"                                                                         \
    "v_min3_f32 v61, v61, " X_RZ ", %[t]\n\t"                                                                            \
    "v_cmp_lt_f32_e64 s[56:57], v60, v61\n\t"                                                                            \
    ORDER /* FIRST / SECOND = first / second child, FIRST_BAL / SECOND_BAL = their ballots */                            \
    "s_cmp_lg_u64 " FIRST_BAL ", 0\n\t"                                                                                       \
    "s_cbranch_scc0 3f\n\t"                                                                                              \
    "s_mov_b32 %[cur], " FIRST "\n\t"                                                                                          \
    "s_cmp_lg_u64 " SECOND_BAL ", 0\n\t"                                                                                       \
    "s_cbranch_scc0 5f\n\t"                                                                                              \
    "s_lshl2_add_u32 s58, %[sp], %[base]\n\t" /* both: push the second */                                                \
    "v_mov_b32 v60, s58\n\t"                                                                                             \
    "ds_write_b32 v60, %[tos]\n\t"                                                                                       \
    "s_add_i32 %[sp], %[sp], 1\n\t"                                                                                      \
    "s_max_i32 %[high], %[high], %[sp]\n\t"                                                                              \
    "v_mov_b32 %[tos], " SECOND "\n\t"                                                                                          \
    "s_branch 5f\n"                                                                                                      \
    "3:\n\t"                                                                                                             \
    "s_mov_b32 %[cur], " SECOND "\n\t"                                                                                          \
    "s_cmp_lg_u64 " SECOND_BAL ", 0\n\t"                                                                                       \
    "s_cbranch_scc1 5f\n\t"                                                                                              \
    "v_readfirstlane_b32 %[cur], %[tos]\n\t" /* neither: pop */                                                          \
    "s_sub_i32 %[sp], %[sp], 1\n\t"                                                                                      \
    "s_max_i32 %[sp], %[sp], 0\n\t"                                                                                      \
    "s_lshl2_add_u32 s58, %[sp], %[base]\n\t"                                                                            \
    "v_mov_b32 v60, s58\n\t"                                                                                             \
    "ds_read_b32 %[tos], v60\n\t"                                                                                        \
    "s_cmp_lt_i32 %[cur], 0\n\t"                                                                                         \
    "s_cbranch_scc0 1b\n\t"                                                                                              \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                           \
    "s_branch 6f\n"                                                                                                      \
    "5:\n\t"                                                                                                             \
    "s_cmp_lt_i32 %[cur], 0\n\t"                                                                                         \
    "s_cbranch_scc0 1b\n"                                                                                                \
    "6:\n\t"
#define RT_WALK_ASM(...) RT_WALK_ASM_(__VA_ARGS__)
// per axis: entry / exit register of the left box's pair, then of the right box's; a ray that runs towards
// smaller coordinates enters a slab through its max plane, the second element of the pair
#define RT_WALK_X_POS "v48", "v49", "v54", "v55"
#define RT_WALK_X_NEG "v49", "v48", "v55", "v54"
#define RT_WALK_Y_POS "v50", "v51", "v56", "v57"
#define RT_WALK_Y_NEG "v51", "v50", "v57", "v56"
#define RT_WALK_Z_POS "v52", "v53", "v58", "v59"
#define RT_WALK_Z_NEG "v53", "v52", "v59", "v58"
#define RT_WALK_RUN(TEXT)                                                                                                   \
    asm volatile(TEXT                                                                                                       \
                 : [cur] "+s"(cur), [sp] "+s"(sp), [high] "+s"(high), [tos] "+v"(tos)                                       \
                 : [nb] "s"(node_base), [base] "s"(lds_base), [ixy] "v"(s.ixy), [nxy] "v"(s.nxy), [zin] "v"(s.zin),         \
                   [tmin] "s"(tmin_u), [t] "v"(s.t)                                                                         \
                 : "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", \
                   "s63", "v48", "v49", "v50", "v51", \
                   "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "scc", "memory")

template <int OCT, bool ORDERED>
__device__ __forceinline__ int32_t packet_walk_asm(const RtSceneView &S, const FastTrav &s, float tmin, WaveStack &stk, int32_t cur) {
    const void *node_base = S.nodes;
    const uint32_t lds_base = (uint32_t)(uintptr_t)stk.base;  // the low half of an LDS pointer is the LDS address
    const float tmin_u = rt_as_float(__builtin_amdgcn_readfirstlane(rt_as_int(tmin)));
    int32_t sp = __builtin_amdgcn_readfirstlane(stk.sp), high = __builtin_amdgcn_readfirstlane(stk.high), tos = stk.tos;
    cur = __builtin_amdgcn_readfirstlane(cur);
#define RT_WALK_CASE(N, X, Y, Z)                                                       \
    if constexpr (OCT == N) {                                                          \
        if constexpr (ORDERED) RT_WALK_RUN(RT_WALK_ASM(X, Y, Z, RT_ORDER_OCT(#N))); \
        else RT_WALK_RUN(RT_WALK_ASM(X, Y, Z, RT_ORDER_LEFT));                    \
    }
    RT_WALK_CASE(0, RT_WALK_X_POS, RT_WALK_Y_POS, RT_WALK_Z_POS)
    RT_WALK_CASE(1, RT_WALK_X_NEG, RT_WALK_Y_POS, RT_WALK_Z_POS)
    RT_WALK_CASE(2, RT_WALK_X_POS, RT_WALK_Y_NEG, RT_WALK_Z_POS)
    RT_WALK_CASE(3, RT_WALK_X_NEG, RT_WALK_Y_NEG, RT_WALK_Z_POS)
    RT_WALK_CASE(4, RT_WALK_X_POS, RT_WALK_Y_POS, RT_WALK_Z_NEG)
    RT_WALK_CASE(5, RT_WALK_X_NEG, RT_WALK_Y_POS, RT_WALK_Z_NEG)
    RT_WALK_CASE(6, RT_WALK_X_POS, RT_WALK_Y_NEG, RT_WALK_Z_NEG)
    RT_WALK_CASE(7, RT_WALK_X_NEG, RT_WALK_Y_NEG, RT_WALK_Z_NEG)
#undef RT_WALK_CASE
    stk.sp = sp;
    stk.high = high;
    stk.tos = tos;
    return cur;
}
#endif

// One triangle of a leaf against the lane's ray: fetches the triangle's map (RtTriXf, 48 bytes at
// tri_base + at) and returns t, beta and gamma as fast_triangle computes them, operation for operation;
// the caller applies the acceptance rules.  Each row of the map meets the ray as three packed
// multiply-adds over the (origin, direction) pairs, the row's coefficient broadcast from its scalar
// register pair by op_sel: 16 vector instructions, none of them a copy.  (Left to the compiler, the same
// arithmetic is re-paired by its vectoriser into shuffles and, at this register budget, a round trip
// through scratch memory inside the loop.)  The reciprocal is issued early: its result may not feed
// an ordinary vector instruction in the next two issue slots on this chip.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void packet_triangle_asm(const void *tri_base, uint32_t at, const FastTrav &s, float &t, float &beta,
                                                    float &gamma) {
    // s[48:51] = (u.x, v.x, u.y, v.y), s[52:55] = (u.z, v.z, u.w, v.w), s[56:59] = the height row n
    asm("s_load_dwordx8 s[48:55], %[tb], %[at]\n\t"
        "s_load_dwordx4 s[56:59], %[tb], %[at] offset:0x20\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_mul_f32 v[48:49], %[pz], s[58:59] op_sel_hi:[1,0]\n\t"                             // n.z * (oz, dz)
        "v_pk_fma_f32 v[48:49], %[py], s[56:57], v[48:49] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"  // + n.y * (oy, dy)
        "v_pk_fma_f32 v[48:49], %[px], s[56:57], v[48:49] op_sel_hi:[1,0,1]\n\t"                 // + n.x * (ox, dx)
        "v_pk_mul_f32 v[50:51], %[pz], s[52:53] op_sel_hi:[1,0]\n\t"                             // u.z
        "v_rcp_f32 v49, v49\n\t"
        "v_pk_mul_f32 v[52:53], %[pz], s[52:53] op_sel:[0,1] op_sel_hi:[1,1]\n\t"                // v.z
        "v_pk_fma_f32 v[50:51], %[py], s[50:51], v[50:51] op_sel_hi:[1,0,1]\n\t"                 // u.y
        "v_pk_fma_f32 v[52:53], %[py], s[50:51], v[52:53] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"  // v.y
        "v_pk_fma_f32 v[50:51], %[px], s[48:49], v[50:51] op_sel_hi:[1,0,1]\n\t"                 // u.x
        "v_pk_fma_f32 v[52:53], %[px], s[48:49], v[52:53] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"  // v.x
        "v_add_f32 v48, s59, v48\n\t"                                                            // n . o + n.w
        "v_mul_f32_e64 %[t], -v48, v49\n\t"
        "v_fma_f32 %[b], %[t], v51, v50\n\t"
        "v_fma_f32 %[g], %[t], v53, v52\n\t"
        "v_add_f32 %[b], s54, %[b]\n\t"
        "v_add_f32 %[g], s55, %[g]\n\t"
        : [t] "=&v"(t), [b] "=&v"(beta), [g] "=&v"(gamma)
        : [tb] "s"(tri_base), [at] "s"(at), [px] "v"(s.px), [py] "v"(s.py), [pz] "v"(s.pz)
        : "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "v48", "v49", "v50", "v51", "v52",
          "v53");
}
#endif

// A whole mesh tree in one block of assembly: the node walk above, the triangles of every leaf it
// reaches (the arithmetic of packet_triangle_asm, then the acceptance rules of fast_triangle), the pops
// in between, and for any-hit packets the test "is any lane still looking" after each leaf.  It returns
// when it pops the marker that was pushed below the tree's root (RT_REF_EXIT_BLAS), or RT_REF_NONE when an
// any-hit wave is done.  Between two leaves of one mesh the compiler's code is then not involved at all;
// with the leaves in C++ it moved the lane's ray (12 registers) out of and back into the registers the
// walk wants around every leaf, and dispatched on the direction octant again.
//
// Acceptance, cheapest first, each step skipped when no lane is left: t >= tmin; gamma >= 0, beta >= 0,
// t <= bound; beta <= 1 - gamma and "not the triangle the ray starts on".  Lanes that pass take the hit
// under an exec mask.  s60 = index of the triangle, s61 = triangles left in the leaf, s62 = byte offset
// of its record; t, beta, gamma in v54..v56; s[64:69] hold the last two words of the record and the lane
// masks of the tests.  Everything a test reads from the triangle's record is read in its first 14
// instructions; the load of the leaf's next record into the same registers is issued right after them, so
// that its latency passes behind the acceptance tests instead of in front of the next triangle.
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_BLAS_ASM_(E_LX, X_LX, E_RX, X_RX, E_LY, X_LY, E_RY, X_RY, E_LZ, X_LZ, E_RZ, X_RZ, ORDER, FIRST, SECOND, FIRST_BAL, SECOND_BAL, \
                     TAKE_T, LEAF_END)    \
    "s_cmp_lt_i32 %[cur], 0\n\t"                                                                                         \
    "s_cbranch_scc1 7f\n"                                                                                                \
    "1:\n\t"                                                                                                             \
    "s_load_dwordx8 s[48:55], %[nb], %[cur]\n\t"                                                                         \
    "s_load_dwordx8 s[56:63], %[nb], %[cur] offset:0x20\n\t"                                                             \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                           \
    "v_pk_fma_f32 v[48:49], s[48:49], %[ixy], %[nxy] op_sel_hi:[1,0,0]\n\t"                                              \
    "v_pk_fma_f32 v[50:51], s[50:51], %[ixy], %[nxy] op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"                               \
    "v_pk_fma_f32 v[52:53], s[56:57], %[zin], %[zin] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"                               \
    "v_pk_fma_f32 v[54:55], s[52:53], %[ixy], %[nxy] op_sel_hi:[1,0,0]\n\t"                                              \
    "v_pk_fma_f32 v[56:57], s[54:55], %[ixy], %[nxy] op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"                               \
    "v_pk_fma_f32 v[58:59], s[58:59], %[zin], %[zin] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"                               \
    "v_max_f32 v60, " E_LX ", " E_LY "\n\t"                                                                              \
    "v_min_f32 v61, " X_LX ", " X_LY "\n\t"                                                                              \
    "v_max3_f32 v60, v60, " E_LZ ", %[tmin]\n\t"                                                                         \
    "v_min3_f32 v61, v61, " X_LZ ", %[t]\n\t"                                                                            \
    "v_cmp_lt_f32_e64 s[54:55], v60, v61\n\t"                                                                            \
    "v_max_f32 v60, " E_RX ", " E_RY "\n\t"                                                                              \
    "v_min_f32 v61, " X_RX ", " X_RY "\n\t"                                                                              \
    "v_max3_f32 v60, v60, " E_RZ ", %[tmin]\n\t"                                                                         \
    "v_min3_f32 v61, v61, " X_RZ ", %[t]\n\t"                                                                            \
    "v_cmp_lt_f32_e64 s[56:57], v60, v61\n\t"                                                                            \
    ORDER /* FIRST / SECOND = first / second child, FIRST_BAL / SECOND_BAL = their ballots */                            \
    "s_cmp_lg_u64 " FIRST_BAL ", 0\n\t"                                                                                       \
    "s_cbranch_scc0 3f\n\t"                                                                                              \
    "s_mov_b32 %[cur], " FIRST "\n\t"                                                                                          \
    "s_cmp_lg_u64 " SECOND_BAL ", 0\n\t"                                                                                       \
    "s_cbranch_scc0 5f\n\t"                                                                                              \
    "s_lshl2_add_u32 s58, %[sp], %[base]\n\t" /* both: push the second */                                                \
    "v_mov_b32 v60, s58\n\t"                                                                                             \
    "ds_write_b32 v60, %[tos]\n\t"                                                                                       \
    "s_add_i32 %[sp], %[sp], 1\n\t"                                                                                      \
    "s_max_i32 %[high], %[high], %[sp]\n\t"                                                                              \
    "v_mov_b32 %[tos], " SECOND "\n\t"                                                                                          \
    "s_branch 5f\n"                                                                                                      \
    "3:\n\t"                                                                                                             \
    "s_mov_b32 %[cur], " SECOND "\n\t"                                                                                          \
    "s_cmp_lg_u64 " SECOND_BAL ", 0\n\t"                                                                                       \
    "s_cbranch_scc1 5f\n"                                                                                                \
    "4:\n\t" /* pop */                                                                                                   \
    "v_readfirstlane_b32 %[cur], %[tos]\n\t"                                                                             \
    "s_sub_i32 %[sp], %[sp], 1\n\t"                                                                                      \
    "s_max_i32 %[sp], %[sp], 0\n\t"                                                                                      \
    "s_lshl2_add_u32 s58, %[sp], %[base]\n\t"                                                                            \
    "v_mov_b32 v60, s58\n\t"                                                                                             \
    "ds_read_b32 %[tos], v60\n"                                                                                          \
    "5:\n\t"                                                                                                             \
    "s_cmp_lt_i32 %[cur], 0\n\t"                                                                                         \
    "s_cbranch_scc0 1b\n"                                                                                                \
    "7:\n\t" /* a leaf, or one of the two markers (0x80000000, 0x80000001) */                                            \
    "s_cmp_le_i32 %[cur], 0x80000001\n\t"                                                                                \
    "s_cbranch_scc1 9f\n\t"                                                                                              \
    "s_not_b32 s60, %[cur]\n\t"                                                                                          \
    "s_and_b32 s61, s60, 7\n\t"                                                                                          \
    "s_lshr_b32 s60, s60, 3\n\t"                                                                                         \
    "s_mul_i32 s62, s60, 48\n\t"                                                                                        \
    "s_load_dwordx8 s[48:55], %[tb], s62\n\t"                                                                            \
    "s_load_dwordx4 s[56:59], %[tb], s62 offset:0x20\n"                                                                  \
    "8:\n\t"                                                                                                             \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                           \
    "v_pk_mul_f32 v[48:49], %[pz], s[58:59] op_sel_hi:[1,0]\n\t"                                                         \
    "v_pk_fma_f32 v[48:49], %[py], s[56:57], v[48:49] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                              \
    "v_pk_fma_f32 v[48:49], %[px], s[56:57], v[48:49] op_sel_hi:[1,0,1]\n\t"                                             \
    "v_pk_mul_f32 v[50:51], %[pz], s[52:53] op_sel_hi:[1,0]\n\t"                                                         \
    "v_rcp_f32 v49, v49\n\t"                                                                                             \
    "v_pk_mul_f32 v[52:53], %[pz], s[52:53] op_sel:[0,1] op_sel_hi:[1,1]\n\t"                                            \
    "v_pk_fma_f32 v[50:51], %[py], s[50:51], v[50:51] op_sel_hi:[1,0,1]\n\t"                                             \
    "v_pk_fma_f32 v[52:53], %[py], s[50:51], v[52:53] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                              \
    "v_pk_fma_f32 v[50:51], %[px], s[48:49], v[50:51] op_sel_hi:[1,0,1]\n\t"                                             \
    "v_pk_fma_f32 v[52:53], %[px], s[48:49], v[52:53] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                              \
    "v_add_f32 v48, s59, v48\n\t"                                                                                        \
    "v_mul_f32_e64 v54, -v48, v49\n\t"                                                                                   \
    "s_mov_b64 s[64:65], s[54:55]\n\t" /* the record is read: its registers take the leaf's next one meanwhile */       \
    "s_add_i32 s62, s62, 48\n\t"                                                                                         \
    "s_sub_i32 s61, s61, 1\n\t"                                                                                          \
    "s_cmp_lg_u32 s61, 0\n\t"                                                                                            \
    "s_cbranch_scc0 12f\n\t"                                                                                             \
    "s_load_dwordx8 s[48:55], %[tb], s62\n\t"                                                                            \
    "s_load_dwordx4 s[56:59], %[tb], s62 offset:0x20\n"                                                                  \
    "12:\n\t"                                                                                                            \
    "v_cmp_le_f32 vcc, %[tmin], v54\n\t"                                                                                 \
    "s_cbranch_vccz 10f\n\t"                                                                                             \
    "v_fma_f32 v55, v54, v51, v50\n\t"                                                                                   \
    "v_fma_f32 v56, v54, v53, v52\n\t"                                                                                   \
    "v_add_f32 v55, s64, v55\n\t"                                                                                        \
    "v_add_f32 v56, s65, v56\n\t"                                                                                        \
    "v_cmp_le_f32_e64 s[68:69], v54, %[t]\n\t"                                                                           \
    "v_cmp_le_f32_e64 s[64:65], 0, v56\n\t"                                                                              \
    "v_cmp_le_f32_e64 s[66:67], 0, v55\n\t"                                                                              \
    "s_and_b64 vcc, vcc, s[68:69]\n\t"                                                                                   \
    "s_and_b64 s[64:65], s[64:65], s[66:67]\n\t"                                                                         \
    "s_and_b64 vcc, vcc, s[64:65]\n\t"                                                                                   \
    "s_cbranch_vccz 10f\n\t"                                                                                             \
    "v_sub_f32 v56, 1.0, v56\n\t"                                                                                        \
    "v_cmp_ne_u32_e64 s[64:65], s60, %[own]\n\t"                                                                         \
    "v_cmp_le_f32_e64 s[66:67], v55, v56\n\t"                                                                            \
    "s_and_b64 vcc, vcc, s[64:65]\n\t"                                                                                   \
    "s_and_b64 vcc, vcc, s[66:67]\n\t"                                                                                   \
    "s_and_saveexec_b64 s[64:65], vcc\n\t"                                                                               \
    TAKE_T                                                                                                               \
    "v_mov_b32 %[shape], %[cs]\n\t"                                                                                      \
    "v_mov_b32 %[tri], s60\n\t"                                                                                          \
    "s_mov_b64 exec, s[64:65]\n"                                                                                         \
    "10:\n\t"                                                                                                            \
    "s_add_i32 s60, s60, 1\n\t"                                                                                          \
    "s_cmp_lg_u32 s61, 0\n\t"                                                                                            \
    "s_cbranch_scc1 8b\n\t"                                                                                              \
    LEAF_END                                                                                                             \
    "s_branch 4b\n"                                                                                                      \
    "11:\n\t"                                                                                                            \
    "s_brev_b32 %[cur], 1\n" /* RT_REF_NONE */                                                                           \
    "9:\n\t"                                                                                                             \
    "s_waitcnt lgkmcnt(0)\n\t"
#define RT_BLAS_ASM(...) RT_BLAS_ASM_(__VA_ARGS__)
// closest hit: the new bound is the hit's t, and the walk goes on; any hit: the lane is done (bound -inf),
// and so is the wave once no lane has a bound above -inf
#define RT_BLAS_TAKE_CLOSEST "v_mov_b32 %[t], v54\n\t"
#define RT_BLAS_TAKE_ANY "v_mov_b32 %[t], 0xff800000\n\t"
#define RT_BLAS_LEAF_END_CLOSEST ""
#define RT_BLAS_LEAF_END_ANY              \
    "v_cmp_lt_f32 vcc, 0xff800000, %[t]\n\t" \
    "s_cbranch_vccz 11f\n\t"
#define RT_BLAS_RUN(TEXT)                                                                                                   \
    asm volatile(TEXT                                                                                                       \
                 : [cur] "+s"(cur), [sp] "+s"(sp), [high] "+s"(high), [tos] "+v"(tos), [t] "+v"(s.t), [shape] "+v"(s.shape), \
                   [tri] "+v"(s.tri)                                                                                        \
                 : [nb] "s"(node_base), [tb] "s"(tri_base), [base] "s"(lds_base), [ixy] "v"(s.ixy), [nxy] "v"(s.nxy),       \
                   [zin] "v"(s.zin), [px] "v"(s.px), [py] "v"(s.py), [pz] "v"(s.pz), [tmin] "s"(tmin_u), [own] "v"(own_tri), \
                   [cs] "s"(cur_shape)                                                                                      \
                 : "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", \
                   "s63", "s64", "s65", "s66", "s67", "s68", "s69", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
                   "v56", "v57", "v58", "v59", "v60", "v61", "vcc", "scc", "memory")

template <int OCT, bool ANYHIT>
__device__ __forceinline__ int32_t packet_blas_asm(const RtSceneView &S, FastTrav &s, float tmin, WaveStack &stk, int32_t cur,
                                                   int32_t cur_shape, int32_t own_tri) {
    const void *node_base = S.nodes;
    const void *tri_base = S.tri_xf;
    const uint32_t lds_base = (uint32_t)(uintptr_t)stk.base;
    const float tmin_u = rt_as_float(__builtin_amdgcn_readfirstlane(rt_as_int(tmin)));
    int32_t sp = __builtin_amdgcn_readfirstlane(stk.sp), high = __builtin_amdgcn_readfirstlane(stk.high), tos = stk.tos;
    cur = __builtin_amdgcn_readfirstlane(cur);
    cur_shape = __builtin_amdgcn_readfirstlane(cur_shape);
#define RT_BLAS_CASE(N, X, Y, Z)                                                                                   \
    if constexpr (OCT == N) {                                                                                      \
        if constexpr (ANYHIT && !RT_ANYHIT_ORDERED)                                                                \
            RT_BLAS_RUN(RT_BLAS_ASM(X, Y, Z, RT_ORDER_LEFT, RT_BLAS_TAKE_ANY, RT_BLAS_LEAF_END_ANY));         \
        else if constexpr (ANYHIT)                                                                                 \
            RT_BLAS_RUN(RT_BLAS_ASM(X, Y, Z, RT_ORDER_OCT(#N), RT_BLAS_TAKE_ANY, RT_BLAS_LEAF_END_ANY));        \
        else                                                                                                       \
            RT_BLAS_RUN(RT_BLAS_ASM(X, Y, Z, RT_ORDER_OCT(#N), RT_BLAS_TAKE_CLOSEST, RT_BLAS_LEAF_END_CLOSEST)); \
    }
    RT_BLAS_CASE(0, RT_WALK_X_POS, RT_WALK_Y_POS, RT_WALK_Z_POS)
    RT_BLAS_CASE(1, RT_WALK_X_NEG, RT_WALK_Y_POS, RT_WALK_Z_POS)
    RT_BLAS_CASE(2, RT_WALK_X_POS, RT_WALK_Y_NEG, RT_WALK_Z_POS)
    RT_BLAS_CASE(3, RT_WALK_X_NEG, RT_WALK_Y_NEG, RT_WALK_Z_POS)
    RT_BLAS_CASE(4, RT_WALK_X_POS, RT_WALK_Y_POS, RT_WALK_Z_NEG)
    RT_BLAS_CASE(5, RT_WALK_X_NEG, RT_WALK_Y_POS, RT_WALK_Z_NEG)
    RT_BLAS_CASE(6, RT_WALK_X_POS, RT_WALK_Y_NEG, RT_WALK_Z_NEG)
    RT_BLAS_CASE(7, RT_WALK_X_NEG, RT_WALK_Y_NEG, RT_WALK_Z_NEG)
#undef RT_BLAS_CASE
    // The optimiser merges the eight callers' copies of this block's result tuple into one value, and a
    // tuple with per-lane members counts as per-lane throughout: without these the walk's position, stack
    // pointer and everything derived from them would move to vector registers and exec-mask branches.
    stk.sp = __builtin_amdgcn_readfirstlane(sp);
    stk.high = __builtin_amdgcn_readfirstlane(high);
    stk.tos = tos;
    return __builtin_amdgcn_readfirstlane(cur);
}
#endif

// LDS of one wave: the stack's rows (padded to an even number of dwords), then the world-space ray as six
// (x, y) pairs per lane -- px, py, pz, ixy, nxy, zin of FastTrav -- where it waits while the wave is inside
// a mesh; a kernel's own words follow at rt_packet_lds_dwords().
#define RT_PACKET_WORLD_DWORDS (12 * 64)
inline __host__ __device__ uint32_t rt_packet_world_offset(int rows) { return ((uint32_t)rows + 1u) & ~1u; }
inline __host__ __device__ uint32_t rt_packet_lds_dwords(int rows) { return rt_packet_world_offset(rows) + RT_PACKET_WORLD_DWORDS; }

// Traverses the scene with the wave's 64 rays (R = 1: one per lane; more per lane was measured and
// dropped).  On entry s[0] holds the lane's world-space ray, search bound s[0].t (-inf for a slot
// without a ray), s[0].shape / .tri = initial hit, and the origin primitive.  `lds_rows`: max_sp + 1
// dwords for the stack, followed by the world-space ray (see rt_packet_lds_dwords).  Returns false on
// stack overflow.
//
// Two levels, two ray variables.  The outer loop walks the scene-level tree with the world ray and
// never changes it.  A mesh (or instanced mesh) found in a scene-level leaf is traversed then and
// there, start to finish, with a ray of its own in the mesh's space -- packet_blas_asm when all rays
// share a direction octant -- and only the hit (bound, shape, triangle) comes back.  The world ray is
// not kept in registers meanwhile: it is fetched again from LDS afterwards, so a mesh's tree runs with
// one ray's worth of registers.  (An earlier form transformed the ray in place on the way into an
// instance and restored it on the way out, inside one loop over both levels; the compiler then held the
// ray twice and moved it from one set of registers to the other and back around every leaf.)
template <bool ANYHIT, int R>
__device__ __forceinline__ bool packet_traverse(const RtSceneView &S, FastTrav (&s)[R], float tmin, int32_t *lds_rows, int max_sp) {
    static_assert(R == 1, "one ray per lane");
    constexpr bool ORDERED = !ANYHIT || RT_ANYHIT_ORDERED;
    const float ninf = -__builtin_inff();
    rt_f2 *world = reinterpret_cast<rt_f2 *>(lds_rows + rt_packet_world_offset(max_sp + 1)) + (threadIdx.x & 63u);  // [6][64] pairs
    const unsigned long long live = __ballot(s[0].t > ninf);
    world[0 * 64] = s[0].px; world[1 * 64] = s[0].py; world[2 * 64] = s[0].pz;
    world[3 * 64] = s[0].ixy; world[4 * 64] = s[0].nxy; world[5 * 64] = s[0].zin;
    const uint32_t world_bits = packet_direction_bits(s[0], live);
    WaveStack stk;
    stk.init(lds_rows);
    int32_t cur = S.tlas_root;
    for (;;) {
        if (cur >= 0) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_PACKET_WALK_IN_CXX)
            switch (world_bits) {
            case 0: cur = packet_walk_asm<0, ORDERED>(S, s[0], tmin, stk, cur); break;
            case 1: cur = packet_walk_asm<1, ORDERED>(S, s[0], tmin, stk, cur); break;
            case 2: cur = packet_walk_asm<2, ORDERED>(S, s[0], tmin, stk, cur); break;
            case 3: cur = packet_walk_asm<3, ORDERED>(S, s[0], tmin, stk, cur); break;
            case 4: cur = packet_walk_asm<4, ORDERED>(S, s[0], tmin, stk, cur); break;
            case 5: cur = packet_walk_asm<5, ORDERED>(S, s[0], tmin, stk, cur); break;
            case 6: cur = packet_walk_asm<6, ORDERED>(S, s[0], tmin, stk, cur); break;
            case 7: cur = packet_walk_asm<7, ORDERED>(S, s[0], tmin, stk, cur); break;
            default: cur = packet_walk<8, 1, ORDERED>(S, s, tmin, stk, cur, world_bits); break;
            }
#else
            cur = packet_walk<8, 1, ORDERED>(S, s, tmin, stk, cur, world_bits);
#endif
        }
        if (cur == RT_REF_NONE) break;
        // a scene-level leaf
        const uint32_t v = (uint32_t)~cur;
        const uint32_t cnt = v & 7u, first = v >> 3;
        bool done = false;
        for (uint32_t k = 0; k < cnt && !done; ++k) {
            rt_f4 h, pp1, pp2, pp3;  // the whole RtTPrim
            rt_fetch_node(S.tprims, (first + k) * (uint32_t)sizeof(RtTPrim), h, pp1, pp2, pp3);
            const int32_t type = rt_as_int(h.x), shape_id = rt_as_int(h.y), blas_root = rt_as_int(h.z), proto_type = rt_as_int(h.w);
            const bool is_inst = type == RT_SHAPE_INSTANCE;
            if (type == RT_SHAPE_MESH || (is_inst && proto_type == RT_SHAPE_MESH)) {
                if (is_inst) {
                    // enter the instance only if some ray of the wave reaches its world box
                    const rt_f2 bx = rt_pair(pp2.z, pp2.w), by = rt_pair(pp3.x, pp3.y), bz = rt_pair(pp3.z, pp3.w);
                    bool reach;
                    switch (world_bits) {
                    case 0: reach = fast_box_signed<0>(s[0], bx, by, bz, tmin); break;
                    case 1: reach = fast_box_signed<1>(s[0], bx, by, bz, tmin); break;
                    case 2: reach = fast_box_signed<2>(s[0], bx, by, bz, tmin); break;
                    case 3: reach = fast_box_signed<3>(s[0], bx, by, bz, tmin); break;
                    case 4: reach = fast_box_signed<4>(s[0], bx, by, bz, tmin); break;
                    case 5: reach = fast_box_signed<5>(s[0], bx, by, bz, tmin); break;
                    case 6: reach = fast_box_signed<6>(s[0], bx, by, bz, tmin); break;
                    case 7: reach = fast_box_signed<7>(s[0], bx, by, bz, tmin); break;
                    default: {
                        float e;
                        reach = fast_box(s[0], pp1.x, pp1.y, pp1.z, pp1.w, pp2.x, pp2.y, tmin, e);
                    }
                    }
                    if (__ballot(reach) == 0ull) continue;
                }
                // the ray in the mesh's space, with the hit so far
                FastTrav m[1];
                uint32_t bits = world_bits;
                if (is_inst) {
                    rt_f4 r0, r1, r2;
                    rt_fetch_48(S.tprim_xf, (first + k) * 48u, r0, r1, r2);
                    fast_set_ray_in_instance(m[0], r0, r1, r2, s[0].px, s[0].py, s[0].pz);
                    bits = packet_direction_bits(m[0], live);
                } else {
                    m[0].px = s[0].px; m[0].py = s[0].py; m[0].pz = s[0].pz;
                    m[0].ixy = s[0].ixy; m[0].nxy = s[0].nxy; m[0].zin = s[0].zin;
                }
                m[0].t = s[0].t;
                m[0].shape = s[0].shape;
                m[0].tri = s[0].tri;
                const int32_t own = shape_id == s[0].org_shape ? s[0].org_tri : -1;
                stk.push(RT_REF_EXIT_BLAS);
                int32_t c = blas_root;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_PACKET_WALK_IN_CXX)
                if (bits < 8u) {
                    switch (bits) {
                    case 0: c = packet_blas_asm<0, ANYHIT>(S, m[0], tmin, stk, c, shape_id, own); break;
                    case 1: c = packet_blas_asm<1, ANYHIT>(S, m[0], tmin, stk, c, shape_id, own); break;
                    case 2: c = packet_blas_asm<2, ANYHIT>(S, m[0], tmin, stk, c, shape_id, own); break;
                    case 3: c = packet_blas_asm<3, ANYHIT>(S, m[0], tmin, stk, c, shape_id, own); break;
                    case 4: c = packet_blas_asm<4, ANYHIT>(S, m[0], tmin, stk, c, shape_id, own); break;
                    case 5: c = packet_blas_asm<5, ANYHIT>(S, m[0], tmin, stk, c, shape_id, own); break;
                    case 6: c = packet_blas_asm<6, ANYHIT>(S, m[0], tmin, stk, c, shape_id, own); break;
                    default: c = packet_blas_asm<7, ANYHIT>(S, m[0], tmin, stk, c, shape_id, own); break;
                    }
                } else
#endif
                {
                    // rays of mixed direction signs: the general node test, triangles one at a time
                    const char *tri_base = reinterpret_cast<const char *>(S.tri_xf);
                    for (;;) {
                        if (c >= 0) c = packet_walk<8, 1, ORDERED>(S, m, tmin, stk, c, bits);
                        if (c == RT_REF_NONE || c == RT_REF_EXIT_BLAS) break;
                        const uint32_t lv = (uint32_t)~c;
                        const uint32_t lcnt = lv & 7u, lfirst = lv >> 3;
                        uint32_t at = lfirst * (uint32_t)sizeof(RtTriXf);
                        for (uint32_t j = 0; j < lcnt; ++j, at += (uint32_t)sizeof(RtTriXf)) {
                            const int32_t gtri = (int32_t)(lfirst + j);
                            float t;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_PACKET_WALK_IN_CXX)
                            float beta, gamma;
                            packet_triangle_asm(tri_base, at, m[0], t, beta, gamma);
                            const bool tri_hit = t >= tmin && t <= m[0].t && gamma >= 0.0f && beta >= 0.0f && beta <= 1.0f - gamma;
#else
                            rt_f4 uv0, uv1, tn;
                            rt_fetch_48(tri_base, at, uv0, uv1, tn);
                            const bool tri_hit = fast_triangle(m[0], uv0, uv1, tn, tmin, t);
#endif
                            if (tri_hit && gtri != own) {
                                m[0].shape = shape_id;
                                m[0].tri = gtri;
                                m[0].t = ANYHIT ? ninf : t;
                            }
                        }
                        c = (ANYHIT && __ballot(m[0].t > ninf) == 0ull) ? RT_REF_NONE : stk.pop();
                    }
                }
                s[0].t = m[0].t;
                s[0].shape = m[0].shape;
                s[0].tri = m[0].tri;
                done = c == RT_REF_NONE;  // an any-hit wave with no lane left looking
                // the world ray again, for the rest of this leaf and the walk after it
                s[0].px = world[0 * 64]; s[0].py = world[1 * 64]; s[0].pz = world[2 * 64];
                s[0].ixy = world[3 * 64]; s[0].nxy = world[4 * 64]; s[0].zin = world[5 * 64];
            } else {
                // sphere / box / triangle at scene level, possibly instanced
                float t;
                bool hit;
                if (is_inst) {
                    FastTrav lr;
                    fast_local_ray(S, first + k, true, s[0].px.x, s[0].py.x, s[0].pz.x, s[0].px.y, s[0].py.y, s[0].pz.y, s[0], lr);
                    hit = fast_payload_hit(lr, proto_type, pp1, pp2, pp3, tmin, t);
                } else {
                    hit = fast_payload_hit(s[0], type, pp1, pp2, pp3, tmin, t);
                }
                if (hit && shape_id != s[0].org_shape) {
                    s[0].shape = shape_id;
                    s[0].tri = -1;
                    s[0].t = ANYHIT ? ninf : t;
                }
            }
        }
        if (done) break;
        cur = (ANYHIT && __ballot(s[0].t > ninf) == 0ull) ? RT_REF_NONE : stk.pop();
    }
    // The host sizes the rows for the deepest path, so this cannot fail; if it ever did, the surplus
    // entries have overwritten the world rays stored behind the rows and the frame is reported as failed.
    return stk.high <= max_sp;
}
