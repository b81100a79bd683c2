// rt_backend.hip -- gfx950 kernels of the wavefront pipeline and the C ABI
// around them (include/rt_hip.h).  Built for MI355X only.
//
// Launch shapes:
//   * camera rays and shadow rays (fast mode): one 64-thread block = one wave = one packet of
//     64 rays (rt_packet.h); blocks beyond the work of a launch return at once, because the
//     number of lit hits is only known on the device;
//   * mirror bounces and the reference-order mode: 256-thread blocks, one ray per lane, with a
//     per-lane traversal stack in LDS laid out [entry][thread] so that a wave's accesses to one
//     stack level fall on 64 consecutive dwords; a queue's length lives in device memory
//     (written by the previous stage's wave-wide ballot allocations), so no stage needs a host
//     round trip to learn its size;
//   * shade / accumulate / resolve: one thread per sample or pixel;
//   * scenes with a GGX shader: one thread per sample walks the sample's whole ray tree
//     (rt_tree.h) in place of the trace / shade / accumulate stages.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bvh_build.h"
#include "rt_backend_internal.h"
#include "rt_packet.h"
#include "rt_tree.h"
// Register budgets (waves per SIMD the compiler must make room for; 512 VGPRs / waves).  These
// kernels wait on memory most of the time, so one more resident wave is worth more than the
// handful of values that get spilled for it -- as long as the spills stay out of the loops:
//   packets   8 waves (64 VGPRs, no scratch since a mesh is walked with a ray variable of its own)
//   camera rays + the stage behind them in one kernel: 7 waves (72 VGPRs, no scratch): bunny frame 0.722 / 0.706 /
//             0.753 ms at 6 / 7 / 8 waves (at 8 the stage's f64 code spills as it does in k_shade<true>)
//   all mirror bounces in one kernel (one ray per lane): 6 waves (80 VGPRs, 144 B of scratch); cornell_room's
//             bounce launch 46 us at 5 waves, 49 at 6, 52 at 7 -- within noise of each other over a frame
//   shade     behind camera rays (k_shade<true>, no loop, no queue code): 7 waves (72 VGPRs, 100 B of scratch in
//             the box and sphere branches), bunny frame 0.192 -> 0.180 ms and cornell_room 0.512 -> 0.505 over 6
//             waves (79 VGPRs, 12 B); at 8 waves the spills reach the triangle path and it takes twice as long;
//             behind mirror bounces (k_shade<false>): 6 waves
//             too (78 VGPRs, no scratch) since it covers the queue with its grid instead of striding over it
//             in a loop, which alone took 125 registers (cornell_room: 0.555 -> 0.538 ms over its six launches)
//   shadow rays + light sums in one kernel: 8 waves (64 VGPRs, no scratch): bunny frame 0.862 -> 0.847 ms,
//             spheres_1K 0.594 -> 0.585, cornell_room 0.914 -> 0.903 over 7 waves.  (While the walk still
//             transformed the ray in place at an instance the kernel needed 72 registers and lost 5 % at 8 waves.)
//   accumulate 8 waves (60 VGPRs, no scratch; 76 without the hint): 0.195 -> 0.183 ms
#define RT_PACKET_WAVES 8
#ifndef RT_FUSED_WAVES
#define RT_FUSED_WAVES 8
#endif
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 6
#endif
#ifndef RT_SHADE_FIRST_WAVES
#define RT_SHADE_FIRST_WAVES 7
#endif
#define RT_ACC_WAVES 8

#ifndef RT_BLOCK
#define RT_BLOCK 256
#endif
#define RT_MAX_GRID 2048  // 256 CUs x 8 resident blocks
#define RT_TREE_ROOM 4u   // ray trees level by level: nodes a level may have for every camera ray of its batch
#define RT_PACKET_BLOCK 64
// kernels that work through a chunk of RT_LIT_CHUNK samples per block
#define RT_LIT_BLOCK RT_LIT_CHUNK
#define RT_LIT_THREADS 256
#define RT_LIT_ROUNDS (RT_LIT_BLOCK / RT_LIT_THREADS)

// Work memory of rt_render_device_static: one array per device the library runs on, part of the
// code object, so it exists from the moment the library is loaded on a device until the process ends
// and no call has to allocate it.  Host code never uses this name as an address (on the host it is
// only a handle for hipGetSymbolAddress); rt_backend_create asks the runtime for the array's device
// address once per backend, with the backend's device current, and keeps it in RtBackend::scratch.
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

namespace {

// All backends of a device share that device's one static array, so the frames rendered in it must not overlap.
// The library sees to that itself: `done` is recorded behind every such frame, and the next one -- whatever
// backend and stream it comes through -- starts with a wait for it on the device (no host wait, no allocation).
// `in_call` catches two host threads inside rt_render_device_static for one device at the same time, whose
// enqueues would interleave: the second gets RT_ERR_INVALID.
struct StaticSlot {
    std::atomic<int> in_call{0};
    hipEvent_t done = nullptr;  // created by the device's first rt_backend_create
};
#define RT_STATIC_MAX_DEVICES 64
StaticSlot &static_slot(int device) {
    static StaticSlot slots[RT_STATIC_MAX_DEVICES];
    return slots[device];
}

// ------------------------------------------------------------------ device helpers

// Per-lane traversal stack in LDS, laid out [entry][thread]; six more rows behind the stack
// rows keep a lane's world-space ray while it is inside an instance.
struct LdsStack {
    int32_t *base;  // &lds[threadIdx.x]
    int depth;      // stack rows
    __device__ __forceinline__ void set(int i, int32_t v) { base[i * RT_BLOCK] = v; }
    __device__ __forceinline__ int32_t get(int i) const { return base[i * RT_BLOCK]; }
    __device__ __forceinline__ void set_aux(int i, float v) { base[(depth + i) * RT_BLOCK] = __float_as_int(v); }
    __device__ __forceinline__ float get_aux(int i) const { return __int_as_float(base[(depth + i) * RT_BLOCK]); }
};
#define RT_LDS_AUX_ROWS 6

// Reserves one slot per lane with `pred` set, using ONE atomic per wave: the
// 64-bit ballot gives each lane its rank among the wave's appenders.  Must be
// reached by every lane of the wave.
__device__ __forceinline__ uint32_t wave_alloc(bool pred, uint32_t *counter) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0u;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, leader, 64);
    return base + rank;
}

// A word that a kernel launched EARLIER left in memory, read at a wave-uniform address through the scalar cache
// (which a kernel starts with empty): one scalar load where the compiler, which cannot know that nothing in
// this kernel writes the word, would send the wave's vector memory unit.
__device__ __forceinline__ uint32_t uniform_load(const uint32_t *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) uint32_t *>(reinterpret_cast<uintptr_t>(p));
}

// Lists the wave's lit hits (RtBatch::direct_lit).  `in` = the lanes whose hit is lit and lies in chunk `chunk`
// (wave-uniform), `entry` = what such a lane lists.  Every lane of the wave must get here.  Counts the hits in
// block_counts[chunk] in any case; with `list` the count's old value is where the wave's entries go in the
// chunk's stretch of the lit list.
// (A list of the packets that hold entries, so that the kernel walking the list would not launch waves for
// the empty ones, was tried: one more counter, bumped by every wave that starts a packet -- a hundred thousand
// returning atomics on ONE address, which the memory side serves one at a time, 8 ns each: the camera-ray
// kernel of the bunny frame took 1.12 ms instead of 0.27.  The per-chunk counters here are a few thousand
// addresses with sixteen waves each.)
__device__ __forceinline__ void wave_list_lit(const RtBuffers &Q, bool list, unsigned long long in, bool mine, uint32_t chunk, uint32_t entry) {
    if (in == 0ull) return;
    const int leader = __ffsll((long long)in) - 1;
    uint32_t base = 0u;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(&Q.block_counts[chunk], (uint32_t)__popcll(in));
    if (!list) return;
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (mine) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(in >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)in, 0u));
        Q.lit_list[chunk * RT_LIT_CHUNK + base + rank] = entry;
    }
}

// Hardware deals consecutive workgroups round-robin over the 8 XCDs, each with its own
// 4 MiB L2.  With work items in image order that makes all eight L2s stream the same
// part of the scene at the same time.  This remap hands XCD x the chunks x, x+8, x+16 ...
// of `chunk` consecutive logical blocks, so each L2 serves a few tiles' worth of BVH at a
// time while the interleave keeps the XCDs evenly loaded.  Bijective on [0, n_blocks).
// (Which physical XCD gets which residue is not defined; only the grouping matters.)
// `chunk` is a power of two (the host sees to that): the remap is shifts and masks, where a division by a
// number only known at run time costs some twenty instructions, four times per wave.
__device__ __forceinline__ uint32_t xcd_chunked_block(uint32_t b, uint32_t n_blocks, uint32_t chunk) {
    const uint32_t chunk_log2 = (uint32_t)__builtin_ctz(chunk), span_log2 = chunk_log2 + 3u;
    const uint32_t full = (n_blocks >> span_log2) << span_log2;  // blocks covered by whole 8-chunk groups
    if (b >= full) return b;                                     // ragged tail: leave in place
    const uint32_t r = b & ((1u << span_log2) - 1u);
    return (b - r) + ((r & 7u) << chunk_log2) + (r >> 3);
}

// Development builds only (-DRT_WAVE_TIMELINE, tools/wave_timeline.py): every wave of the two packet kernels that
// finds work notes when it started and ended (the 100 MHz constant clock) in a log the tool hands over, entry =
// packet index (+ the camera kernel's grid for the shadow kernel): the shape of a kernel's tail, wave by wave.
#ifdef RT_WAVE_TIMELINE
__device__ unsigned long long *g_wave_log;
__device__ __forceinline__ void wave_log(uint32_t entry, unsigned long long t0) {
    const unsigned long long t1 = wall_clock64();
    if (g_wave_log != nullptr && threadIdx.x == 0u) {
        g_wave_log[2 * (size_t)entry] = t0;
        g_wave_log[2 * (size_t)entry + 1] = t1;
    }
}
#define RT_WAVE_T0() const unsigned long long wave_t0 = wall_clock64()
#define RT_WAVE_LOG(entry) wave_log((entry), wave_t0)
#else
#define RT_WAVE_T0() ((void)0)
#define RT_WAVE_LOG(entry) ((void)0)
#endif

// ------------------------------------------------------------------ kernels

// idx = local sample at bounce 0 (count = samples of the batch, passed by the host),
// queue slot afterwards (count left in device memory by the previous shade stage).
template <bool ORDERED, bool EXACT>
__global__ void __launch_bounds__(RT_BLOCK) k_trace_closest(RtSceneView S, RtBuffers Q, RtBatch B, int bounce, uint32_t n_samples, uint32_t xcd_chunk) {
    extern __shared__ int32_t lds[];
    LdsStack stk{lds + threadIdx.x, S.stack_depth};
    if (bounce == 0) {  // one thread per sample, launched with exactly ceil(n_samples / 256) blocks
        const uint32_t idx = xcd_chunked_block(blockIdx.x, gridDim.x, xcd_chunk) * RT_BLOCK + threadIdx.x;
        if (idx < n_samples) trace_closest_body<LdsStack, ORDERED, EXACT>(S, Q, B, 0, idx, stk);
        return;
    }
    const uint32_t count = Q.counters[RT_CNT_RAYS0 + bounce];
    for (uint32_t idx = blockIdx.x * RT_BLOCK + threadIdx.x; idx < count; idx += gridDim.x * RT_BLOCK)
        trace_closest_body<LdsStack, ORDERED, EXACT>(S, Q, B, bounce, idx, stk);
}

// FIRST: the stage behind the camera rays (bounce 0), launched with one thread per sample: no queue to
// read, and the compiler folds every "bounce == 0 ?" of the stage body.  Otherwise the grid covers the
// longest queue of mirror rays there can be (its length only the device knows) and the blocks beyond
// the queue's end leave at once.
template <bool FIRST>
__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(FIRST ? RT_SHADE_FIRST_WAVES : RT_SHADE_WAVES, FIRST ? RT_SHADE_FIRST_WAVES : RT_SHADE_WAVES)))
k_shade(RtSceneView S, RtBuffers Q, RtBatch B, RtIdOut ids, int bounce_arg, uint32_t n_samples) {
    const int bounce = FIRST ? 0 : bounce_arg;
    const uint32_t count = FIRST ? n_samples : Q.counters[RT_CNT_RAYS0 + bounce];
    {
        const uint32_t base = blockIdx.x * RT_BLOCK;
        if (base >= count) return;  // the grid covers the longest queue there can be
        const uint32_t idx = base + threadIdx.x;
        MirrorTmp tmp;
        const int act = idx < count ? shade_classify(S, Q, B, ids, bounce, idx, tmp) : RT_ACT_NONE;
        // mirror rays are sparse: compact them into the next bounce's queue
        const uint32_t slot = wave_alloc(act == RT_ACT_MIRROR, &Q.counters[RT_CNT_RAYS0 + bounce + 1]);
        if (act == RT_ACT_MIRROR) shade_commit_mirror(Q, bounce, slot, tmp);
        // Lit hits per chunk of RT_LIT_CHUNK samples, for the compaction of the lit table below.  At
        // bounce 0 a wave holds 64 consecutive samples of one chunk: one add per wave.  Later bounces
        // reach samples in queue order, which follows the sample order loosely: the wave adds once per
        // distinct chunk among its lit lanes (one add per lane made this kernel 1.7x slower on a frame
        // of mirrors: the lanes of a wave mostly hit the same counter and queue up on it).
        unsigned long long todo = __ballot(act == RT_ACT_LIT);
        if (FIRST) {
            if (todo != 0ull && (threadIdx.x & 63u) == 0u) atomicAdd(&Q.block_counts[idx / RT_LIT_CHUNK], (uint32_t)__popcll(todo));
        } else if (todo != 0ull) {
            const uint32_t chunk = act == RT_ACT_LIT ? (uint32_t)Q.q_sample[bounce & 1][idx] / RT_LIT_CHUNK : ~0u;
            while (todo != 0ull) {
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t which = (uint32_t)__shfl((int)chunk, leader, 64);
                const unsigned long long same = __ballot(chunk == which);
                if ((int)(threadIdx.x & 63u) == leader) atomicAdd(&Q.block_counts[which], (uint32_t)__popcll(same));
                todo &= ~same;
            }
        }
    }
}

// ---- fast mode.
//
// One thread per work item, no persistent loop.  (Handing idle lanes the next ray of a
// wave-private range was tried: it costs 1.2x .. 1.8x on coherent camera and shadow rays, because
// the lanes of a wave then sit at different depths of the tree and stop sharing nodes and branches.)
//
// KIND 0: closest hit of camera (bounce 0) / mirror rays.  KIND 1: shadow rays of light blockIdx.y.
template <int KIND>
__global__ void __launch_bounds__(RT_BLOCK)
k_trace_fast(RtSceneView S, RtBuffers Q, RtBatch B, int bounce, uint32_t n_samples, uint32_t xcd_chunk) {
    extern __shared__ int32_t lds[];
    LdsStack stk{lds + threadIdx.x, S.stack_depth};
    uint32_t count;
    if (KIND == 1) count = Q.counters[RT_CNT_LIT];
    else count = bounce == 0 ? n_samples : Q.counters[RT_CNT_RAYS0 + bounce];
    const uint32_t n_blocks = (count + RT_BLOCK - 1) / RT_BLOCK;  // blocks that have work; the rest leave
    if (blockIdx.x >= n_blocks) return;
    const uint32_t idx = xcd_chunked_block(blockIdx.x, n_blocks, xcd_chunk) * RT_BLOCK + threadIdx.x;
    if (idx >= count) return;
    if (KIND == 1) trace_shadow_fast_body<LdsStack>(S, Q, blockIdx.y, Q.lit_list[idx], stk);
    else trace_closest_fast_body<LdsStack>(S, Q, B, bounce, idx, stk);
}

// A kernel's arguments as they lie in its argument segment (by-value arguments follow one another at
// their natural alignment, like the members of a struct).  The compiler loads every argument field a
// kernel uses into scalar registers at the top and keeps it there; the packet kernels use dozens of
// pointers but have fifty-odd scalar registers besides the walk's, so most of them were parked in lanes of
// a vector register and read back, one vector instruction per register, wherever they were needed again
// (an eighth of the fused kernel's vector instructions, a tenth of the camera-ray kernel's).  Fields that
// the walk itself does not need are therefore read where they are used, through a pointer to the argument
// segment that the compiler cannot see through (`fresh`): scalar loads that hit the scalar cache, and no
// long-lived registers.  Each kernel checks the assumed layout against two of its arguments as the
// compiler passes them, and reports a failed frame if they disagree.
struct PacketKernargs {
    RtSceneView S;
    RtBuffers Q;
    RtBatch B;
    uint32_t n_samples, xcd_chunk, n_clear;
};
struct FusedKernargs {
    RtSceneView S;
    RtBuffers Q;
    RtBatch B;
    RtIdOut ids;
    uint32_t xcd_chunk;
};
template <class T>
__device__ __forceinline__ const T &fresh(const __attribute__((address_space(4))) T *p) {
    asm volatile("" : "+s"(p));
    return *(const T *)p;
}
typedef const __attribute__((address_space(4))) PacketKernargs *PacketKernargsPtr;
typedef const __attribute__((address_space(4))) FusedKernargs *FusedKernargsPtr;

// Packet form of the same two kernels (rt_packet.h): a wave traverses with its 64 rays together.
// Used for camera rays and shadow rays, whose neighbours in a wave are neighbouring samples.
// Dynamic LDS: one uniform stack of (stack_depth + 1) dwords per wave and the parked world ray
// (rt_packet_lds_dwords).  One wave per block: the
// waves share nothing, and their run times differ widely (a wave whose rays all miss the
// meshes is done in a few steps), so a block of four would hold its slots for the slowest
// (measured on the bunny frame: 0.51 -> 0.39 ms camera rays, 1.08 -> 0.96 ms shadow rays).
// R rays per lane: lane l of block b holds work items (b * R + r) * 64 + l, r < R -- each r a run of
// 64 consecutive samples, so loads and stores stay coalesced.  (Only R = 1 is built: two rays per lane
// were measured and dropped, and the traversal is written for one.)
template <int KIND, int R>
__global__ void __launch_bounds__(RT_PACKET_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_PACKET_WAVES, RT_PACKET_WAVES)))
k_trace_packet(RtSceneView S, RtBuffers Q, RtBatch B, uint32_t n_samples, uint32_t xcd_chunk, uint32_t n_clear) {
    extern __shared__ int32_t lds[];
    constexpr uint32_t per_block = RT_PACKET_BLOCK * R;
    const PacketKernargsPtr args = (PacketKernargsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t count, base;
    int rows;
    {
        const PacketKernargs &K = fresh(args);
        if (K.Q.lit_list != Q.lit_list || K.n_clear != n_clear) {  // the layout assumed for the argument segment
            *Q.overflow = 1;
            return;
        }
        // As a batch's first kernel (camera rays) it also zeroes the counters and per-chunk lit counts that the
        // NEXT batch is going to add to (the set this batch does not use): a store by the first few waves
        // instead of a memset (two more launches in front of every frame).  n_clear = 0: nothing to clear.
        if (KIND == 0)
            for (uint32_t i = blockIdx.x * RT_PACKET_BLOCK + threadIdx.x; i < K.n_clear; i += gridDim.x * RT_PACKET_BLOCK) K.Q.counters_next[i] = 0u;
        count = KIND != 0 ? K.Q.counters[RT_CNT_LIT] : K.n_samples;
        const uint32_t n_blocks = (count + per_block - 1) / per_block;  // blocks that have work; the rest leave
        if (blockIdx.x >= n_blocks) return;
        base = xcd_chunked_block(blockIdx.x, n_blocks, K.xcd_chunk * RT_BLOCK / per_block) * per_block + threadIdx.x;
        rows = K.S.stack_depth + 1;
    }
    FastTrav st[R];
    uint32_t sample[R];
    bool live[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const PacketKernargs &K = fresh(args);
        const uint32_t idx = base + (uint32_t)k * RT_PACKET_BLOCK;
        RaySetup r;
        sample[k] = 0;
        live[k] = idx < count;
        if (live[k]) {
            if (KIND == 1) {
                sample[k] = K.Q.lit_list[idx];
                live[k] = shadow_setup_lit(K.S, K.Q, blockIdx.y, sample[k], r);
            } else if (KIND == 2) {  // shadow rays that run to the CLOSEST blocker (light/point.rs:35), the own primitive's re-hit as their bound
                sample[k] = K.Q.lit_list[idx];
                live[k] = shadow_setup<false>(K.S, K.Q, blockIdx.y, sample[k], r);
            } else {
                live[k] = closest_setup(K.S, K.Q, K.B, 0, idx, r);
            }
        }
        if (!live[k]) {  // a slot without a ray: finite dummy values and an empty search interval
            r.o = D3{0.0, 0.0, 0.0};
            r.d = D3{0.0, 0.0, -1.0};
            r.t0 = -__builtin_inff();
            r.shape0 = -1;
            r.org_shape = r.org_tri = -1;
        }
        fast_set_ray(st[k], (float)r.o.x, (float)r.o.y, (float)r.o.z, (float)r.d.x, (float)r.d.y, (float)r.d.z);
        st[k].t = r.t0;
        st[k].shape = r.shape0;
        st[k].tri = -1;
        st[k].org_shape = r.org_shape;
        st[k].org_tri = r.org_tri;
    }
    const float tmin = KIND != 0 ? (float)RT_VERY_SMALL_NUMBER : closest_tmin(0);
    const bool ok = packet_traverse<KIND == 1, R>(S, st, tmin, lds, rows - 1);  // (the walk's fields of S stay in registers)
    const PacketKernargs &K = fresh(args);
    if (!ok) *K.Q.overflow = 1;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (!live[k]) continue;
        if (KIND != 0) shadow_store(K.Q, blockIdx.y, sample[k], st[k].shape, st[k].tri);
        else closest_store(K.Q, base + (uint32_t)k * RT_PACKET_BLOCK, st[k].t, st[k].shape, st[k].tri);
    }
}

// Shadow rays AND the light sums of the lit hits in one kernel (fast mode): the wave takes 64 entries
// of the lit list, walks the tree once per point light with their 64 shadow rays as a packet, keeps
// one "hidden" bit per light and lane, and then adds up the lights in scene order as the accumulate
// stage does.  Against one packet launch per light plus an accumulate launch this drops the
// per-light blocker arrays (a write and a read of 4 B per ray), reads a hit's point, primitive and
// origin ids once instead of once per light, and -- what matters most -- puts the f64 arithmetic of
// the light sums into the issue slots that the tree walk of other waves leaves idle while it waits
// for its node records.
//
// The walk runs at the register budget of eight waves per SIMD with nothing to spare, so nothing but
// the walk's own state may stay in vector registers across it: a lane's "hidden" bits wait in an
// LDS word of its own, and everything else about its sample is fetched again where it is needed
// (`opaque` keeps the compiler from hoisting those fetches, and the addresses they go to, out of
// the loop over the lights and across the walk).
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}

__global__ void __launch_bounds__(RT_PACKET_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_FUSED_WAVES, RT_FUSED_WAVES)))
k_shadow_shade_packet(RtSceneView S, RtBuffers Q, RtBatch B, RtIdOut ids, uint32_t xcd_chunk) {
    extern __shared__ int32_t lds[];
    const FusedKernargsPtr args = (FusedKernargsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t count, idx0;
    uint32_t sample0 = 0u;  // the lane's entry of the lit list
    int rows, n_point_lights;
    bool want_ids, shadows;
    {
        const FusedKernargs &K = fresh(args);
        // Everything the way in needs, fetched together: a third of a frame's waves find their packet of the lit
        // list empty and leave here, and what they cost is the round trips they wait for one after the other.
        const uint32_t *const k_lit_list = K.Q.lit_list, *const k_block_counts = K.Q.block_counts, *const k_counters = K.Q.counters;
        const uint32_t k_chunk = K.xcd_chunk, k_direct = K.B.direct_lit;
        const uint32_t grid = gridDim.x;
        asm volatile("" ::"s"(k_lit_list), "s"(k_block_counts), "s"(k_counters), "s"(k_chunk), "s"(k_direct), "s"(grid));  // (all of them issued before the first wait)
        // the layout assumed above, checked against two arguments as the compiler passes them
        if (k_lit_list != Q.lit_list || k_chunk != xcd_chunk) {
            *Q.overflow = 1;
            return;
        }
        if (k_direct) {
            // The lit list has a stretch per chunk of samples, filled from its start (RtBuffers::lit_list): the grid
            // covers every stretch, and the waves behind a stretch's last entry leave.
            constexpr uint32_t per_chunk = RT_LIT_CHUNK / RT_PACKET_BLOCK;
            const uint32_t blk = xcd_chunked_block(blockIdx.x, grid, k_chunk * RT_BLOCK / RT_PACKET_BLOCK);
            const uint32_t chunk = blk / per_chunk, first = (blk % per_chunk) * RT_PACKET_BLOCK;
            idx0 = chunk * RT_LIT_CHUNK + first + threadIdx.x;
            // The lane's list entry is asked for BEFORE the chunk's count is known (the slot exists whether or not it
            // has been filled): the two round trips then pass side by side, and a wave that has work -- two in three --
            // starts its record fetches one round trip earlier.
            sample0 = k_lit_list[idx0];
            const uint32_t listed = uniform_load(k_block_counts + chunk);
            if (first >= listed) return;
            count = chunk * RT_LIT_CHUNK + listed;
        } else {
            count = uniform_load(k_counters + RT_CNT_LIT);
            const uint32_t n_blocks = (count + RT_PACKET_BLOCK - 1) / RT_PACKET_BLOCK;  // blocks that have work; the rest leave
            if (blockIdx.x >= n_blocks) return;
            idx0 = xcd_chunked_block(blockIdx.x, n_blocks, k_chunk * RT_BLOCK / RT_PACKET_BLOCK) * RT_PACKET_BLOCK + threadIdx.x;
            if (idx0 < count) sample0 = k_lit_list[idx0];
        }
        rows = K.S.stack_depth + 1;
        n_point_lights = K.S.n_point_lights;
        want_ids = K.ids.shadow_shape != nullptr || K.ids.shadow_tri != nullptr;
        shadows = !K.S.disable_shadows;
    }
    RT_WAVE_T0();
    // One word per lane behind the walk's LDS, bit k = point light k is hidden from the lane's hit.  It starts as the
    // bits of the lights that the hit's own primitive hides (the f64 test of the stage that found the hit; all ones for
    // a slot without a hit, which traces nothing): light k's turn reads bit k before anything but that test can have
    // set it, and sets it if the walk finds a blocker.
    uint32_t *hidden_slot = reinterpret_cast<uint32_t *>(lds + rt_packet_lds_dwords(rows)) + threadIdx.x;
    *hidden_slot = 0u;
    // What every light's shadow ray needs of the lane's lit hit -- the point, the primitive it lies on, the lights that
    // primitive hides by itself -- is fetched from the lit table ONCE and waits in LDS, three pairs per lane behind the
    // word above: fetched again per light it was a chain of two memory round trips (list entry, then the record: 2-3 us)
    // at the head of every walk, and 28 bytes per sample and light more from HBM (the work buffers bypass the L2).
    rt_f2 *rec = reinterpret_cast<rt_f2 *>(lds + rt_packet_lds_dwords(rows) + RT_PACKET_BLOCK) + threadIdx.x;  // [3][64] pairs
    const bool live = idx0 < count;
    if (shadows) {
        const FusedKernargs &K = fresh(args);
        float px = 0.0f, py = 0.0f, pz = 0.0f;
        int32_t org_shape = -1, org_tri = -1;
        uint32_t self_mask = ~0u;  // a slot without a hit traces nothing
        if (live) {
            const uint32_t sample = sample0;
            org_shape = RT_STREAM_LD(K.Q.l_shape[sample]);
            org_tri = RT_STREAM_LD(K.Q.l_tri[sample]);
            self_mask = RT_STREAM_LD(K.Q.l_self[sample]);
            px = RT_STREAM_LD(reinterpret_cast<const float *>(K.Q.l_p[0])[sample]);  // stored as f32: the fused kernel is fast mode's
            py = RT_STREAM_LD(reinterpret_cast<const float *>(K.Q.l_p[1])[sample]);
            pz = RT_STREAM_LD(reinterpret_cast<const float *>(K.Q.l_p[2])[sample]);
        }
        rec[0 * RT_PACKET_BLOCK] = rt_pair(px, py);
        rec[1 * RT_PACKET_BLOCK] = rt_pair(pz, rt_as_float((int32_t)sample0));  // (the list entry: not fetched again at the end)
        rec[2 * RT_PACKET_BLOCK] = rt_pair(rt_as_float(org_shape), rt_as_float(org_tri));
        *hidden_slot = self_mask;
    }
    bool ok = true;
#ifdef RT_WAVE_TIMELINE
    wave_log(16u * gridDim.x + blockIdx.x, wave_t0);  // (third record: the wave's way in, up to its first walk)
#endif
    if (shadows) {
        for (int ls = 0; ls < n_point_lights; ++ls) {
            FastTrav st[1];
            bool self = false;
            {
                const FusedKernargs &K = fresh(args);
                D3 o = D3{0.0, 0.0, 0.0}, d = D3{0.0, 0.0, -1.0};
                const rt_f2 r0 = rec[0 * RT_PACKET_BLOCK], r1 = rec[1 * RT_PACKET_BLOCK], r2 = rec[2 * RT_PACKET_BLOCK];
                const int32_t org_shape = rt_as_int(r2.x), org_tri = rt_as_int(r2.y);
                self = ((*hidden_slot >> ls) & 1u) != 0u;
                const bool trace = !self;
                if (trace) {
                    o = D3{(double)r0.x, (double)r0.y, (double)r1.x};
                    d = d3(K.S.lights[K.S.point_lights[ls]].position) - o;  // Ray::atob, ray.rs:15-20
                }
                fast_set_ray(st[0], (float)o.x, (float)o.y, (float)o.z, (float)d.x, (float)d.y, (float)d.z);
                st[0].t = trace ? 1.0f : -__builtin_inff();  // Hit::to_light's bound, or an empty search interval
                st[0].shape = -1;
                st[0].tri = -1;
                st[0].org_shape = org_shape;
                st[0].org_tri = org_tri;
            }
            ok = packet_traverse<true, 1>(S, st, (float)RT_VERY_SMALL_NUMBER, lds, rows - 1) && ok;  // (the walk's fields of S stay in registers)
            if (live && st[0].shape >= 0) *hidden_slot |= 1u << ls;
            if (want_ids && live) {
                const FusedKernargs &K = fresh(args);
                const uint32_t sample = (uint32_t)rt_as_int(rec[1 * RT_PACKET_BLOCK].y);
                const size_t at = global_sample(K.B, sample) * (size_t)n_point_lights + (size_t)ls;
                const int32_t blocker = self ? RT_STREAM_LD(K.Q.l_shape[sample]) : st[0].shape;
                if (K.ids.shadow_shape) K.ids.shadow_shape[at] = blocker;
                if (K.ids.shadow_tri) K.ids.shadow_tri[at] = self ? -1 : tri_file_index(K.S, st[0].tri);
            }
        }
    }
#ifdef RT_WAVE_TIMELINE
    const unsigned long long walked = wall_clock64();
#endif
    const FusedKernargs &K = fresh(args);
    if (!ok) *K.Q.overflow = 1;
    const uint32_t idx = opaque(idx0);
    if (idx < count) {
        uint32_t sample;
        D3 p;
        int32_t shape;
        if (shadows) {  // list entry, point and primitive wait in LDS
            const rt_f2 r0 = rec[0 * RT_PACKET_BLOCK], r1 = rec[1 * RT_PACKET_BLOCK], r2 = rec[2 * RT_PACKET_BLOCK];
            sample = (uint32_t)rt_as_int(r1.y);
            p = D3{(double)r0.x, (double)r0.y, (double)r1.x};
            shape = rt_as_int(r2.x);
        } else {
            sample = K.Q.lit_list[idx];
            p = lit_load_vec(K.Q.l_p, sample, true);
            shape = RT_STREAM_LD(K.Q.l_shape[sample]);
        }
        accumulate_masked<true>(K.S, K.Q, sample, shape, p, *hidden_slot);  // fast mode only
    }
    RT_WAVE_LOG(gridDim.x + blockIdx.x);
#ifdef RT_WAVE_TIMELINE
    wave_log(15u * gridDim.x + blockIdx.x, walked);
#endif
}

// Camera rays AND the stage behind them in one kernel (fast mode): the wave walks the tree with its 64 camera
// rays as a packet and then decides what happens to each hit (k_shade<true>'s work) without the hit records
// going through memory.  Like k_trace_packet<0> it zeroes the next batch's counters (its own are being
// added to by its waves).
#ifndef RT_CAMERA_WAVES
#define RT_CAMERA_WAVES 7
#endif
struct CameraKernargs {
    RtSceneView S;
    RtBuffers Q;
    RtBatch B;
    RtIdOut ids;
    uint32_t n_samples, xcd_chunk, n_clear;
};
typedef const __attribute__((address_space(4))) CameraKernargs *CameraKernargsPtr;

__global__ void __launch_bounds__(RT_PACKET_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_CAMERA_WAVES, RT_CAMERA_WAVES)))
k_camera_shade_packet(RtSceneView S, RtBuffers Q, RtBatch B, RtIdOut ids, uint32_t n_samples, uint32_t xcd_chunk, uint32_t n_clear) {
    extern __shared__ int32_t lds[];
    const CameraKernargsPtr args = (CameraKernargsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t count, idx0;
    int rows;
    {
        const CameraKernargs &K = fresh(args);
        if (K.Q.lit_list != Q.lit_list || K.n_clear != n_clear) {  // the layout assumed for the argument segment
            *Q.overflow = 1;
            return;
        }
        for (uint32_t i = blockIdx.x * RT_PACKET_BLOCK + threadIdx.x; i < K.n_clear; i += gridDim.x * RT_PACKET_BLOCK) K.Q.counters_next[i] = 0u;
        count = K.n_samples;
        const uint32_t n_blocks = (count + RT_PACKET_BLOCK - 1) / RT_PACKET_BLOCK;
        if (blockIdx.x >= n_blocks) return;
        idx0 = xcd_chunked_block(blockIdx.x, n_blocks, K.xcd_chunk * RT_BLOCK / RT_PACKET_BLOCK) * RT_PACKET_BLOCK + threadIdx.x;
        rows = K.S.stack_depth + 1;
    }
    // The sample's point on the image plane waits in LDS behind the walk's rows, a column of four words per
    // lane: the ray is needed again for the hit point, and the plane point is the expensive part of it
    // (index arithmetic and four f64 divisions) while the walk has no register to spare for it.
    RT_WAVE_T0();
    int32_t *plane = lds + rt_packet_lds_dwords(rows) + threadIdx.x;
    FastTrav st[1];
    bool live;
    {
        const CameraKernargs &K = fresh(args);
        RaySetup r;
        double u = 0.0, v = 0.0;
        live = idx0 < count && sample_plane_point(K.S, K.B, idx0, u, v);
        if (live) {
            camera_ray_through(K.S.camera, u, v, r.o, r.d);
            r.t0 = __builtin_inff();
        } else {  // a slot without a ray: finite dummy values and an empty search interval
            r.o = D3{0.0, 0.0, 0.0};
            r.d = D3{0.0, 0.0, -1.0};
            r.t0 = -__builtin_inff();
        }
        plane[0 * RT_PACKET_BLOCK] = __double2loint(u);
        plane[1 * RT_PACKET_BLOCK] = __double2hiint(u);
        plane[2 * RT_PACKET_BLOCK] = __double2loint(v);
        plane[3 * RT_PACKET_BLOCK] = __double2hiint(v);
        fast_set_ray(st[0], (float)r.o.x, (float)r.o.y, (float)r.o.z, (float)r.d.x, (float)r.d.y, (float)r.d.z);
        st[0].t = r.t0;
        st[0].shape = -1;
        st[0].tri = -1;
        st[0].org_shape = -1;
        st[0].org_tri = -1;
    }
    const bool ok = packet_traverse<false, 1>(S, st, closest_tmin(0), lds, rows - 1);  // (the walk's fields of S stay in registers)
#ifdef RT_WAVE_TIMELINE
    const unsigned long long walked = wall_clock64();  // (second record of the wave: from the end of the walk to the end)
#endif
    const CameraKernargs &K = fresh(args);
    if (!ok) *K.Q.overflow = 1;
    const uint32_t idx = opaque(idx0);
    MirrorTmp tmp;
    int act = RT_ACT_NONE;
    if (live) {
        D3 o, d;
        camera_ray_through(K.S.camera, __hiloint2double(plane[1 * RT_PACKET_BLOCK], plane[0 * RT_PACKET_BLOCK]),
                           __hiloint2double(plane[3 * RT_PACKET_BLOCK], plane[2 * RT_PACKET_BLOCK]), o, d);
        act = shade_hit(K.S, K.Q, K.B, K.ids, 0, (int32_t)idx, o, d, st[0].shape, st[0].tri, st[0].t, tmp);
    }
    // every sample of the batch passes here exactly once: this clears the lit table where no lit hit was stored
    if (idx < count && act != RT_ACT_LIT) K.Q.l_shape[idx] = -1;
    // Mirror rays go into queue 1.  With chunked_queue (the packet kernel for the bounces follows) into the stretch
    // of the wave's chunk of samples, a counter per chunk: one counter for the whole queue is a returning atomic
    // per wave on ONE address, served one at a time (8 ns each: 1.1 ms for a frame in which every wave sees a
    // mirror).  Otherwise compacted behind one counter, as the separate stage does.
    {
        const bool chunked = K.B.chunked_queue != 0u;
        const uint32_t chunk = idx / RT_LIT_CHUNK;
        const uint32_t slot = wave_alloc(act == RT_ACT_MIRROR, chunked ? &K.Q.mirror_counts[chunk] : &K.Q.counters[RT_CNT_RAYS0 + 1]);
        if (act == RT_ACT_MIRROR) shade_commit_mirror(K.Q, 0, (chunked ? chunk * RT_LIT_CHUNK : 0u) + slot, tmp);
    }
    // Lit hits: counted per chunk of samples (the wave's 64 samples lie in one chunk) and, with direct_lit,
    // listed at once in the chunk's stretch of the lit list, at the place the count gives.
    // (The atomic's round trip is 1.1 us of a wave's 13.  Asking for the places as soon as it is known which hits are
    // lit -- the stage behind the hits split in two, the stores and f64 tests of a lit hit behind the request -- was
    // measured against this on one box: bunny 0.657 vs 0.643 ms, cornell_room 0.536 vs 0.527, spheres_1K 0.403 vs
    // 0.400.  Loads, stores and atomics answer in order on this hardware, so every wait of the second half waits
    // for the atomic too; and the split alone, without the early request, already cost 1 - 2 %.)
    wave_list_lit(K.Q, K.B.direct_lit != 0u, __ballot(act == RT_ACT_LIT), act == RT_ACT_LIT, idx / RT_LIT_CHUNK, idx);
    RT_WAVE_LOG(blockIdx.x);
#ifdef RT_WAVE_TIMELINE
    wave_log(14u * gridDim.x + blockIdx.x, walked);
#endif
}

// ALL mirror bounces in one kernel (fast mode): a lane takes a ray from the first bounce's queue, follows it
// to its hit, decides what happens there (the stage behind the rays; the hit record stays in registers), and
// where the hit is another mirror it goes on with the reflected ray itself.  A ray's way through one bounce
// is a chain of dependent memory round trips and f64 arithmetic some 25 us long, so a launch per bounce and
// one more for the stage behind it took that long each however few rays were left, with the GPU drained in
// between (cornell_room, six bounces: 0.14 ms for the first, 0.13 ms for the five behind it, whose rays are
// a tenth of the first's).  One launch takes as long as those did together (0.265 ms) but leaves no gaps
// for a frame in flight beside it: cornell_room 0.785 -> 0.757 ms per delivered frame.  Lanes whose ray has
// ended idle until the wave's last one has: against compacting the rays after every bounce that costs
// nothing measurable even where mirrors dominate (the same room with floor and back wall mirrors, 6.5 M
// mirror rays: 2.35 vs 2.37 ms; every wall a mirror, 27.7 M: 7.0 vs 7.3 ms).
// The reflected ray goes through the lane's own queue slot, so that a round of the loop starts from memory
// like the first one and nothing but the slot's index lives across it.  The rounds of a wave are in step
// (round k is bounce k + 1 for every lane that is still going), so the counts of rays per bounce that the
// statistics report are kept with one add per wave and round.
// Only the device knows how long the queue is, so the grid is small and its blocks stride over the queue.
// The arguments are read again in every round (`fresh`), so that the compiler does not keep the address of
// every queue and table in registers across the loops.
#ifndef RT_BOUNCE_WAVES
#define RT_BOUNCE_WAVES 6
#endif
// as many blocks as are resident at once, 256 CUs x (4 SIMDs x RT_BOUNCE_WAVES waves / 4 waves per block):
// cornell_room 1.036 / 1.018 / 1.045 / 1.088 ms per instrumented frame with 1024 / 1536 / 2048 / 4096 blocks,
// the same room with every wall a mirror (27.7 M mirror rays) 7.11 / 7.00 / 7.35 / 7.21
#define RT_BOUNCE_GRID (256u * RT_BOUNCE_WAVES)
struct BounceKernargs {
    RtSceneView S;
    RtBuffers Q;
    RtBatch B;
    RtIdOut ids;
    uint32_t xcd_chunk;
};
typedef const __attribute__((address_space(4))) BounceKernargs *BounceKernargsPtr;

__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_BOUNCE_WAVES, RT_BOUNCE_WAVES)))
k_bounces_fast(RtSceneView S, RtBuffers Q, RtBatch B, RtIdOut ids, uint32_t xcd_chunk) {
    extern __shared__ int32_t lds[];
    const BounceKernargsPtr args = (BounceKernargsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    {
        const BounceKernargs &K = fresh(args);
        if (K.Q.lit_list != Q.lit_list || K.xcd_chunk != xcd_chunk) {  // the layout assumed for the argument segment
            *Q.overflow = 1;
            return;
        }
    }
    // Rays that go on into bounce b + 1, added up per block and handed to the batch's counters once at the end:
    // an add per wave and round straight to the counters is tens of thousands of atomics on a handful of
    // addresses, which the memory side serves one after the other (8 ns each) -- as long as the kernel itself.
    __shared__ uint32_t tally[32];
    if (threadIdx.x < 32u) tally[threadIdx.x] = 0u;
    __syncthreads();
    constexpr int which = 1;  // the queue of bounce 1
    const uint32_t count = Q.counters[RT_CNT_RAYS0 + 1];
    const uint32_t n_blocks = (count + RT_BLOCK - 1) / RT_BLOCK;
    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint32_t idx = xcd_chunked_block(blk, n_blocks, fresh(args).xcd_chunk) * RT_BLOCK + threadIdx.x;
        bool going = idx < count;
        for (int b = 1; __any(going); ++b) {
            const BounceKernargs &K = fresh(args);
            LdsStack stk{lds + threadIdx.x, K.S.stack_depth};
            FastTrav st;
            st.t = __builtin_inff();
            st.shape = st.tri = -1;
            if (going) {
                {
                    D3 o, d;
                    queue_read_ray(K.Q, which, idx, o, d);
                    const int32_t org_shape = K.Q.q_org_shape[which][idx];
                    const float self_t = K.Q.q_self_t[which][idx];
                    const bool self = self_t < __builtin_inff();  // the origin primitive's own re-hit
                    fast_begin(K.S, st, stk, o, d, self ? self_t : __builtin_inff(), self ? org_shape : -1, org_shape, K.Q.q_org_tri[which][idx]);
                }
                bool overflow = false;
                const float tmin = closest_tmin(1);
                while (fast_step<LdsStack, false>(K.S, st, stk, tmin, K.S.stack_depth, overflow)) {
                }
                if (overflow) *K.Q.overflow = 1;
            }
            const BounceKernargs &L = fresh(args);
            MirrorTmp tmp;
            int act = RT_ACT_NONE;
            int32_t sample = 0;
            if (going) {
                D3 o, d;
                sample = L.Q.q_sample[which][idx];
                queue_read_ray(L.Q, which, idx, o, d);
                act = shade_hit(L.S, L.Q, L.B, L.ids, b, sample, o, d, st.shape, st.tri, st.t, tmp);
            }
            going = act == RT_ACT_MIRROR;
            if (going) queue_write_ray(L.Q, which, idx, tmp);
            const unsigned long long on = __ballot(going);
            if (on != 0ull && (threadIdx.x & 63u) == 0u) atomicAdd(&tally[b & 31], (uint32_t)__popcll(on));
            // lit hits per chunk of samples, one add per distinct chunk among the wave's lit lanes (as k_shade does)
            unsigned long long todo = __ballot(act == RT_ACT_LIT);
            const uint32_t chunk = act == RT_ACT_LIT ? (uint32_t)sample / RT_LIT_CHUNK : ~0u;
            while (todo != 0ull) {
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t of = (uint32_t)__shfl((int)chunk, leader, 64);
                const unsigned long long same = __ballot(chunk == of);
                wave_list_lit(L.Q, L.B.direct_lit != 0u, same, chunk == of, of, (uint32_t)sample);
                todo &= ~same;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 31u && tally[threadIdx.x] != 0u) atomicAdd(&Q.counters[RT_CNT_RAYS0 + threadIdx.x + 1], tally[threadIdx.x]);
}

// The same with the rays of a wave walked as a packet (rt_packet.h), one wave per block: rays that leave a mirror
// next to each other mostly stay next to each other, and a packet's walk costs a fraction of 64 separate ones.
// Queue slots are dealt to waves in runs of 64, which is how the camera-ray kernel filled them: a run is the
// mirror hits of one or two of its waves.  This is what the fast mode runs; RT_RENDER_NO_PACKETS keeps the
// per-lane kernel above.  Instrumented frames, per-lane -> packets: cornell_room 0.996 -> 0.952 ms (delivered,
// four in flight: 0.746 -> 0.702), the same room with floor and back wall mirrors 2.04 -> 1.92, every wall a
// mirror 4.38 -> 4.33, the bunny scene with the large bunny and the ground as mirrors 4.78 -> 3.24.
// With statistics the grid is one resident set of waves striding over the queue, each handing its ray counts to the
// batch's counters once at the end (with a block per packet the delivered cornell_room frame took 1.0 ms, because
// every block then ended with its own adds to those counters).  A frame without statistics needs no counts:
// `tally_on` = 0, and the grid is a wave per packet, which the hardware deals out as waves finish -- mirror
// chains differ in length (22 616 packets at bounce 1, 2909 at bounce 2, 42 at bounce 6 in cornell_room) and a fixed
// deal waits for its unluckiest wave (one blocking frame 0.900 -> 0.858 ms).
// Ray trees level by level (see k_tree_count): where the rays of a level come from.  The rays of level L >= 1 are
// never written anywhere: the lane that walks ray c makes it, in its first round, from the GGX hit its parent left
// (the k-th reflected ray of ggx_mirror.rs:66-76; k = c minus the parent's first child), and notes the ray's tree
// and path for the level below.  level = 0: not a level of ray trees, the rays lie in mirror-ray queue 1.
struct TreeSpawn {
    const uint32_t *parent, *base;       // per ray of this level: its parent in the level above; per parent: its first child
    const double *p[3], *n[3], *in[3];   // the parents' GGX hits (RtBuffers::g_* of the level above)
    const int32_t *shape, *tri, *mark;
    const uint64_t *root_in, *path_in;   // per parent: its tree's frame-wide sample index, its ray's path (level >= 2)
    uint64_t *root_out, *path_out;       // per ray of this level
    uint32_t level;
};

__device__ __forceinline__ void tree_spawn_ray(const RtSceneView &S, const RtBatch &B, const TreeSpawn &T, uint32_t c, MirrorTmp &ray, int &depth) {
    const uint32_t parent = T.parent[c], k = c - T.base[parent];
    const int32_t mark = T.mark[parent];
    const RtShader &sh = S.shaders[(uint32_t)mark >> 8];
    const D3 p = D3{T.p[0][parent], T.p[1][parent], T.p[2][parent]};
    const D3 n = D3{T.n[0][parent], T.n[1][parent], T.n[2][parent]};
    const D3 incoming = D3{T.in[0][parent], T.in[1][parent], T.in[2][parent]};
    const uint64_t root = T.level == 1u ? frame_sample(B, parent) : T.root_in[parent];
    const uint64_t path = tree_child_path(T.level == 1u ? 0ull : T.path_in[parent], k);
    ray.p = p;
    ray.d = ggx_child_direction(n, incoming, sh.roughness, root, path, B.seed);
    ray.sample = (int32_t)c;
    ray.shape = T.shape[parent];
    ray.tri = T.tri[parent];
    double st;
    ray.self_t = self_hit_f64(S, ray.shape, ray.p, ray.d, RT_VERY_SMALL_NUMBER, (double)__builtin_inff(), dot(n, ray.d) > 0.0, false, st)
                     ? (float)st
                     : __builtin_inff();
    depth = (mark & 0xff) + 1;
    RT_STREAM_ST(T.root_out[c], root);
    RT_STREAM_ST(T.path_out[c], path);
}

#define RT_BOUNCE_KEEP 17
__device__ __forceinline__ void keep_ray(int32_t *keep, const D3 &o, const D3 &d) {
    const double v[6] = {o.x, o.y, o.z, d.x, d.y, d.z};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        keep[(2 * k) * RT_PACKET_BLOCK] = __double2loint(v[k]);
        keep[(2 * k + 1) * RT_PACKET_BLOCK] = __double2hiint(v[k]);
    }
}
__device__ __forceinline__ void kept_ray(const int32_t *keep, D3 &o, D3 &d) {
    double v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = __hiloint2double(keep[(2 * k + 1) * RT_PACKET_BLOCK], keep[(2 * k) * RT_PACKET_BLOCK]);
    o = D3{v[0], v[1], v[2]};
    d = D3{v[3], v[4], v[5]};
}
struct BouncePacketKernargs {
    RtSceneView S;
    RtBuffers Q;
    RtBatch B;
    RtIdOut ids;
    uint32_t n_samples;
    uint32_t tally_on;
    TreeSpawn T;
};
typedef const __attribute__((address_space(4))) BouncePacketKernargs *BouncePacketKernargsPtr;
// Waves per SIMD: the stage behind a mirror ray's hit is a chain of dependent loads with f64 arithmetic in between,
// and at the 72 registers of 7 waves it keeps 172 bytes per lane in scratch memory, every reload one more round trip
// in that chain.  At 5 waves (96 registers, 24 bytes) a packet's bounce is shorter by more than the two waves are
// worth: cornell_room delivered 0.577 ms at 7, 0.560 at 6, 0.549 at 5, 0.553 at 4 (117 registers, no scratch).
#ifndef RT_BOUNCE_PACKET_WAVES
#define RT_BOUNCE_PACKET_WAVES 5
#endif
#ifndef RT_BOUNCE_PACKET_RESIDENT
#define RT_BOUNCE_PACKET_RESIDENT RT_BOUNCE_PACKET_WAVES  // waves per SIMD that the kernel's grid asks for
#endif
// TREE: the rays are a level of ray trees and are made here (TreeSpawn); a kernel of its own, so that the code that
// makes a ray (trigonometry in f64) costs the mirror bounces of every other scene neither registers nor scratch.
template <bool TREE>
__device__ __forceinline__ void bounces_packet_body(const RtSceneView &S, const RtBuffers &Q, uint32_t n_samples, uint32_t tally_on, int32_t *lds) {
    const BouncePacketKernargsPtr args = (BouncePacketKernargsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    int rows;
    {
        const BouncePacketKernargs &K = fresh(args);
        if (K.Q.lit_list != Q.lit_list || K.n_samples != n_samples) {  // the layout assumed for the argument segment
            *Q.overflow = 1;
            return;
        }
        rows = K.S.stack_depth + 1;
    }
    // rays that go on into bounce b + 1, handed to the batch's counters once at the end (see k_bounces_fast):
    // a word per bounce behind the walk's LDS
    uint32_t *tally = reinterpret_cast<uint32_t *>(lds + rt_packet_lds_dwords(rows));
    tally[threadIdx.x] = 0u;
    __syncthreads();
    // What a lane knows about its ray waits in LDS, a column of RT_BOUNCE_KEEP words behind the tally: the f64 ray
    // (needed for the fp32 search before the walk and for the hit point after it), the sample it belongs to, the
    // depth it started at, and for a ray that a mirror of this kernel made, the primitive it starts on.  The queue
    // is read once, by a ray's first round; a reflected ray goes from round to round through these words and never
    // to memory.  (Reading the ray again after the walk, and a reflected ray back from its queue slot, was a memory
    // round trip of 1.5 - 2 us each in a round of 11, and 120 bytes per ray from HBM where the rays of a ray tree's
    // level make this kernel's time.)
    int32_t *keep = lds + rt_packet_lds_dwords(rows) + RT_PACKET_BLOCK + threadIdx.x;
    constexpr int which = 1;  // the queue of bounce 1
    // the queue has a stretch per chunk of samples (RtBuffers::mirror_counts): the waves stride over all packets
    // of all stretches and pass over those behind a stretch's last ray
    constexpr uint32_t per_chunk = RT_LIT_CHUNK / RT_PACKET_BLOCK;
    const uint32_t n_packets = (n_samples + RT_PACKET_BLOCK - 1) / RT_PACKET_BLOCK;
    bool ok = true;
    for (uint32_t blk = blockIdx.x; blk < n_packets; blk += gridDim.x) {
        const uint32_t chunk = blk / per_chunk, first = (blk % per_chunk) * RT_PACKET_BLOCK;
        const uint32_t queued = uniform_load(fresh(args).Q.mirror_counts + chunk);
        if (first >= queued) continue;
        const uint32_t idx0 = chunk * RT_LIT_CHUNK + first + threadIdx.x;
        bool going = first + threadIdx.x < queued;
        if (tally_on != 0u && threadIdx.x == 0u) tally[0] += min(queued - first, (uint32_t)RT_PACKET_BLOCK);  // rays of bounce 1
        for (int b = 1; __any(going); ++b) {
            FastTrav st[1];
#ifdef RT_WAVE_TIMELINE
            const unsigned long long bounce_t0 = wall_clock64();
#endif
            {
                const BouncePacketKernargs &K = fresh(args);
                const uint32_t idx = opaque(idx0);
                D3 o = D3{0.0, 0.0, 0.0}, d = D3{0.0, 0.0, -1.0};
                float t0 = -__builtin_inff();  // a slot without a ray: an empty search interval
                int32_t shape0 = -1, org_shape = -1, org_tri = -1;
                if (going) {
                    if (TREE && b == 1) {  // a level of ray trees: the ray is made here
                        MirrorTmp made;
                        int depth;
                        tree_spawn_ray(K.S, K.B, K.T, idx, made, depth);
                        o = made.p;
                        d = made.d;
                        org_shape = made.shape;
                        org_tri = made.tri;
                        t0 = made.self_t;
                        keep[12 * RT_PACKET_BLOCK] = made.sample;
                        keep[13 * RT_PACKET_BLOCK] = depth;  // reflections in front of this ray
                        keep_ray(keep, o, d);
                    } else if (b == 1) {
                        queue_read_ray(K.Q, which, idx, o, d);
                        org_shape = K.Q.q_org_shape[which][idx];
                        org_tri = K.Q.q_org_tri[which][idx];
                        t0 = K.Q.q_self_t[which][idx];  // the origin primitive's own re-hit, or +inf
                        keep[12 * RT_PACKET_BLOCK] = K.Q.q_sample[which][idx];
                        keep[13 * RT_PACKET_BLOCK] = 1;  // reflections in front of this ray
                        keep_ray(keep, o, d);
                    } else {
                        kept_ray(keep, o, d);
                        org_shape = keep[14 * RT_PACKET_BLOCK];
                        org_tri = keep[15 * RT_PACKET_BLOCK];
                        t0 = __int_as_float(keep[16 * RT_PACKET_BLOCK]);
                    }
                    if (t0 < __builtin_inff()) shape0 = org_shape;
                }
                fast_set_ray(st[0], (float)o.x, (float)o.y, (float)o.z, (float)d.x, (float)d.y, (float)d.z);
                st[0].t = t0;
                st[0].shape = shape0;
                st[0].tri = -1;
                st[0].org_shape = org_shape;
                st[0].org_tri = org_tri;
            }
            ok = packet_traverse<false, 1>(S, st, closest_tmin(1), lds, rows - 1) && ok;  // (the walk's fields of S stay in registers)
#ifdef RT_WAVE_TIMELINE
            const unsigned long long bounce_t1 = wall_clock64();
            if (b <= 6) wave_log(2u * n_packets + (blk * 6u + (uint32_t)(b - 1)) * 2u, bounce_t0);
#endif
            const BouncePacketKernargs &L = fresh(args);
            MirrorTmp tmp;
            int act = RT_ACT_NONE;
            int32_t sample = 0;
            if (going) {
                D3 o, d;
                kept_ray(keep, o, d);
                sample = keep[12 * RT_PACKET_BLOCK];
                const int depth = keep[13 * RT_PACKET_BLOCK] + b - 1;  // what the ray started with plus the mirrors since
                act = shade_hit(L.S, L.Q, L.B, L.ids, depth, sample, o, d, st[0].shape, st[0].tri, st[0].t, tmp);
            }
            going = act == RT_ACT_MIRROR;
            if (going) {  // the reflected ray, for the next round
                keep_ray(keep, tmp.p, tmp.d);
                keep[14 * RT_PACKET_BLOCK] = tmp.shape;
                keep[15 * RT_PACKET_BLOCK] = tmp.tri;
                keep[16 * RT_PACKET_BLOCK] = __float_as_int(tmp.self_t);
            }
            const unsigned long long on = __ballot(going);
            if (tally_on != 0u && on != 0ull && threadIdx.x == 0u) tally[b & 31] += (uint32_t)__popcll(on);  // rays of bounce b + 1
            unsigned long long todo = __ballot(act == RT_ACT_LIT);
            const uint32_t chunk = act == RT_ACT_LIT ? (uint32_t)sample / RT_LIT_CHUNK : ~0u;
            while (todo != 0ull) {
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t of = (uint32_t)__shfl((int)chunk, leader, 64);
                const unsigned long long same = __ballot(chunk == of);
                wave_list_lit(L.Q, L.B.direct_lit != 0u, same, chunk == of, of, (uint32_t)sample);
                todo &= ~same;
            }
#ifdef RT_WAVE_TIMELINE
            if (b <= 6) wave_log(2u * n_packets + (blk * 6u + (uint32_t)(b - 1)) * 2u + 1u, bounce_t1);
#endif
        }
    }
    if (!ok) *fresh(args).Q.overflow = 1;
    __syncthreads();
    if (tally_on != 0u && threadIdx.x < 32u && tally[threadIdx.x] != 0u && RT_CNT_RAYS0 + threadIdx.x + 1 < RT_CNT_LIT)
        atomicAdd(&fresh(args).Q.counters[RT_CNT_RAYS0 + threadIdx.x + 1], tally[threadIdx.x]);
}

__global__ void __launch_bounds__(RT_PACKET_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_BOUNCE_PACKET_WAVES, RT_BOUNCE_PACKET_WAVES)))
k_bounces_packet(RtSceneView S, RtBuffers Q, RtBatch B, RtIdOut ids, uint32_t n_samples, uint32_t tally_on, TreeSpawn T) {
    extern __shared__ int32_t lds[];
    bounces_packet_body<false>(S, Q, n_samples, tally_on, lds);
}
#ifndef RT_BOUNCE_TREE_WAVES
#define RT_BOUNCE_TREE_WAVES 4  // (128 registers, no scratch: ggx_scene 2.14 ms; 5 waves, 96 registers, 88 bytes of scratch: 2.35)
#endif
__global__ void __launch_bounds__(RT_PACKET_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_BOUNCE_TREE_WAVES, RT_BOUNCE_TREE_WAVES)))
k_bounces_packet_tree(RtSceneView S, RtBuffers Q, RtBatch B, RtIdOut ids, uint32_t n_samples, uint32_t tally_on, TreeSpawn T) {
    extern __shared__ int32_t lds[];
    bounces_packet_body<true>(S, Q, n_samples, tally_on, lds);
}

// ---- scenes with a GGX shader: one thread walks its sample's whole ray tree (rt_tree.h).
// The traversal stack lives in LDS like the per-lane kernels'; the tree's own stack of waiting
// GGX hits is per-thread scratch memory.
template <bool FAST>
__global__ void __launch_bounds__(RT_BLOCK) k_render_tree(RtSceneView S, RtBuffers Q, RtBatch B, RtIdOut ids, uint32_t n_samples) {
    extern __shared__ int32_t lds[];
    LdsStack stk{lds + threadIdx.x, S.stack_depth};
    const uint32_t ls = blockIdx.x * RT_BLOCK + threadIdx.x;
    TreeCounts counts = {0ull, 0ull, 0ull};
    if (ls < n_samples) tree_body<LdsStack, FAST>(S, Q, B, ids, ls, stk, counts);
    // one atomic per wave and counter
    for (int d = 32; d > 0; d >>= 1) {
        counts.secondary += __shfl_down(counts.secondary, d, 64);
        counts.shadow += __shfl_down(counts.shadow, d, 64);
        counts.shaded += __shfl_down(counts.shaded, d, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&Q.counters[RT_CNT_TREE_SECONDARY]), counts.secondary);
        atomicAdd(reinterpret_cast<unsigned long long *>(&Q.counters[RT_CNT_TREE_SHADOW]), counts.shadow);
        atomicAdd(reinterpret_cast<unsigned long long *>(&Q.counters[RT_CNT_TREE_SHADED]), counts.shaded);
    }
}

// ---- scenes with a GGX shader, ray trees traced LEVEL BY LEVEL (fast mode, in a call that may wait):
// the rays of level L of all the batch's trees -- level 0 = the camera rays, level L + 1 = the reflected rays
// that the GGX hits of level L spawn -- go through the fast mode's stages as if they were a batch's samples:
// packet walk, the stage behind it (mirror chains included), shadow packets and light sums, colour into the
// ray's slot of the level's colour array.  A ray whose chain ends on a GGX surface leaves the hit in its slot
// (shade_hit's RT_ACT_GGX); the kernels below number the next level's rays (count -> scan -> link), the lane that
// walks a ray makes it from its parent's hit (tree_spawn_ray), and when the deepest level is done the children's
// colours are added up into their parents, level by level upwards, in the order of the reference's loop
// (ggx_mirror.rs:66-93).  Against the depth-first walk of one
// lane per tree (k_render_tree) the rays of a level are coherent enough for packets, no lane waits for a
// neighbour's deeper tree, and nothing lives in scratch memory.
struct TreeLevelView {
    uint32_t *off, *chunk_base, *parent;
    uint32_t *base, *count;               // of the level that spawns
};

// One block per chunk of RT_LIT_CHUNK nodes: how many rays each node spawns (its shader's `samples` where its
// chain ended on a GGX surface), the exclusive sums inside the chunk, the chunk's total.
__global__ void __launch_bounds__(RT_LIT_THREADS) k_tree_count(RtSceneView S, RtBuffers Q, TreeLevelView T, uint32_t n_nodes) {
    __shared__ uint32_t wave_sum[RT_LIT_THREADS / 64];
    const uint32_t first = blockIdx.x * RT_LIT_CHUNK + threadIdx.x * RT_LIT_ROUNDS;  // RT_LIT_ROUNDS consecutive nodes per thread
    uint32_t want[RT_LIT_ROUNDS], mine = 0;
    for (uint32_t r = 0; r < RT_LIT_ROUNDS; ++r) {
        const uint32_t node = first + r;
        const int32_t mark = node < n_nodes ? Q.g_mark[node] : -1;
        want[r] = mark >= 0 ? S.shaders[(uint32_t)mark >> 8].samples : 0u;
        mine += want[r];
    }
    uint32_t incl = mine;  // inclusive scan over the wave, then over the block's waves
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
        if ((int)(threadIdx.x & 63u) >= d) incl += v;
    }
    if ((threadIdx.x & 63u) == 63u) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < RT_LIT_THREADS / 64; ++w) {
        if (w < (threadIdx.x >> 6)) before += wave_sum[w];
        total += wave_sum[w];
    }
    uint32_t run = before + incl - mine;
    for (uint32_t r = 0; r < RT_LIT_ROUNDS; ++r) {
        if (first + r < n_nodes) T.off[first + r] = run;
        run += want[r];
    }
    if (threadIdx.x == 0) T.chunk_base[blockIdx.x] = total;
}

// One block: chunk_base[0 .. n_chunks) -> exclusive sums in place, the total behind them.
__global__ void __launch_bounds__(RT_LIT_THREADS) k_tree_scan(TreeLevelView T, uint32_t n_chunks) {
    __shared__ uint32_t part[RT_LIT_THREADS];
    const uint32_t per = (n_chunks + RT_LIT_THREADS - 1u) / RT_LIT_THREADS;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, n_chunks);
    uint32_t acc = 0;
    for (uint32_t b = lo; b < hi; ++b) acc += T.chunk_base[b];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 1; d < RT_LIT_THREADS; d <<= 1) {
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - acc;
    for (uint32_t b = lo; b < hi; ++b) {
        const uint32_t c = T.chunk_base[b];
        T.chunk_base[b] = run;
        run += c;
    }
    if (threadIdx.x == RT_LIT_THREADS - 1) T.chunk_base[n_chunks] = part[RT_LIT_THREADS - 1];
}

// One thread per node of the level that spawns: where its children start in the next level and how many they are
// (kept per level for the sums on the way back up), and each child's parent.  `room`: nodes the next level can hold;
// the host has checked the total against it before this launch.
__global__ void __launch_bounds__(RT_BLOCK) k_tree_link(RtSceneView S, RtBuffers Q, TreeLevelView T, uint32_t n_nodes, uint32_t *next_queue_counts,
                                                        uint32_t n_children) {
    const uint32_t node = blockIdx.x * RT_BLOCK + threadIdx.x;
    // the stretches of the next level's queue (RtBuffers::mirror_counts of that level): its rays fill them without
    // gaps (they are fewer than this level's nodes times the largest `samples`, far fewer than 1024 per node)
    const uint32_t next_chunks = (n_children + RT_LIT_CHUNK - 1u) / RT_LIT_CHUNK;
    for (uint32_t c = node; c < next_chunks; c += gridDim.x * RT_BLOCK)
        next_queue_counts[c] = min(n_children - c * RT_LIT_CHUNK, (uint32_t)RT_LIT_CHUNK);
    if (node >= n_nodes) return;
    const int32_t mark = Q.g_mark[node];
    const uint32_t want = mark >= 0 ? S.shaders[(uint32_t)mark >> 8].samples : 0u;
    const uint32_t base = T.chunk_base[node / RT_LIT_CHUNK] + T.off[node];
    T.base[node] = base;
    T.count[node] = want;
    for (uint32_t k = 0; k < want; ++k) T.parent[base + k] = node;
}

// One thread per node of a level whose children are done: ggx_mirror.rs:89-93, the children's colours added up in
// their order, divided by their number.
__global__ void __launch_bounds__(RT_BLOCK) k_tree_gather(const uint32_t *base, const uint32_t *count, const float *child_color, float *color, uint32_t n_nodes) {
    const uint32_t node = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (node >= n_nodes) return;
    const uint32_t n = count[node];
    if (n == 0u) return;
    const float *c = child_color + (size_t)base[node] * 3;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t k = 0; k < n; ++k) {
        r += c[3 * k + 0];
        g += c[3 * k + 1];
        b += c[3 * k + 2];
    }
    const float div = (float)n;
    color[(size_t)node * 3 + 0] = r / div;
    color[(size_t)node * 3 + 1] = g / div;
    color[(size_t)node * 3 + 2] = b / div;
}

// ---- compaction of the lit table into a list of sample indices, in sample order: the shade stage
// has counted the lit hits of every chunk of RT_LIT_CHUNK samples; an exclusive scan of the chunk counts
// and a scatter follow.  Large chunks keep the single-block scan in the middle short: it is pure
// latency on the frame's critical path.  The list itself is filled without atomics (thousands of
// waves taking slots from one counter serialise on it), so it comes out in sample order.
//
// A chunk is worked on by a 256-thread block in four rounds of 256 consecutive samples.  Blocks
// of 1024 threads need sixteen free wave slots on ONE compute unit at once; with several frames
// in flight the packet kernels of the other frames (one-wave blocks, eight per SIMD) rarely
// leave that many, and these small kernels -- which every frame waits for -- sat in the queue
// for hundreds of microseconds.


// One block: counters[RT_CNT_LIT] = sum of block_counts[0..n_blocks), for the statistics of a batch whose lit
// hits were listed directly (nothing on the device needs the total then).
__global__ void __launch_bounds__(RT_BLOCK) k_lit_total(RtBuffers Q, uint32_t n_blocks) {
    __shared__ uint32_t part[RT_BLOCK];
    uint32_t acc = 0;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += RT_BLOCK) acc += Q.block_counts[b];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t step = RT_BLOCK / 2; step > 0; step /= 2) {
        if (threadIdx.x < step) part[threadIdx.x] += part[threadIdx.x + step];
        __syncthreads();
    }
    if (threadIdx.x == 0) Q.counters[RT_CNT_LIT] = part[0];
}

// One block: block_counts[0..n_blocks) -> exclusive prefix sums in place, total -> counters[RT_CNT_LIT].
__global__ void __launch_bounds__(RT_LIT_THREADS) k_lit_scan(RtBuffers Q, uint32_t n_blocks) {
    __shared__ uint32_t part[RT_LIT_THREADS];
    const uint32_t per = (n_blocks + RT_LIT_THREADS - 1u) / RT_LIT_THREADS;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, n_blocks);
    uint32_t acc = 0;
    for (uint32_t b = lo; b < hi; ++b) acc += Q.block_counts[b];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 1; d < RT_LIT_THREADS; d <<= 1) {  // Hillis-Steele inclusive scan
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - acc;  // exclusive
    for (uint32_t b = lo; b < hi; ++b) {
        const uint32_t c = Q.block_counts[b];
        Q.block_counts[b] = run;
        run += c;
    }
    if (threadIdx.x == RT_LIT_THREADS - 1) Q.counters[RT_CNT_LIT] = part[RT_LIT_THREADS - 1];
}

// OWN_OFFSET: no scan has run; block_counts still holds the counts, and every block adds up those of the
// chunks in front of it by itself (the last one also leaves the total in counters[RT_CNT_LIT]).  For a grid
// of a thousand chunks -- an eighth of a frame -- that is four loads per thread and one launch fewer on the
// frame's critical path (every 1/8 part of the bunny frame: 0.1115 -> 0.109 ms).  For a whole frame's 8800
// chunks the sums move more data than the scatter itself: one frame at a time gains 0.5 %, four in flight
// lose 0.3 %, and the scan kernel stays.
#ifndef RT_LIT_OWN_OFFSET_MAX
#define RT_LIT_OWN_OFFSET_MAX 4608u
#endif
template <bool OWN_OFFSET>
__global__ void __launch_bounds__(RT_LIT_THREADS) k_lit_scatter(RtBuffers Q, uint32_t n_samples) {
    __shared__ uint32_t wave_total[RT_LIT_ROUNDS][RT_LIT_THREADS / 64];
    __shared__ uint32_t before[RT_LIT_THREADS / 64];
    const uint32_t wave = threadIdx.x >> 6;
    bool lit[RT_LIT_ROUNDS];
    uint32_t rank[RT_LIT_ROUNDS];
    for (uint32_t r = 0; r < RT_LIT_ROUNDS; ++r) {
        const uint32_t s = blockIdx.x * RT_LIT_BLOCK + r * RT_LIT_THREADS + threadIdx.x;
        lit[r] = s < n_samples && Q.l_shape[s] >= 0;
        const unsigned long long mask = __ballot(lit[r]);
        rank[r] = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if ((threadIdx.x & 63u) == 0) wave_total[r][wave] = (uint32_t)__popcll(mask);
    }
    if (OWN_OFFSET) {
        uint32_t acc = 0;
        for (uint32_t b = threadIdx.x; b < blockIdx.x; b += RT_LIT_THREADS) acc += Q.block_counts[b];
        for (int d = 32; d > 0; d >>= 1) acc += (uint32_t)__shfl_down((int)acc, d, 64);
        if ((threadIdx.x & 63u) == 0) before[wave] = acc;
    }
    __syncthreads();
    // samples of a chunk are ordered round by round, wave by wave inside a round
    uint32_t off;
    if (OWN_OFFSET) {
        off = 0;
        for (uint32_t w = 0; w < RT_LIT_THREADS / 64; ++w) off += before[w];
    } else {
        off = Q.block_counts[blockIdx.x];
    }
    for (uint32_t r = 0; r < RT_LIT_ROUNDS; ++r) {
        for (uint32_t w = 0; w < RT_LIT_THREADS / 64; ++w) {
            if (w == wave && lit[r]) Q.lit_list[off + rank[r]] = blockIdx.x * RT_LIT_BLOCK + r * RT_LIT_THREADS + threadIdx.x;
            off += wave_total[r][w];
        }
    }
    if (OWN_OFFSET && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) Q.counters[RT_CNT_LIT] = off;
}

// Work item = (point light blockIdx.y, lit entry): consecutive lanes are neighbouring
// samples aiming at the SAME light, so their shadow rays stay coherent.
template <bool ORDERED, bool ANYHIT, bool EXACT>
__global__ void __launch_bounds__(RT_BLOCK) k_trace_shadow(RtSceneView S, RtBuffers Q, uint32_t xcd_chunk) {
    extern __shared__ int32_t lds[];
    LdsStack stk{lds + threadIdx.x, S.stack_depth};
    const uint32_t n_lit = Q.counters[RT_CNT_LIT];
    const uint32_t n_blocks = (n_lit + RT_BLOCK - 1) / RT_BLOCK;  // blocks that have work; the rest leave
    if (blockIdx.x >= n_blocks) return;
    const uint32_t e = xcd_chunked_block(blockIdx.x, n_blocks, xcd_chunk) * RT_BLOCK + threadIdx.x;
    if (e >= n_lit) return;
    trace_shadow_body<LdsStack, ORDERED, ANYHIT, EXACT>(S, Q, blockIdx.y, Q.lit_list[e], stk);
}

__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_ACC_WAVES, RT_ACC_WAVES))) k_accumulate(RtSceneView S, RtBuffers Q, RtBatch B, RtIdOut ids) {
    const uint32_t n_lit = Q.counters[RT_CNT_LIT];
    const uint32_t e = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (e >= n_lit) return;
    accumulate_body(S, Q, B, ids, Q.lit_list[e]);
}

// The per-pixel sum of resolve_body, with the samples brought in through LDS.  A pixel's samples are
// consecutive floats (27 of them at 9 rays per pixel), so a lane that reads its own pixel's samples
// walks through memory with a stride of 108 bytes: every load instruction of the wave touches about a
// hundred cache lines, and the kernel ran at the rate the vector memory unit looks lines up (2.2 TB/s)
// instead of the rate HBM delivers them.  Here the wave loads the samples of its 64 pixels, which are
// consecutive in memory, as consecutive floats (RT_RESOLVE_ROUND samples per pixel at a time), puts them
// into a table in LDS whose row length is odd (no bank conflicts), and each lane then adds up its own
// row in the same order as before: same sums, bit for bit.
// One-wave blocks: with frames in flight this kernel runs beside the packet kernels of other frames, whose
// one-wave blocks take every wave slot the moment it frees; a 256-thread block needs a free slot on all four
// SIMDs of one compute unit at the same moment and waited for it (a 1/8 frame's resolve, 5 us alone, took 170 us
// with eight frames in flight, and its stream's next frame waited with it).
#define RT_RESOLVE_ROUND 9
#define RT_RESOLVE_ROW (3 * RT_RESOLVE_ROUND)  // odd
#define RT_RESOLVE_BLOCK 64
__global__ void __launch_bounds__(RT_RESOLVE_BLOCK) k_resolve(RtBuffers Q, RtBatch B, float *fb) {
    __shared__ float stage[RT_RESOLVE_BLOCK / 64][64 * RT_RESOLVE_ROW];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float *st = stage[wave];
    const uint32_t spp = B.n * B.n;
    const uint32_t n_groups = (B.n_pixels + 63u) / 64u;
    const float div = (float)(uint16_t)(B.n * B.n);
    for (uint32_t g0 = blockIdx.x * (RT_RESOLVE_BLOCK / 64); g0 < n_groups; g0 += gridDim.x * (RT_RESOLVE_BLOCK / 64)) {  // the same trips for the whole block
        const uint32_t group = g0 + wave;
        const uint32_t lp = group * 64u + lane;
        const bool mine = group < n_groups && lp < B.n_pixels;
        float r = 0.0f, g = 0.0f, b = 0.0f;
        for (uint32_t k0 = 0; k0 < spp; k0 += RT_RESOLVE_ROUND) {
            const uint32_t per_px = 3u * min(spp - k0, (uint32_t)RT_RESOLVE_ROUND);  // floats per pixel in this round
            // float f = lane + 64 m of the round belongs to pixel f / per_px of the group, element f % per_px
            const uint32_t step_px = 64u / per_px, step_el = 64u % per_px;
            uint32_t px = lane / per_px, el = lane - px * per_px;
            const uint32_t last_px = B.n_pixels - 1u;  // a pixel past the end reads the last one's samples; nobody adds them up
            if (per_px == RT_RESOLVE_ROW) {
                // a full round: all loads first, then all stores, so that the loads are in flight together.  (Float
                // f = lane + 64 m of the round lies at px * RT_RESOLVE_ROW + el = f of the table: no index to keep per
                // load, which had cost 27 registers and, at 148 in all, left three waves per SIMD to hide HBM behind.)
                float v[RT_RESOLVE_ROW];
#pragma unroll
                for (uint32_t m = 0; m < RT_RESOLVE_ROW; ++m) {
                    const uint32_t src_px = min(group * 64u + px, last_px);
                    v[m] = RT_STREAM_LD(Q.color[((size_t)src_px * spp + k0) * 3 + el]);
                    px += 64u / RT_RESOLVE_ROW;
                    el += 64u % RT_RESOLVE_ROW;
                    if (el >= RT_RESOLVE_ROW) { el -= RT_RESOLVE_ROW; px += 1; }
                }
#pragma unroll
                for (uint32_t m = 0; m < RT_RESOLVE_ROW; ++m) st[lane + 64u * m] = v[m];
            } else {
                for (uint32_t m = 0; m < per_px; ++m) {
                    const uint32_t src_px = min(group * 64u + px, last_px);
                    st[px * RT_RESOLVE_ROW + el] = RT_STREAM_LD(Q.color[((size_t)src_px * spp + k0) * 3 + el]);
                    px += step_px;
                    el += step_el;
                    if (el >= per_px) { el -= per_px; px += 1; }
                }
            }
            __syncthreads();
            if (mine)
                for (uint32_t e = 0; e < per_px; e += 3) {  // p-major, q-minor: sample k = p*n + q
                    r += st[lane * RT_RESOLVE_ROW + e + 0];
                    g += st[lane * RT_RESOLVE_ROW + e + 1];
                    b += st[lane * RT_RESOLVE_ROW + e + 2];
                }
            __syncthreads();
        }
        float *out;
        if (mine && resolve_address(B, lp, fb, out)) {
            out[0] = r / div;
            out[1] = g / div;
            out[2] = b / div;
        }
    }
}

// The same for more than RT_RESOLVE_ROUND samples per pixel, as long as two pixels' samples fit the table
// (rt_resolve_by_pixels): WHOLE pixels per round, as many as fit.  The round's floats are then one contiguous piece of
// memory, read line by line; in rounds of RT_RESOLVE_ROUND samples a pixel's piece is 108 bytes out of every 12 * spp,
// which touches every cache line twice with loads that are told not to keep it.  A lane per (pixel, channel) then adds
// up its row in sample order: the same sums, bit for bit.  (A kernel of its own: its loop bodies cost k_resolve, which
// every 9-rays-per-pixel frame runs, two waves per SIMD.)
inline bool rt_resolve_by_pixels(uint32_t spp) { return 3u * spp > RT_RESOLVE_ROW && ((3u * spp) | 1u) * 2u <= 64u * RT_RESOLVE_ROW; }
__global__ void __launch_bounds__(RT_RESOLVE_BLOCK) k_resolve_pixels(RtBuffers Q, RtBatch B, float *fb) {
    __shared__ float st[64 * RT_RESOLVE_ROW];
    const uint32_t lane = threadIdx.x;
    const uint32_t spp = B.n * B.n;
    const uint32_t n_groups = (B.n_pixels + 63u) / 64u;
    const float div = (float)(uint16_t)(B.n * B.n);
    const uint32_t row = 3u * spp;
    {
        const uint32_t pitch = row | 1u;  // odd: the lanes that add up different pixels' rows start in different banks
        const uint32_t per_round = min(64u, (64u * RT_RESOLVE_ROW) / pitch);
        const size_t all_floats = (size_t)B.n_pixels * row;
        const uint32_t step_px = 64u / row, step_el = 64u % row;
        for (uint32_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
            for (uint32_t p0 = 0; p0 < 64u; p0 += per_round) {
                const uint32_t n_px = min(per_round, 64u - p0), floats = n_px * row;
                const size_t base = ((size_t)group * 64u + p0) * row;
                float v[RT_RESOLVE_ROW];
#pragma unroll
                for (uint32_t m = 0; m < RT_RESOLVE_ROW; ++m) {  // 64 * RT_RESOLVE_ROW >= floats: every float of the round has a load
                    const uint32_t f = lane + 64u * m;
                    v[m] = f < floats ? RT_STREAM_LD(Q.color[min(base + f, all_floats - 1u)]) : 0.0f;
                }
                uint32_t px = lane / row, el = lane - px * row;
#pragma unroll
                for (uint32_t m = 0; m < RT_RESOLVE_ROW; ++m) {
                    if (lane + 64u * m < floats) st[px * pitch + el] = v[m];
                    px += step_px;
                    el += step_el;
                    if (el >= row) { el -= row; px += 1; }
                }
                __syncthreads();
                for (uint32_t u = lane; u < n_px * 3u; u += 64u) {  // (pixel, channel) pairs of the round
                    const uint32_t q = u / 3u, c = u - q * 3u;
                    const float *at = st + q * pitch + c;
                    float acc = 0.0f;
                    for (uint32_t k = 0; k < spp; ++k) acc += at[3u * k];  // p-major, q-minor: sample k = p*n + q
                    const uint32_t lp = group * 64u + p0 + q;
                    float *out;
                    if (lp < B.n_pixels && resolve_address(B, lp, fb, out)) out[c] = acc / div;
                }
                __syncthreads();
            }
        }
    }
}

__global__ void __launch_bounds__(RT_BLOCK) k_fill_i32(int32_t *p, size_t n, int32_t v) {
    for (size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * RT_BLOCK) p[i] = v;
}

// Scatters gathered tile-major part buffers back into the i + j*width layout.
// tile_where[k] = (owner part << 20) | slot of tile k inside that part; a part's block of tiles
// starts at part * pitch floats.  One-wave blocks: this kernel runs beside the packet kernels of
// other frames, whose one-wave blocks take every wave slot as it frees; a 256-thread block needs a
// free slot on all four SIMDs of one compute unit at the same moment and waits far longer for it.
#define RT_UNTILE_BLOCK 64
__global__ void __launch_bounds__(RT_UNTILE_BLOCK)
k_untile(const float *gathered, size_t pitch, float *fb, const uint32_t *tile_where, uint32_t width, uint32_t height,
         uint32_t tw, uint32_t th, uint32_t tiles_x, uint32_t parts, uint32_t unrotate) {
    const uint32_t n_px = width * height;
    for (uint32_t px = blockIdx.x * RT_UNTILE_BLOCK + threadIdx.x; px < n_px; px += gridDim.x * RT_UNTILE_BLOCK) {
        const uint32_t i = px % width, j = px / width;
        const uint32_t tile = (j / th) * tiles_x + i / tw;
        const uint32_t where = tile_where[tile];
        uint32_t block = (where >> 20) + unrotate;  // part p lies in block (p - rotation) mod parts
        block = block >= parts ? block - parts : block;
        const size_t src = (size_t)block * pitch + ((size_t)(where & 0xfffffu) * tw * th + (size_t)(j % th) * tw + i % tw) * 3;
        fb[(size_t)px * 3 + 0] = gathered[src + 0];
        fb[(size_t)px * 3 + 1] = gathered[src + 1];
        fb[(size_t)px * 3 + 2] = gathered[src + 2];
    }
}

// BVH::closest_hit on caller-supplied rays (rt_trace_rays).
template <bool ORDERED, bool EXACT>
__global__ void __launch_bounds__(RT_BLOCK)
k_trace_user(RtSceneView S, uint64_t n, const double *o, const double *d, const double *tmin, const double *tmax,
             float *t_out, int32_t *shape_out, int32_t *tri_out, uint32_t *overflow) {
    extern __shared__ int32_t lds[];
    LdsStack stk{lds + threadIdx.x, S.stack_depth};
    for (uint64_t r = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; r < n; r += (uint64_t)gridDim.x * RT_BLOCK) {
        HitF hit;
        hit.t = (float)tmax[r];
        hit.shape = -1;
        hit.tri = -1;
        RayF ray;
        ray_from_f64<EXACT>(D3{o[r * 3], o[r * 3 + 1], o[r * 3 + 2]}, D3{d[r * 3], d[r * 3 + 1], d[r * 3 + 2]}, ray);
        bool ok;
        if (EXACT) {
            ok = traverse<LdsStack, ORDERED, false, EXACT>(S, ray, (float)tmin[r], hit, -1, -1, stk, S.stack_depth);
        } else {
            FastTrav st;
            fast_begin(S, st, stk, D3{o[r * 3], o[r * 3 + 1], o[r * 3 + 2]}, D3{d[r * 3], d[r * 3 + 1], d[r * 3 + 2]}, hit.t, -1, -1, -1);
            bool overflow = false;
            while (fast_step<LdsStack, false>(S, st, stk, (float)tmin[r], S.stack_depth, overflow)) {
            }
            ok = !overflow;
            hit.t = st.t;
            hit.shape = st.shape;
            hit.tri = st.tri;
        }
        if (!ok) *overflow = 1;
        t_out[r] = hit.t;
        shape_out[r] = hit.shape;
        tri_out[r] = tri_file_index(S, hit.tri);
    }
}

// ------------------------------------------------------------------ host side

template <typename T>
int dev_upload(const std::vector<T> &v, T **out) {
    *out = nullptr;
    if (v.empty()) return RT_OK;
    RT_HIP(hipMalloc((void **)out, v.size() * sizeof(T)));
    RT_HIP(hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

void free_all(std::vector<void *> &v) {
    for (void *p : v)
        if (p) (void)hipFree(p);
    v.clear();
}

template <typename T>
int alloc_into(std::vector<void *> &owner, T **out, size_t count) {
    void *p = nullptr;
    RT_HIP(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    owner.push_back(p);
    *out = (T *)p;
    return RT_OK;
}

// Points the arrays of `Q` at memory for `capacity` samples and `n_point_lights` lights, taken piece by
// piece from `take(bytes)` (nullptr = out of memory).  Every array starts on a 256-byte boundary of
// whatever `take` hands out.
template <class Take>
bool assign_buffers(RtBuffers &Q, uint32_t capacity, uint32_t n_point_lights, Take take) {
    std::memset(&Q, 0, sizeof(Q));
    const size_t cap = std::max<uint32_t>(capacity, 1);
    const size_t npl = std::max<uint32_t>(n_point_lights, 1);
    bool ok = true;
    auto get = [&](auto **out, size_t count) {
        typedef std::remove_reference_t<decltype(**out)> T;
        *out = ok ? (T *)take(count * sizeof(T)) : nullptr;
        ok = ok && *out != nullptr;
    };
    for (int w = 0; w < 2; ++w) {
        for (int k = 0; k < 3; ++k) {
            get(&Q.q_o[w][k], cap);
            get(&Q.q_d[w][k], cap);
        }
        get(&Q.q_sample[w], cap);
        get(&Q.q_org_shape[w], cap);
        get(&Q.q_org_tri[w], cap);
        get(&Q.q_self_t[w], cap);
    }
    get(&Q.h_t, cap);
    get(&Q.h_shape, cap);
    get(&Q.h_tri, cap);
    for (int k = 0; k < 3; ++k) {
        get(&Q.l_p[k], cap);
        get(&Q.l_n[k], cap);
        get(&Q.l_d[k], cap);
    }
    get(&Q.l_shape, cap);
    get(&Q.l_tri, cap);
    get(&Q.l_self, cap);
    get(&Q.occ_shape, cap * npl);
    get(&Q.occ_tri, cap * npl);
    get(&Q.occ_t, cap * npl);
    get(&Q.color, cap * 3);
    // the counters and, behind them, the per-chunk lit counts (one memset clears both), twice: see counters_next
    const size_t set_words = ((size_t)RT_CNT_SIZE + 2 * (cap / RT_LIT_CHUNK + 1) + 63u) & ~(size_t)63u;
    get(&Q.counters, 2 * set_words);
    get(&Q.lit_list, cap);
    if (!ok) return false;
    Q.block_counts = Q.counters + RT_CNT_SIZE;
    Q.counters_next = Q.counters + set_words;
    Q.capacity = capacity;
    return true;
}

// Device bytes assign_buffers takes for a batch of `capacity` samples, every array padded to 256 bytes.
size_t buffer_bytes(uint32_t capacity, uint32_t n_point_lights) {
    size_t total = 0;
    RtBuffers Q;
    assign_buffers(Q, capacity, n_point_lights, [&](size_t bytes) {
        total += (bytes + 255u) & ~(size_t)255u;
        return (void *)(uintptr_t)256;  // any non-null value: nothing is dereferenced
    });
    return total;
}

int ensure_buffers(RtBackend *be, uint32_t capacity, uint32_t n_point_lights) {
    if (be->buf_capacity >= capacity && be->buf_point_lights >= n_point_lights) return RT_OK;
    free_all(be->buf_allocs);
    be->buf_capacity = 0;
    hipError_t err = hipSuccess;
    const bool ok = assign_buffers(be->buf, capacity, n_point_lights, [&](size_t bytes) -> void * {
        void *p = nullptr;
        err = hipMalloc(&p, std::max<size_t>(bytes, 1));
        if (err != hipSuccess) return nullptr;
        be->buf_allocs.push_back(p);
        return p;
    });
    if (!ok) {
        free_all(be->buf_allocs);
        return rt::set_error(RT_ERR_DEVICE, std::string("hipMalloc of the batch buffers: ") + hipGetErrorString(err));
    }
    be->buf.overflow = be->sticky;
    be->counter_dirty[0] = be->counter_dirty[1] = ~0u;  // fresh memory: nothing is known to be zero
    be->buf_capacity = capacity;
    be->buf_point_lights = n_point_lights;
    return RT_OK;
}

// The arrays of the level-by-level ray trees for `capacity` nodes per level and levels 0 .. `levels` - 1.
int ensure_tree_buffers(RtBackend *be, uint32_t capacity, int levels) {
    RtBackend::TreeLevels &t = be->tree;
    if (!t.host_words) RT_HIP(hipHostMalloc((void **)&t.host_words, 64 * sizeof(uint32_t), hipHostMallocDefault));
    if (!t.copied) RT_HIP(hipEventCreateWithFlags(&t.copied, hipEventDisableTiming));
    if (t.capacity < capacity) {
        free_all(t.allocs);
        uint32_t *const keep_words = t.host_words;
        const hipEvent_t keep_event = t.copied;
        t = RtBackend::TreeLevels();
        t.host_words = keep_words;
        t.copied = keep_event;
        const size_t cap = capacity;
        for (int w = 0; w < 2; ++w) {
            for (int k = 0; k < 3; ++k) {
                if (int rc = alloc_into(t.allocs, &t.g_p[w][k], cap)) return rc;
                if (int rc = alloc_into(t.allocs, &t.g_n[w][k], cap)) return rc;
                if (int rc = alloc_into(t.allocs, &t.g_in[w][k], cap)) return rc;
            }
            if (int rc = alloc_into(t.allocs, &t.g_shape[w], cap)) return rc;
            if (int rc = alloc_into(t.allocs, &t.g_tri[w], cap)) return rc;
            if (int rc = alloc_into(t.allocs, &t.g_mark[w], cap)) return rc;
        }
        if (int rc = alloc_into(t.allocs, &t.off, cap)) return rc;
        if (int rc = alloc_into(t.allocs, &t.chunk_base, cap / RT_LIT_CHUNK + 2)) return rc;
        if (int rc = alloc_into(t.allocs, &t.parent, cap)) return rc;
        for (int w = 0; w < 2; ++w) {
            if (int rc = alloc_into(t.allocs, &t.root[w], cap)) return rc;
            if (int rc = alloc_into(t.allocs, &t.path[w], cap)) return rc;
        }
        t.capacity = capacity;
    }
    while ((int)t.base.size() < levels) {
        float *color = nullptr;
        uint32_t *base = nullptr, *count = nullptr;
        if (!t.base.empty())  // (level 0's colours are the batch's)
            if (int rc = alloc_into(t.allocs, &color, (size_t)t.capacity * 3)) return rc;
        if (int rc = alloc_into(t.allocs, &base, t.capacity)) return rc;
        if (int rc = alloc_into(t.allocs, &count, t.capacity)) return rc;
        t.color.push_back(color);
        t.base.push_back(base);
        t.count.push_back(count);
    }
    return RT_OK;
}

uint32_t grid_for(uint64_t work) {
    const uint64_t blocks = (work + RT_BLOCK - 1) / RT_BLOCK;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(blocks, RT_MAX_GRID));
}

// Tiles owned by `part` of `parts` (rt_tile_map), ascending.
void owned_tiles(uint32_t tiles_x, uint32_t tiles_y, uint32_t part, uint32_t parts, uint32_t part0_weight,
                 std::vector<int32_t> &out) {
    out.clear();
    const RtTileMap map = rt_tile_map(tiles_x, tiles_y, parts, part0_weight);
    for (uint32_t ty = 0; ty < tiles_y; ++ty)
        for (uint32_t tx = 0; tx < tiles_x; ++tx)
            if (rt_tile_map_owner(map, tx, ty) == part) out.push_back((int32_t)(ty * tiles_x + tx));
}

// Tiles per part of the split (rt_tile_map).
std::vector<uint32_t> part_tile_counts(uint32_t tiles_x, uint32_t tiles_y, uint32_t parts, uint32_t part0_weight) {
    std::vector<uint32_t> per_part(std::max<uint32_t>(parts, 1), 0);
    const RtTileMap map = rt_tile_map(tiles_x, tiles_y, parts, part0_weight);
    for (uint32_t ty = 0; ty < tiles_y; ++ty)
        for (uint32_t tx = 0; tx < tiles_x; ++tx) per_part[rt_tile_map_owner(map, tx, ty)]++;
    return per_part;
}

// The pixels a render covers: RtRenderArgs' window, or the whole frame.
struct Window {
    uint32_t x0, y0, w, h;
};
int resolve_window(const RtRenderArgs *args, uint32_t W, uint32_t H, Window &win) {
    win = Window{0, 0, W, H};
    if (args->window_w == 0 || args->window_h == 0) return RT_OK;
    if ((uint64_t)args->window_x0 + args->window_w > W || (uint64_t)args->window_y0 + args->window_h > H)
        return rt::set_error(RT_ERR_INVALID, "the window sticks out of the " + std::to_string(W) + " x " + std::to_string(H) + " frame");
    if (args->part_count > 1) return rt::set_error(RT_ERR_INVALID, "a window is rendered whole (part_count <= 1)");
    win = Window{args->window_x0, args->window_y0, args->window_w, args->window_h};
    return RT_OK;
}

// Shared argument check of everything that splits a frame.
int check_partition(const RtRenderArgs *args) {
    const uint32_t parts = std::max<uint32_t>(args->part_count, 1);
    if (args->part_index >= parts) return rt::set_error(RT_ERR_INVALID, "part_index out of range");
    if (args->part0_weight > 8) return rt::set_error(RT_ERR_INVALID, "part0_weight must be 0 (equal shares) or 1..8");
    return RT_OK;
}

// The tiles laid over W x H pixels.  At most 2^26 of them: tiles are numbered in 32 bits and listed on the host.
int tile_grid(uint32_t W, uint32_t H, uint32_t tw, uint32_t th, uint32_t &tiles_x, uint32_t &tiles_y) {
    const uint64_t tx = ((uint64_t)W + tw - 1) / tw, ty = ((uint64_t)H + th - 1) / th;
    if (tx * ty > (1ull << 26)) return rt::set_error(RT_ERR_INVALID, "frame too large: more than 2^26 tiles (use larger tiles)");
    tiles_x = (uint32_t)tx;
    tiles_y = (uint32_t)ty;
    return RT_OK;
}

typedef RtEventPair EventPair;

struct Timing {
    Timing() = default;
    Timing(const Timing &) = delete;
    Timing &operator=(const Timing &) = delete;
    ~Timing() {  // pairs that were neither collected nor handed to the backend (an error return)
        for (auto &p : pairs) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
    }
    bool on = false;
    uint32_t mask = ~0u;  // kernel classes to time
    bool open = false;
    std::vector<EventPair> pairs;
    hipStream_t stream = nullptr;
    int begin(int klass) {
        open = on && ((mask >> klass) & 1u) != 0u;
        if (!open) return 0;
        EventPair p;
        p.klass = klass;
        if (hipEventCreate(&p.a) != hipSuccess) return -1;
        if (hipEventCreate(&p.b) != hipSuccess) {
            (void)hipEventDestroy(p.a);
            return -1;
        }
        (void)hipEventRecord(p.a, stream);
        pairs.push_back(p);
        return 0;
    }
    void end() {
        if (open) (void)hipEventRecord(pairs.back().b, stream);
        open = false;
    }
    void collect(RtStats *st) {
        for (auto &p : pairs) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                st->kernel_ms[p.klass] += ms;
                st->kernel_launches[p.klass] += 1;
            }
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
        pairs.clear();
    }
};

}  // namespace

int rt::finish_stream(RtBackend *be, hipStream_t stream) {
    RT_HIP(hipMemcpyAsync(be->sticky_host, be->sticky, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    RT_HIP(hipStreamSynchronize(stream));
    RT_HIP(hipGetLastError());
    if (*be->sticky_host != 0u) {
        *be->sticky_host = 0u;
        RT_HIP(hipMemsetAsync(be->sticky, 0, sizeof(uint32_t), stream));
        return rt::set_error(RT_ERR_DEVICE, "traversal stack overflow (internal error)");
    }
    return RT_OK;
}

int rt::ensure_device_fb(RtBackend *be, size_t floats) {
    if (be->fb_floats >= floats) return RT_OK;
    if (be->fb) (void)hipFree(be->fb);
    be->fb = nullptr;
    be->fb_floats = 0;
    RT_HIP(hipMalloc((void **)&be->fb, floats * sizeof(float)));
    be->fb_floats = floats;
    return RT_OK;
}

int rt::ensure_pinned(RtBackend *be, size_t floats) {
    if (be->pinned_floats >= floats) return RT_OK;
    if (be->pinned) (void)hipHostFree(be->pinned);
    be->pinned = nullptr;
    be->pinned_floats = 0;
    RT_HIP(hipHostMalloc((void **)&be->pinned, floats * sizeof(float), hipHostMallocDefault));
    be->pinned_floats = floats;
    return RT_OK;
}

bool rt::is_pinned_host(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory is unknown to the runtime: that is the answer, not a failure
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

namespace {

// What a frame's arguments come to: checked once (plan_frame), read by every step of rt::render_frame.
struct FramePlan {
    Window win{};                   // the rendered pixels: RtRenderArgs' window, else the whole frame
    uint32_t W = 0, H = 0;          // win.w x win.h, which the tiles are laid over
    uint32_t tw = 0, th = 0, tiles_x = 0, tiles_y = 0;
    uint32_t n = 0, spp = 0;        // sqrt_rays_per_pixel and its square
    uint64_t tile_samples = 0;
    uint32_t parts = 1;             // the frame's split ...
    uint32_t span = 1;              // ... how many of its parts this call renders (RtRenderArgs::part_span) ...
    uint32_t pitch_tiles = 0;       // ... and, for more than one, the distance between their compact blocks in tiles
    // dynamic LDS: per-lane kernels (stack rows + auxiliary rows per thread); packet kernels (stack rows and the
    // world-space rays of a wave inside an instance, plus the fused shadow kernel's seven words per lane / the
    // camera kernel's plane point / the bounce kernel's tally and kept ray)
    size_t lds_bytes = 0, packet_lds = 0, camera_lds = 0, bounce_lds = 0;
    // which kernels a batch goes through
    bool ref_order = false;         // RT_MODE_REFERENCE_ORDER: the separate per-lane kernels throughout
    bool use_packets = false;       // fast mode: wave packets for camera and shadow rays, unless the caller asks for the per-lane kernels
    bool closest_blockers = false;  // fast mode with shadow rays that run to the closest blocker, so that their ids compare with
                                    // the reference's: the separate shadow kernel, which keeps a blocker per ray and light
    bool fused = false;             // shadow rays and light sums in one kernel: needs the packet path and a bit per point light
    bool fuse_camera = false;       // camera rays and the stage behind them in one kernel: needs the packet path
    bool fuse_bounce = false;       // all mirror bounces (their rays and the stage behind them) in one kernel
    // The kernels that find the lit hits list them as they go, each chunk of samples in a stretch of its own, and the
    // fused shadow kernel takes the stretches as they are: no scan and no scatter launch between them.  (Every other
    // combination of kernels goes through the compact, ascending list.)
    bool direct_lit = false;
    // Ray trees level by level (see k_tree_count): fast mode with all its fused kernels.  The call waits for the device
    // between levels (it reads how many rays the next level has), enqueue-only entry points too: a frame takes a
    // quarter of the depth-first kernel's time, which is worth more to a stream of frames than a call that returns
    // at once (RT_RENDER_TREE_DEPTH_FIRST is for callers who must not wait; the static work memory, whose users
    // may be capturing a graph, always goes depth first).
    bool tree_levels = false;
    int max_bounce = 0;
    // a wait per batch: to read the counters for the statistics, or to serve the progress callback
    bool serve_progress = false, wait_batches = false;
};

int plan_frame(const RtBackend *be, const RtDeviceScene *ds, const RtRenderArgs *args, const float *fb_dev, bool with_stats,
               bool blocking, bool static_memory, FramePlan &P) {
    if (!be || !ds || !args || !fb_dev) return rt::set_error(RT_ERR_INVALID, "null argument");
    if (ds->device != be->device) return rt::set_error(RT_ERR_INVALID, "scene was uploaded to another GPU than the backend's");
    if (args->antialias_method < RT_AA_NORMAL || args->antialias_method > RT_AA_RANDOM)
        return rt::set_error(RT_ERR_INVALID, "antialias_method must be RT_AA_NORMAL, RT_AA_JITTERED or RT_AA_RANDOM");
    P.n = args->sqrt_rays_per_pixel;
    if (P.n == 0 || P.n > 255) return rt::set_error(RT_ERR_INVALID, "sqrt_rays_per_pixel must be in 1..255");
    if (args->mode != RT_MODE_FAST && args->mode != RT_MODE_REFERENCE_ORDER)
        return rt::set_error(RT_ERR_INVALID, "unknown traversal mode");
    P.parts = std::max<uint32_t>(args->part_count, 1);
    if (int bad = check_partition(args)) return bad;
    P.tw = args->tile_w ? args->tile_w : 32;
    P.th = args->tile_h ? args->tile_h : 32;
    if (int bad = resolve_window(args, ds->width, ds->height, P.win)) return bad;
    P.W = P.win.w;
    P.H = P.win.h;
    if (int bad = tile_grid(P.W, P.H, P.tw, P.th, P.tiles_x, P.tiles_y)) return bad;
    P.spp = P.n * P.n;
    P.tile_samples = (uint64_t)P.tw * P.th * P.spp;
    if (P.tile_samples > (1ull << 30)) return rt::set_error(RT_ERR_INVALID, "tile too large");
    if (ds->view.recursion_depth > 30) return rt::set_error(RT_ERR_INVALID, "recursion depth above 30");

    P.ref_order = args->mode == RT_MODE_REFERENCE_ORDER;
    const RtSceneView &S = P.ref_order ? ds->view : ds->view_fast;
    P.lds_bytes = (size_t)(std::max(S.stack_depth, 1) + RT_LDS_AUX_ROWS) * RT_BLOCK * sizeof(int32_t);
    if (P.lds_bytes > 64 * 1024) return rt::set_error(RT_ERR_UNSUPPORTED, "scene trees too deep for the LDS traversal stack");
    const size_t packet_rows = (size_t)rt_packet_lds_dwords(std::max(S.stack_depth, 1) + 1);
    P.packet_lds = (packet_rows + 7u * 64u) * sizeof(int32_t);
    P.camera_lds = (packet_rows + 4u * 64u) * sizeof(int32_t);
    P.bounce_lds = (packet_rows + (1u + RT_BOUNCE_KEEP) * 64u) * sizeof(int32_t);

    if (static_memory && (P.parts > 1 || with_stats))
        return rt::set_error(RT_ERR_INVALID, "a render in static memory takes whole frames (part_count <= 1) and no statistics");
    // Several parts in one set of launches (RtRenderArgs::part_span): the tiles of all of them, in image order, each
    // with the place it goes to -- part (part_index + j) % parts writes its compact block j * pitch tiles into the
    // framebuffer.  One part: slot k of the list goes to place k.
    P.span = P.parts > 1 ? std::max<uint32_t>(args->part_span, 1) : 1;
    if (P.span > P.parts) return rt::set_error(RT_ERR_INVALID, "part_span exceeds part_count");
    if (P.span > 1 && static_memory) return rt::set_error(RT_ERR_INVALID, "a render in static memory takes whole frames");
    if (P.span > 1) {
        const std::vector<uint32_t> counts = part_tile_counts(P.tiles_x, P.tiles_y, P.parts, args->part0_weight);
        const uint32_t most = *std::max_element(counts.begin(), counts.end());
        const uint64_t tile_floats = (uint64_t)P.tw * P.th * 3;
        if (args->part_span_pitch_floats % tile_floats != 0)
            return rt::set_error(RT_ERR_INVALID, "part_span_pitch_floats must be a whole number of tiles");
        P.pitch_tiles = args->part_span_pitch_floats ? (uint32_t)(args->part_span_pitch_floats / tile_floats) : most;
        if (P.pitch_tiles < most) return rt::set_error(RT_ERR_INVALID, "part_span_pitch_floats smaller than a part's block of tiles");
    }

    const uint32_t flags = args->flags;
    P.use_packets = (flags & RT_RENDER_NO_PACKETS) == 0u;
    P.closest_blockers = !P.ref_order && P.use_packets && (flags & RT_RENDER_CLOSEST_BLOCKERS) != 0u;
    P.fused = !P.ref_order && P.use_packets && ds->n_point_lights <= 32 && (flags & RT_RENDER_UNFUSED_SHADOWS) == 0u && !P.closest_blockers;
    P.fuse_camera = !P.ref_order && P.use_packets && (flags & RT_RENDER_UNFUSED_CAMERA) == 0u;
    P.fuse_bounce = !P.ref_order && (flags & RT_RENDER_UNFUSED_CAMERA) == 0u;
    P.direct_lit = P.fused && P.fuse_camera && P.fuse_bounce;
    P.tree_levels = ds->has_ggx && !P.ref_order && !static_memory && ds->n_point_lights <= 32 &&
                    (flags & (RT_RENDER_TREE_DEPTH_FIRST | RT_RENDER_NO_PACKETS | RT_RENDER_UNFUSED_SHADOWS | RT_RENDER_UNFUSED_CAMERA |
                              RT_RENDER_CLOSEST_BLOCKERS)) == 0u;
    P.max_bounce = ds->has_mirror ? S.recursion_depth : 0;
    P.serve_progress = (blocking || with_stats) && args->progress != nullptr;
    P.wait_batches = with_stats || P.serve_progress;
    return RT_OK;
}

// The backend's list of the tiles this call renders (and, for several parts, of the places they go to), kept from
// call to call for as long as the partition stays the same.
int refresh_tile_list(RtBackend *be, const RtRenderArgs *args, const FramePlan &P) {
    const uint32_t key[8] = {P.W, P.H, P.tw, P.th, args->part_index, P.parts | (args->part0_weight << 24), P.span, P.pitch_tiles};
    if (std::memcmp(key, be->tile_key, sizeof(key)) == 0 && be->tile_list) return RT_OK;
    std::vector<int32_t> &tiles = be->tile_list_host, &places = be->tile_place_host;
    places.clear();
    if (P.span <= 1) {
        owned_tiles(P.tiles_x, P.tiles_y, args->part_index, P.parts, args->part0_weight, tiles);
    } else {
        std::vector<std::pair<int32_t, int32_t>> all;  // (tile, place)
        std::vector<int32_t> one;
        for (uint32_t j = 0; j < P.span; ++j) {
            owned_tiles(P.tiles_x, P.tiles_y, (args->part_index + j) % P.parts, P.parts, args->part0_weight, one);
            for (size_t k = 0; k < one.size(); ++k) all.emplace_back(one[k], (int32_t)(j * P.pitch_tiles + k));
        }
        std::sort(all.begin(), all.end());  // image order: neighbouring tiles are worked on together
        tiles.clear();
        for (const auto &e : all) {
            tiles.push_back(e.first);
            places.push_back(e.second);
        }
    }
    if (be->tile_list) (void)hipFree(be->tile_list);
    be->tile_list = nullptr;
    // the tiles, and behind them their places if they have any
    const size_t count = tiles.size();
    RT_HIP(hipMalloc((void **)&be->tile_list, std::max<size_t>(2 * count, 1) * sizeof(int32_t)));
    if (count) RT_HIP(hipMemcpy(be->tile_list, tiles.data(), count * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!places.empty()) RT_HIP(hipMemcpy(be->tile_list + count, places.data(), count * sizeof(int32_t), hipMemcpyHostToDevice));
    std::memcpy(be->tile_key, key, sizeof(key));
    return RT_OK;
}

// As many tiles per batch as the device's static array holds (at most `most`), by bisection: the bytes grow with the
// tiles, but one tile's bytes are no guide to it -- every array is padded to 256 bytes and the counter sets have a
// fixed part, some 13 KB that a batch pays once, not once per tile.
int static_batch_slots(const RtDeviceScene *ds, const FramePlan &P, uint64_t most, uint32_t &fit) {
    auto fits = [&](uint64_t tiles) { return buffer_bytes((uint32_t)(tiles * P.tile_samples), ds->n_point_lights) <= (size_t)RT_STATIC_SCRATCH_BYTES; };
    fit = 0;
    if (fits(1)) {
        uint64_t lo = 1, hi = most;  // lo fits
        while (lo < hi) {
            const uint64_t mid = (lo + hi + 1) / 2;
            if (fits(mid)) lo = mid;
            else hi = mid - 1;
        }
        fit = (uint32_t)lo;
    }
    if (fit == 0)
        return rt::set_error(RT_ERR_INVALID, "one tile of " + std::to_string(P.tile_samples) + " samples with " + std::to_string(ds->n_point_lights) +
                                                 " point lights needs " + std::to_string(buffer_bytes((uint32_t)P.tile_samples, ds->n_point_lights)) +
                                                 " bytes of work memory; the static array holds " + std::to_string((size_t)RT_STATIC_SCRATCH_BYTES) +
                                                 ": use smaller tiles or fewer rays per pixel");
    return RT_OK;
}

// One batch of ray trees, level by level (see k_tree_count): level 0 = the batch's camera rays, level L + 1 = what the
// GGX hits of level L spawn, each level through the fast mode's fused kernels with `Q`'s arrays as its work memory;
// then the children's colours into their parents, deepest level first, which leaves the samples' colours in Q.color.
// `too_large`: a level has more rays than Q holds and nothing of the batch is to be used (the caller renders the
// same tiles in smaller batches).  `secondary`, `lit`: what the levels' counters came to (with `stats`).
struct TreeBatch {
    bool too_large = false;
    uint64_t secondary = 0, lit = 0;
};
int tree_levels_batch(RtBackend *be, const RtDeviceScene *ds, const RtRenderArgs *args, const FramePlan &P, hipStream_t stream,
                      bool stats, Timing &tm, const RtSceneView &S, RtBuffers Q, const RtBatch &B, const RtIdOut &io, uint32_t n_samples,
                      TreeBatch &out) {
    const size_t camera_lds = P.camera_lds, packet_lds = P.packet_lds, bounce_lds = P.bounce_lds;
    int rc = RT_OK;
    // ---- level 0: the camera rays; level L + 1: what the GGX hits of level L spawn (see k_tree_count)
    RtBackend::TreeLevels &tl = be->tree;
    const uint32_t room = Q.capacity;
    uint32_t *const sets[2] = {Q.counters < Q.counters_next ? Q.counters : Q.counters_next,
                               Q.counters < Q.counters_next ? Q.counters_next : Q.counters};
    std::vector<uint32_t> level_nodes;  // rays per level
    bool too_large = false;
    uint64_t batch_secondary = 0, batch_lit = 0;
    uint32_t nodes = n_samples;
    for (int level = 0;; ++level) {
        // this level's rays through the stages
        RtBuffers L = Q;
        const uint32_t chunks = (nodes + RT_LIT_CHUNK - 1) / RT_LIT_CHUNK;
        L.counters = sets[level & 1];
        L.block_counts = L.counters + RT_CNT_SIZE;
        L.mirror_counts = L.block_counts + chunks;
        L.counters_next = sets[(level & 1) ^ 1];
        L.color = level == 0 ? Q.color : tl.color[(size_t)level];
        // where this level's rays leave their GGX hits, and (levels >= 1) the hits of the level above, which they are made from
        const int mine = level & 1, above = mine ^ 1;
        for (int k = 0; k < 3; ++k) {
            L.g_p[k] = tl.g_p[mine][k];
            L.g_n[k] = tl.g_n[mine][k];
            L.g_in[k] = tl.g_in[mine][k];
        }
        L.g_shape = tl.g_shape[mine]; L.g_tri = tl.g_tri[mine]; L.g_mark = tl.g_mark[mine];
        TreeSpawn from{};
        if (level > 0) {
            from.parent = tl.parent;
            from.base = tl.base[(size_t)level - 1];
            for (int k = 0; k < 3; ++k) {
                from.p[k] = tl.g_p[above][k];
                from.n[k] = tl.g_n[above][k];
                from.in[k] = tl.g_in[above][k];
            }
            from.shape = tl.g_shape[above]; from.tri = tl.g_tri[above]; from.mark = tl.g_mark[above];
            from.root_in = tl.root[above]; from.path_in = tl.path[above];
            from.root_out = tl.root[mine]; from.path_out = tl.path[mine];
            from.level = (uint32_t)level;
        }
        const uint32_t grid_packets = (nodes + RT_PACKET_BLOCK - 1) / RT_PACKET_BLOCK;
        const uint32_t grid_threads = (nodes + RT_BLOCK - 1) / RT_BLOCK;
        uint32_t xcd_chunk = args->xcd_chunk ? args->xcd_chunk : std::min(128u, std::max(8u, grid_threads / 256u));
        while ((xcd_chunk & (xcd_chunk - 1u)) != 0u) xcd_chunk &= xcd_chunk - 1u;
        const RtIdOut level_io = level == 0 ? io : RtIdOut{};
        // (the counters of a level >= 1 were cleared before k_tree_link wrote its queue's stretch lengths into them)
        if (level == 0) RT_HIP(hipMemsetAsync(L.counters, 0, ((size_t)RT_CNT_SIZE + 2u * chunks) * sizeof(uint32_t), stream));
        RT_HIP(hipMemsetAsync(L.g_mark, 0xff, (size_t)nodes * sizeof(int32_t), stream));
        if (tm.begin(RT_K_TRACE_CLOSEST)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
        if (level == 0)
            hipLaunchKernelGGL(k_camera_shade_packet, dim3(grid_packets), dim3(RT_PACKET_BLOCK), camera_lds, stream, S, L, B, level_io, nodes, xcd_chunk, 0u);
        if (level > 0 || ds->has_mirror) {
            const uint32_t gp = stats ? std::min<uint32_t>(grid_packets, 256u * 4u * RT_BOUNCE_PACKET_RESIDENT) : grid_packets;
            if (level == 0)
                hipLaunchKernelGGL(k_bounces_packet, dim3(gp), dim3(RT_PACKET_BLOCK), bounce_lds, stream, S, L, B, level_io, nodes, stats ? 1u : 0u, from);
            else
                hipLaunchKernelGGL(k_bounces_packet_tree, dim3(gp), dim3(RT_PACKET_BLOCK), bounce_lds, stream, S, L, B, level_io, nodes, stats ? 1u : 0u, from);
        }
        tm.end();
        level_nodes.push_back(nodes);
        // how many rays the GGX hits of this level spawn
        if ((rc = ensure_tree_buffers(be, room, level + 2))) return rc;
        TreeLevelView T{};
        T.off = tl.off; T.chunk_base = tl.chunk_base; T.parent = tl.parent;
        T.base = tl.base[(size_t)level]; T.count = tl.count[(size_t)level];
        if (tm.begin(RT_K_SHADE)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
        hipLaunchKernelGGL(k_tree_count, dim3(chunks), dim3(RT_LIT_THREADS), 0, stream, S, L, T, nodes);
        hipLaunchKernelGGL(k_tree_scan, dim3(1), dim3(RT_LIT_THREADS), 0, stream, T, chunks);
        tm.end();
        // The count travels to page-locked memory and the host waits for that copy alone: the level's shadow rays
        // and light sums, which the count does not depend on, are enqueued behind the copy and keep the GPU
        // busy while the host learns the count and enqueues the next level.
        RT_HIP(hipMemcpyAsync(tl.host_words, T.chunk_base + chunks, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        RT_HIP(hipEventRecord(tl.copied, stream));
        if (ds->has_lit) {
            if (stats) hipLaunchKernelGGL(k_lit_total, dim3(1), dim3(RT_BLOCK), 0, stream, L, chunks);
            if (tm.begin(RT_K_TRACE_SHADOW)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
            hipLaunchKernelGGL(k_shadow_shade_packet, dim3(grid_packets), dim3(RT_PACKET_BLOCK), packet_lds, stream, S, L, B, level_io, xcd_chunk);
            tm.end();
        }
        uint32_t level_counters[RT_CNT_SIZE];
        if (stats) {
            RT_HIP(hipMemcpyAsync(level_counters, L.counters, sizeof(level_counters), hipMemcpyDeviceToHost, stream));
            RT_HIP(hipStreamSynchronize(stream));
        } else {
            RT_HIP(hipEventSynchronize(tl.copied));
        }
        const uint32_t next_nodes = tl.host_words[0];
        if (stats) {
            for (int b = 1; b < 32; ++b) batch_secondary += level_counters[RT_CNT_RAYS0 + b];
            batch_lit += level_counters[RT_CNT_LIT];
        }
        if (next_nodes == 0u) break;
        if (next_nodes > room) {
            too_large = true;
            break;
        }
        // the next level: its counters, where each of its rays hangs, how long the stretches of its queue are
        const uint32_t next_chunks = (next_nodes + RT_LIT_CHUNK - 1) / RT_LIT_CHUNK;
        uint32_t *const next_counters = sets[(level & 1) ^ 1];
        RT_HIP(hipMemsetAsync(next_counters, 0, ((size_t)RT_CNT_SIZE + 2u * next_chunks) * sizeof(uint32_t), stream));
        if (tm.begin(RT_K_SHADE)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
        hipLaunchKernelGGL(k_tree_link, dim3(grid_threads), dim3(RT_BLOCK), 0, stream, S, L, T, nodes, next_counters + RT_CNT_SIZE + next_chunks, next_nodes);
        tm.end();
        nodes = next_nodes;
    }
    out.secondary = batch_secondary;
    out.lit = batch_lit;
    out.too_large = too_large;
    if (too_large) return RT_OK;
    // the children's colours into their parents, deepest level first
    if (tm.begin(RT_K_ACCUMULATE)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
    for (int level = (int)level_nodes.size() - 2; level >= 0; --level)
        hipLaunchKernelGGL(k_tree_gather, dim3((level_nodes[(size_t)level] + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, stream,
                           tl.base[(size_t)level], tl.count[(size_t)level], tl.color[(size_t)level + 1],
                           level == 0 ? Q.color : tl.color[(size_t)level], level_nodes[(size_t)level]);
    tm.end();
    return RT_OK;
}


// What a batch's kernels are told about it: the tiles `slot0 .. slot0 + slots - 1` of the call's list.
RtBatch describe_batch(const RtBackend *be, const RtDeviceScene *ds, const RtRenderArgs *args, const FramePlan &P, bool static_memory,
                       uint32_t slot0, uint32_t slots) {
    RtBatch B{};
    // (a render that owns every tile needs no list: slot k is tile k, and every wave is spared a fetch that its
    // first arithmetic would have had to wait for)
    B.tile_list = static_memory || P.parts <= 1 ? nullptr : be->tile_list;
    B.tile_place = !static_memory && !be->tile_place_host.empty() ? be->tile_list + be->tile_list_host.size() : nullptr;
    B.slot0 = slot0;
    B.n_pixels = slots * P.tw * P.th;
    B.tile_w = P.tw; B.tile_h = P.th; B.tiles_x = P.tiles_x;
    B.width = ds->width; B.height = ds->height;
    B.win_x0 = P.win.x0; B.win_y0 = P.win.y0; B.win_w = P.win.w; B.win_h = P.win.h;
    B.n = P.n;
    B.compact_fb = P.parts > 1 ? 1u : 0u;
    B.fused_shadows = P.fused ? 1u : 0u;
    B.direct_lit = P.direct_lit ? 1u : 0u;
    B.chunked_queue = P.fuse_camera && P.fuse_bounce ? 1u : 0u;  // k_camera_shade_packet, then k_bounces_packet
    B.light_sums_f32 = P.ref_order ? 0u : 1u;
    B.aa_method = args->antialias_method;
    B.seed = args->seed;
    if ((args->flags & RT_RENDER_ROW_MAJOR_PIXELS) == 0u) rt_pixel_block(P.tw, P.th, P.spp, B.blk_w_log2, B.blk_h_log2);
    rt_batch_finish(B);
    return B;
}

// Counters and per-chunk lit counts start at zero.  A backend's batches take turns with two sets of them: where the
// batch's first kernel is a camera-ray packet kernel (`clear_in_kernel`), that kernel zeroes the set the previous
// batch used (for the next one), so that no memset stands in front of a frame; any other batch clears its own set with
// a memset.  (The static work array is shared by all backends of a device, so nothing is known about its sets from
// one call to the next: always a memset.)  Points Q's counter fields at the batch's set; `n_clear` = the words of the
// other set that the batch's first kernel is to zero.
int take_counter_set(RtBackend *be, RtBuffers &Q, uint32_t lit_grid, bool clear_in_kernel, bool static_memory, hipStream_t stream,
                     uint32_t &n_clear) {
    const uint32_t n_counters = (uint32_t)RT_CNT_SIZE + 2u * lit_grid;  // counters, lit hits per chunk, mirror rays per chunk
    uint32_t *const sets[2] = {Q.counters < Q.counters_next ? Q.counters : Q.counters_next,
                               Q.counters < Q.counters_next ? Q.counters_next : Q.counters};
    const uint32_t set_words = (uint32_t)(sets[1] - sets[0]);
    uint32_t dirty_static[2] = {~0u, ~0u};
    uint32_t *dirty = static_memory ? dirty_static : be->counter_dirty;  // per set: how many leading words may be non-zero
    // a clean set if there is one and this batch keeps the turns going, else a dirty one (and the memset)
    int use = 0;
    if (clear_in_kernel) use = dirty[0] == 0u ? 0 : dirty[1] == 0u ? 1 : 0;
    else use = dirty[0] == 0u && dirty[1] != 0u ? 1 : 0;
    if (dirty[use] != 0u) RT_HIP(hipMemsetAsync(sets[use], 0, (size_t)n_counters * sizeof(uint32_t), stream));
    dirty[use] = std::max(dirty[use], n_counters);
    n_clear = 0;
    if (clear_in_kernel) {
        n_clear = std::min(dirty[use ^ 1], set_words);
        dirty[use ^ 1] = 0u;
    }
    Q.counters = sets[use];
    Q.block_counts = Q.counters + RT_CNT_SIZE;
    Q.mirror_counts = Q.block_counts + lit_grid;
    Q.counters_next = sets[use ^ 1];
    return RT_OK;
}

// The grids of a batch of `n_samples` samples.
struct BatchGrids {
    uint32_t full;       // a thread per sample (worst case for the per-lane kernels; blocks without work return at once)
    uint32_t packets;    // a wave per 64 samples: one ray per lane (two per lane -- the packet kernels' R -- halve the steps per ray but were
                         // slower in every measurement, bunny camera rays 0.36 -> 0.49 ms, shadow rays 0.88 -> 1.13: twice the registers, a larger union of nodes)
    uint32_t lit;        // a block per chunk of the lit list
    uint32_t xcd_chunk;  // chunk of the XCD remap in 256-sample blocks, a power of two (the kernels shift and mask)
};
BatchGrids batch_grids(const RtRenderArgs *args, uint32_t n_samples) {
    BatchGrids g;
    g.full = (n_samples + RT_BLOCK - 1) / RT_BLOCK;
    g.packets = (n_samples + RT_PACKET_BLOCK - 1) / RT_PACKET_BLOCK;
    g.lit = (n_samples + RT_LIT_CHUNK - 1) / RT_LIT_CHUNK;
    // 128 (a few tiles) for a whole frame, smaller for a small launch so that every XCD still gets >= 32 chunks --
    // chunks differ in cost, and with only a handful each the XCDs finish far apart (a 1/8 frame: 0.272 -> 0.254 ms)
    g.xcd_chunk = args->xcd_chunk ? args->xcd_chunk : std::min(128u, std::max(8u, g.full / 256u));
    while ((g.xcd_chunk & (g.xcd_chunk - 1u)) != 0u) g.xcd_chunk &= g.xcd_chunk - 1u;
    return g;
}

// A batch without ray trees through its stages: camera rays and the stage behind their hits, mirror bounces, the
// lit list, shadow rays and light sums -- fused or kernel by kernel, as the plan says.  Leaves the samples'
// colours in Q.color.
int launch_stages(const RtDeviceScene *ds, const FramePlan &P, hipStream_t stream, bool stats, Timing &tm, const RtSceneView &S,
                  const RtBuffers &Q, const RtBatch &B, const RtIdOut &io, uint32_t n_samples, const BatchGrids &G, uint32_t n_clear) {
    for (int bounce = 0; bounce <= P.max_bounce; ++bounce) {
        // bounce 0: one thread per sample.  Later bounces: the queue length is only known on
        // the device, so a fixed grid strides over it (and costs little when it is empty).
        const uint32_t g = bounce == 0 ? G.full : std::min<uint32_t>(G.full, RT_MAX_GRID);
        if (bounce > 0 && P.fuse_bounce) {  // one launch follows the rays of bounce 1 through all later bounces
            if (tm.begin(RT_K_TRACE_CLOSEST)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
            if (P.use_packets) {
                // With statistics: one resident set of waves striding over the packets, which hand their ray
                // counts to the batch's counters once each.  Without: a wave per packet, dealt by the hardware
                // as waves finish (mirror chains differ in length; a fixed deal waits for its unluckiest wave).
                const uint32_t gp = stats ? std::min<uint32_t>(G.packets, 256u * 4u * RT_BOUNCE_PACKET_RESIDENT) : G.packets;
                hipLaunchKernelGGL(k_bounces_packet, dim3(gp), dim3(RT_PACKET_BLOCK), P.bounce_lds, stream, S, Q, B, io, n_samples, stats ? 1u : 0u, TreeSpawn{});
            } else {
                const uint32_t gb = std::min<uint32_t>(G.full, RT_BOUNCE_GRID);
                hipLaunchKernelGGL(k_bounces_fast, dim3(gb), dim3(RT_BLOCK), P.lds_bytes, stream, S, Q, B, io, G.xcd_chunk);
            }
            tm.end();
            break;
        }
        if (tm.begin(RT_K_TRACE_CLOSEST)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
        if (P.ref_order)
            hipLaunchKernelGGL((k_trace_closest<false, true>), dim3(g), dim3(RT_BLOCK), P.lds_bytes, stream, S, Q, B, bounce, n_samples, G.xcd_chunk);
        else if (bounce == 0 && P.fuse_camera)
            hipLaunchKernelGGL(k_camera_shade_packet, dim3(G.packets), dim3(RT_PACKET_BLOCK), P.camera_lds, stream, S, Q, B, io, n_samples, G.xcd_chunk, n_clear);
        else if (bounce == 0 && P.use_packets)
            hipLaunchKernelGGL((k_trace_packet<0, 1>), dim3(G.packets), dim3(RT_PACKET_BLOCK), P.packet_lds, stream, S, Q, B, n_samples, G.xcd_chunk, n_clear);
        else
            hipLaunchKernelGGL((k_trace_fast<0>), dim3(G.full), dim3(RT_BLOCK), P.lds_bytes, stream, S, Q, B, bounce, n_samples, G.xcd_chunk);
        tm.end();
        if (bounce == 0 && P.fuse_camera) continue;  // the camera-ray kernel has done the stage behind it too
        if (tm.begin(RT_K_SHADE)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
        if (bounce == 0)
            hipLaunchKernelGGL((k_shade<true>), dim3(g), dim3(RT_BLOCK), 0, stream, S, Q, B, io, 0, n_samples);
        else
            hipLaunchKernelGGL((k_shade<false>), dim3(G.full), dim3(RT_BLOCK), 0, stream, S, Q, B, io, bounce, n_samples);
        tm.end();
    }
    if (!ds->has_lit) return RT_OK;
    if (P.direct_lit) {
        if (stats) hipLaunchKernelGGL(k_lit_total, dim3(1), dim3(RT_BLOCK), 0, stream, Q, G.lit);
    } else {
        if (tm.begin(RT_K_ACCUMULATE)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
        if (G.lit <= RT_LIT_OWN_OFFSET_MAX) {
            hipLaunchKernelGGL((k_lit_scatter<true>), dim3(G.lit), dim3(RT_LIT_THREADS), 0, stream, Q, n_samples);
        } else {
            hipLaunchKernelGGL(k_lit_scan, dim3(1), dim3(RT_LIT_THREADS), 0, stream, Q, G.lit);
            hipLaunchKernelGGL((k_lit_scatter<false>), dim3(G.lit), dim3(RT_LIT_THREADS), 0, stream, Q, n_samples);
        }
        tm.end();
    }
    // the number of lit hits is only known on the device: the grids are sized for the worst case, surplus blocks
    // return at once
    if (P.fused) {
        if (tm.begin(RT_K_TRACE_SHADOW)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
        hipLaunchKernelGGL(k_shadow_shade_packet, dim3(G.packets), dim3(RT_PACKET_BLOCK), P.packet_lds, stream, S, Q, B, io, G.xcd_chunk);
        tm.end();
        return RT_OK;
    }
    if (!S.disable_shadows && ds->n_point_lights > 0) {
        if (tm.begin(RT_K_TRACE_SHADOW)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
        if (P.ref_order)
            hipLaunchKernelGGL((k_trace_shadow<false, false, true>), dim3(G.full, ds->n_point_lights), dim3(RT_BLOCK), P.lds_bytes, stream, S, Q, G.xcd_chunk);
        else if (P.closest_blockers)
            hipLaunchKernelGGL((k_trace_packet<2, 1>), dim3(G.packets, ds->n_point_lights), dim3(RT_PACKET_BLOCK), P.packet_lds, stream, S, Q, B, n_samples, G.xcd_chunk, 0u);
        else if (P.use_packets)
            hipLaunchKernelGGL((k_trace_packet<1, 1>), dim3(G.packets, ds->n_point_lights), dim3(RT_PACKET_BLOCK), P.packet_lds, stream, S, Q, B, n_samples, G.xcd_chunk, 0u);
        else
            hipLaunchKernelGGL((k_trace_fast<1>), dim3(G.full, ds->n_point_lights), dim3(RT_BLOCK), P.lds_bytes, stream, S, Q, B, 0, n_samples, G.xcd_chunk);
        tm.end();
    }
    if (tm.begin(RT_K_ACCUMULATE)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
    hipLaunchKernelGGL(k_accumulate, dim3(G.full), dim3(RT_BLOCK), 0, stream, S, Q, B, io);
    tm.end();
    return RT_OK;
}

// Pixels of the call's tiles `from .. to - 1` that lie inside the rendered rectangle.
uint64_t pixels_inside(const RtBackend *be, const FramePlan &P, uint32_t from, uint32_t to) {
    uint64_t px = 0;
    for (uint32_t k = from; k < to; ++k) {
        const uint32_t tile = (uint32_t)be->tile_list_host[k];
        const uint32_t x0 = (tile % P.tiles_x) * P.tw, y0 = (tile / P.tiles_x) * P.th;
        px += (uint64_t)std::min(P.tw, P.W - x0) * std::min(P.th, P.H - y0);
    }
    return px;
}

// The ray counts of a frame from what its batches left: `host_counters` holds RT_CNT_SIZE words per batch that
// went through the stages or the depth-first tree kernel, `tree_secondary` / `tree_lit` the sums over the batches
// that went level by level.
void add_up_ray_counts(RtStats *stats, const RtBackend *be, const RtDeviceScene *ds, const FramePlan &P, const RtSceneView &S,
                       const std::vector<uint32_t> &host_counters, uint64_t tree_secondary, uint64_t tree_lit) {
    // camera rays = samples of the owned pixels that lie inside the image
    stats->primary_rays += pixels_inside(be, P, 0, (uint32_t)be->tile_list_host.size()) * P.spp;
    stats->secondary_rays += tree_secondary;
    stats->shaded_hits += tree_lit;
    if (!S.disable_shadows) stats->shadow_rays += tree_lit * ds->n_point_lights;
    for (size_t at = 0; at < host_counters.size(); at += RT_CNT_SIZE) {
        const uint32_t *c = host_counters.data() + at;
        if (ds->has_ggx) {
            auto wide = [&](int word) { return (uint64_t)c[word] | ((uint64_t)c[word + 1] << 32); };
            stats->secondary_rays += wide(RT_CNT_TREE_SECONDARY);
            stats->shadow_rays += wide(RT_CNT_TREE_SHADOW);
            stats->shaded_hits += wide(RT_CNT_TREE_SHADED);
        } else {
            for (int b = 1; b <= P.max_bounce && b < 32; ++b) stats->secondary_rays += c[RT_CNT_RAYS0 + b];
            stats->shaded_hits += c[RT_CNT_LIT];
            if (!S.disable_shadows) stats->shadow_rays += (uint64_t)c[RT_CNT_LIT] * ds->n_point_lights;
        }
    }
}

}  // namespace

static int render_frame_body(RtBackend *be, const RtDeviceScene *ds, const RtRenderArgs *args, float *fb_dev, const RtRayIds *ids,
                             hipStream_t stream, RtStats *stats, bool blocking, bool static_memory) {
    FramePlan P;
    if (int bad = plan_frame(be, ds, args, fb_dev, stats != nullptr, blocking, static_memory, P)) return bad;
    RT_HIP(hipSetDevice(be->device));
    if (!stream) stream = be->stream;
    struct InCall {  // rt_render_device_static: one call per device at a time (StaticSlot)
        std::atomic<int> *flag = nullptr;
        ~InCall() {
            if (flag) flag->store(0);
        }
    } in_call;
    if (static_memory) {
        if (!be->scratch) return rt::set_error(RT_ERR_DEVICE, "the static work memory of this device is not available");
        StaticSlot &slot = static_slot(be->device);
        if (slot.in_call.exchange(1) != 0)
            return rt::set_error(RT_ERR_INVALID, "two calls of rt_render_device_static for one GPU overlap: they share that GPU's work array");
        in_call.flag = &slot.in_call;
        // the frame rendered in the array before this one (any backend, any stream) ends first
        RT_HIP(hipStreamWaitEvent(stream, slot.done, 0));
    }

    // ---- the tiles of this call; a whole frame in static memory needs no list: slot k is tile k
    int rc = RT_OK;
    if (!static_memory && (rc = refresh_tile_list(be, args, P))) return rc;
    const uint32_t n_slots = static_memory ? P.tiles_x * P.tiles_y : (uint32_t)be->tile_list_host.size();

    // ---- tiles per batch, and work memory for a batch
    // (32 Mi samples, 8.5 GB of work buffers with two point lights: the 3840 x 2160 frame at 64 rays per pixel takes
    // 36.3 ms in batches of 16 Mi, 35.4 of 32 Mi, 34.9 of 64 Mi -- fewer, larger launches, fewer kernel tails)
    // (at most 2^30: samples are numbered in 32 bits, signed where an index doubles as a flag)
    const uint64_t max_batch = std::min<uint64_t>(args->batch_samples ? args->batch_samples : (32ull << 20), 1ull << 30);
    uint32_t slots_per_batch = (uint32_t)std::max<uint64_t>(1, max_batch / P.tile_samples);
    if (static_memory && (rc = static_batch_slots(ds, P, std::min<uint64_t>(slots_per_batch, std::max<uint32_t>(n_slots, 1)), slots_per_batch)))
        return rc;
    uint32_t batch_slots = std::min(slots_per_batch, std::max<uint32_t>(n_slots, 1));
    uint32_t capacity = (uint32_t)(batch_slots * P.tile_samples);
    if (P.tree_levels) {
        // A level's rays lie in the batch buffers like a batch's samples, so a batch of camera rays is sized to
        // leave room for levels that are larger than it: RT_TREE_ROOM times.  A level that still does not fit
        // halves the batch (below).
        const uint32_t room_slots = (uint32_t)std::min<uint64_t>(slots_per_batch, (uint64_t)std::max<uint32_t>(n_slots, 1) * RT_TREE_ROOM);
        batch_slots = std::min(std::max<uint32_t>(n_slots, 1), std::max<uint32_t>(1, room_slots / RT_TREE_ROOM));
        capacity = (uint32_t)(std::max(room_slots, batch_slots) * P.tile_samples);
    }
    RtBuffers Q;
    if (static_memory) {  // the arrays are laid out in the device's static array afresh on every call
        size_t used = 0;
        unsigned char *base = be->scratch;
        assign_buffers(Q, capacity, ds->n_point_lights, [&](size_t bytes) -> void * {
            void *p = base + used;
            used += (bytes + 255u) & ~(size_t)255u;
            return p;
        });
        Q.overflow = be->sticky;
    } else {
        if ((rc = ensure_buffers(be, capacity, ds->n_point_lights))) return rc;
        if (P.tree_levels && (rc = ensure_tree_buffers(be, be->buf_capacity, 2))) return rc;
        Q = be->buf;
    }
    Q.lit_f32 = P.ref_order ? 0u : 1u;
    const RtSceneView &S = P.ref_order ? ds->view : ds->view_fast;

    RtIdOut io{};
    if (ids) {
        io.primary_shape = ids->primary_shape; io.primary_tri = ids->primary_tri;
        io.final_shape = ids->final_shape; io.final_tri = ids->final_tri; io.bounces = ids->bounces;
        io.shadow_shape = ids->shadow_shape; io.shadow_tri = ids->shadow_tri;
        const size_t total = (size_t)P.W * P.H * P.spp * ds->n_point_lights;
        if (io.shadow_shape && total) hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(total)), dim3(RT_BLOCK), 0, stream, io.shadow_shape, total, -2);
        if (io.shadow_tri && total) hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(total)), dim3(RT_BLOCK), 0, stream, io.shadow_tri, total, -2);
    }

    Timing tm;
    tm.on = !static_memory && (stats != nullptr || args->timed_classes != 0);  // an event pair is an allocation
    if (!stats) tm.mask = args->timed_classes;
    tm.stream = stream;
    struct FrameEvents {  // destroyed on every return path
        hipEvent_t a = nullptr, b = nullptr;
        ~FrameEvents() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } frame_ev;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        RT_HIP(hipEventCreate(&frame_ev.a));
        RT_HIP(hipEventCreate(&frame_ev.b));
        RT_HIP(hipEventRecord(frame_ev.a, stream));
    }

    // Which of the backend's two counter sets is clean is noted (take_counter_set) BEFORE the kernels that do the
    // clearing are enqueued; any way out of this function other than the last line leaves that note untrue, so it is
    // taken back: both sets count as dirty and the next frame clears what it uses with a memset.
    struct DirtyGuard {
        RtBackend *be;
        bool armed;
        ~DirtyGuard() {
            if (armed) be->counter_dirty[0] = be->counter_dirty[1] = ~0u;
        }
    } dirty_guard{be, !static_memory};
    std::vector<uint32_t> host_counters;  // with statistics: RT_CNT_SIZE words per batch
    // ray trees level by level: tiles per batch (halved when a level outgrows the buffers), what the levels' own
    // counters came to, and whether the batch at hand has fallen back to the depth-first kernel
    uint32_t tree_slots = batch_slots;
    uint64_t tree_secondary = 0, tree_lit = 0;
    bool tree_depth_first = false;
    for (uint32_t slot0 = 0, step = 0; slot0 < n_slots; slot0 += step) {
        const uint32_t slots = std::min(P.tree_levels ? tree_slots : batch_slots, n_slots - slot0);
        step = slots;
        const RtBatch B = describe_batch(be, ds, args, P, static_memory, slot0, slots);
        const uint32_t n_samples = B.n_pixels * P.spp;
        const BatchGrids G = batch_grids(args, n_samples);
        uint32_t n_clear = 0;
        const bool clear_in_kernel = !ds->has_ggx && !P.ref_order && P.use_packets && !static_memory;
        if ((rc = take_counter_set(be, Q, G.lit, clear_in_kernel, static_memory, stream, n_clear))) return rc;
        const bool by_levels = P.tree_levels && !tree_depth_first;
        if (by_levels) {
            TreeBatch levels;
            if ((rc = tree_levels_batch(be, ds, args, P, stream, stats != nullptr, tm, S, Q, B, io, n_samples, levels))) return rc;
            be->counter_dirty[0] = be->counter_dirty[1] = ~0u;  // the levels took turns with both sets
            if (levels.too_large) {
                // A level has more rays than the buffers hold: the same tiles again in smaller batches (a ray's numbers
                // do not depend on its batch), and a single tile that does not fit through the depth-first kernel.
                if (slots > 1) tree_slots = std::max(1u, slots / 2u);
                else tree_depth_first = true;
                step = 0;
                continue;
            }
            tree_secondary += levels.secondary;
            tree_lit += levels.lit;
        } else if (ds->has_ggx) {  // one thread per ray tree, depth first, instead of the stages
            if (tm.begin(RT_K_TRACE_CLOSEST)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
            if (P.ref_order)
                hipLaunchKernelGGL((k_render_tree<false>), dim3(G.full), dim3(RT_BLOCK), P.lds_bytes, stream, S, Q, B, io, n_samples);
            else
                hipLaunchKernelGGL((k_render_tree<true>), dim3(G.full), dim3(RT_BLOCK), P.lds_bytes, stream, S, Q, B, io, n_samples);
            tm.end();
        } else if ((rc = launch_stages(ds, P, stream, stats != nullptr, tm, S, Q, B, io, n_samples, G, n_clear))) {
            return rc;
        }
        // ---- the samples' colours into the pixels
        if (tm.begin(RT_K_RESOLVE)) return rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed");
        const dim3 resolve_grid((uint32_t)std::min<uint64_t>(((uint64_t)B.n_pixels + 63u) / 64u, 4u * RT_MAX_GRID));
        if (rt_resolve_by_pixels(P.spp)) hipLaunchKernelGGL(k_resolve_pixels, resolve_grid, dim3(RT_RESOLVE_BLOCK), 0, stream, Q, B, fb_dev);
        else hipLaunchKernelGGL(k_resolve, resolve_grid, dim3(RT_RESOLVE_BLOCK), 0, stream, Q, B, fb_dev);
        tm.end();

        if (P.wait_batches) {
            if (stats && !by_levels) {  // counters must be read before the next batch clears them
                const size_t at = host_counters.size();
                host_counters.resize(at + RT_CNT_SIZE);
                RT_HIP(hipMemcpyAsync(host_counters.data() + at, Q.counters, RT_CNT_SIZE * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            }
            RT_HIP(hipStreamSynchronize(stream));
            if (P.serve_progress)  // pixels of the owned tiles that lie inside the image, finished / in all
                args->progress(pixels_inside(be, P, 0, slot0 + slots), pixels_inside(be, P, 0, n_slots), args->progress_user);
        }
        tree_depth_first = false;  // (it was this batch's alone)
    }
    if (hipError_t e = hipGetLastError(); e != hipSuccess)  // a launch failed: which counters were zeroed is not known
        return rt::set_error(RT_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(e));
    if (!stats && !tm.pairs.empty()) {  // read later by rt_collect_timings
        be->deferred.insert(be->deferred.end(), tm.pairs.begin(), tm.pairs.end());
        be->deferred_stream = stream;
        tm.pairs.clear();
    }
    if (stats) {
        RT_HIP(hipEventRecord(frame_ev.b, stream));
        RT_HIP(hipStreamSynchronize(stream));
        float ms = 0.0f;
        RT_HIP(hipEventElapsedTime(&ms, frame_ev.a, frame_ev.b));
        stats->render_ms = ms;
        tm.collect(stats);
        add_up_ray_counts(stats, be, ds, P, S, host_counters, tree_secondary, tree_lit);
    }
    // A frame on a stream of the caller's: rt_backend_sync must be able to wait for it without keeping the
    // stream's handle (the caller may destroy the stream), so an event of the backend's marks its end.
    if (stream != be->stream) {
        RT_HIP(hipEventRecord(be->foreign_done, stream));
        be->foreign_pending = true;
    }
    if (static_memory) RT_HIP(hipEventRecord(static_slot(be->device).done, stream));  // see StaticSlot
    rc = stats ? rt::finish_stream(be, stream) : RT_OK;
    dirty_guard.armed = rc != RT_OK;
    return rc;
}

// (what every render entry point, and the worker threads of rt_render_multi, go through)
int rt::render_frame(RtBackend *be, const RtDeviceScene *ds, const RtRenderArgs *args, float *fb_dev, const RtRayIds *ids,
                     hipStream_t stream, RtStats *stats, bool blocking, bool static_memory) {
    return rt::guarded([&]() -> int { return render_frame_body(be, ds, args, fb_dev, ids, stream, stats, blocking, static_memory); });
}

// ------------------------------------------------------------------ C ABI

extern "C" const char *rt_last_error(void) { return rt::last_error_slot().c_str(); }

#ifdef RT_WAVE_TIMELINE
// development builds only: where the packet kernels log their waves' start and end times (nullptr: nowhere)
extern "C" int rt_debug_wave_log(void *log_dev) {
    RT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wave_log), &log_dev, sizeof(log_dev)));
    return RT_OK;
}
#endif

extern "C" int rt_backend_create(int device_ordinal, RtBackend **backend_out) {
    return rt::guarded([&]() -> int {
        if (!backend_out) return rt::set_error(RT_ERR_INVALID, "null argument");
        *backend_out = nullptr;
        int count = 0;
        RT_HIP(hipGetDeviceCount(&count));
        if (device_ordinal < 0 || device_ordinal >= count)
            return rt::set_error(RT_ERR_DEVICE, "no HIP device with ordinal " + std::to_string(device_ordinal));
        RT_HIP(hipSetDevice(device_ordinal));
        RtBackend *be = new RtBackend();
        be->device = device_ordinal;
        hipError_t e = hipStreamCreateWithFlags(&be->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&be->sticky, 256);
        if (e == hipSuccess) e = hipMemset(be->sticky, 0, 256);
        if (e == hipSuccess) e = hipHostMalloc((void **)&be->sticky_host, 64, hipHostMallocDefault);
        // the device address of this device's copy of the static work array (this device is current)
        if (e == hipSuccess) e = hipGetSymbolAddress((void **)&be->scratch, HIP_SYMBOL(scratch));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&be->foreign_done, hipEventDisableTiming);
        if (e == hipSuccess) {
            if (device_ordinal < RT_STATIC_MAX_DEVICES) {
                static std::mutex mu;
                std::lock_guard<std::mutex> lock(mu);
                StaticSlot &slot = static_slot(device_ordinal);
                if (!slot.done) e = hipEventCreateWithFlags(&slot.done, hipEventDisableTiming);
            } else {
                be->scratch = nullptr;  // rt_render_device_static then reports that it is not available
            }
        }
        if (e != hipSuccess) {
            rt_backend_destroy(be);
            return rt::set_error(RT_ERR_DEVICE, std::string("backend set-up: ") + hipGetErrorString(e));
        }
        *be->sticky_host = 0u;
        *backend_out = be;
        return RT_OK;
    });
}

extern "C" void rt_backend_destroy(RtBackend *be) {
    if (!be) return;
    (void)hipSetDevice(be->device);
    if (be->stream) (void)hipStreamSynchronize(be->stream);
    if (be->foreign_pending && be->foreign_done) (void)hipEventSynchronize(be->foreign_done);  // frames on the caller's streams
    if (be->foreign_done) (void)hipEventDestroy(be->foreign_done);
    for (auto &p : be->deferred) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    free_all(be->buf_allocs);
    free_all(be->tree.allocs);
    if (be->tree.host_words) (void)hipHostFree(be->tree.host_words);
    if (be->tree.copied) (void)hipEventDestroy(be->tree.copied);
    if (be->tile_list) (void)hipFree(be->tile_list);
    if (be->tile_where) (void)hipFree(be->tile_where);
    if (be->fb) (void)hipFree(be->fb);
    if (be->pinned) (void)hipHostFree(be->pinned);
    if (be->sticky) (void)hipFree(be->sticky);
    if (be->sticky_host) (void)hipHostFree(be->sticky_host);
    if (be->stream) (void)hipStreamDestroy(be->stream);
    delete be;
}

extern "C" int rt_backend_sync(RtBackend *be) {
    return rt::guarded([&]() -> int {
        if (!be) return rt::set_error(RT_ERR_INVALID, "null argument");
        RT_HIP(hipSetDevice(be->device));
        if (be->foreign_pending) {  // frames that went to a stream of the caller's: their end is marked by an event
            RT_HIP(hipStreamWaitEvent(be->stream, be->foreign_done, 0));
            be->foreign_pending = false;
        }
        return rt::finish_stream(be, be->stream);
    });
}

extern "C" int rt_host_alloc(size_t bytes, void **ptr_out) {
    return rt::guarded([&]() -> int {
        if (!ptr_out) return rt::set_error(RT_ERR_INVALID, "null argument");
        *ptr_out = nullptr;
        RT_HIP(hipHostMalloc(ptr_out, std::max<size_t>(bytes, 1), hipHostMallocDefault));
        return RT_OK;
    });
}

extern "C" void rt_host_free(void *ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

extern "C" int rt_scene_upload(RtBackend *be, const RtScene *scene, RtDeviceScene **out) {
    return rt_scene_upload_with(be, scene, nullptr, out);
}

extern "C" int rt_scene_upload_with(RtBackend *be, const RtScene *scene, const RtUploadOptions *options, RtDeviceScene **out) {
    return rt::guarded([&]() -> int {
        if (!be || !scene || !out) return rt::set_error(RT_ERR_INVALID, "null argument");
        *out = nullptr;
        const uint32_t upload_flags = options ? options->flags : 0u;
        if (upload_flags & ~(uint32_t)RT_UPLOAD_REFERENCE_MESH_TREES) return rt::set_error(RT_ERR_INVALID, "unknown upload flag");
        bool uses_ggx = false;
        for (uint32_t s = 0; s < scene->n_shapes; ++s) {
            const int32_t sh = scene->shapes[s].shader;
            uses_ggx = uses_ggx || (sh >= 0 && (uint32_t)sh < scene->n_shaders && scene->shaders[sh].type == RT_SHADER_GGX_MIRROR);
        }
        if (uses_ggx && scene->recursion_depth > RT_TREE_MAX_DEPTH)
            return rt::set_error(RT_ERR_UNSUPPORTED, "GGXMirror with a recursion depth above " + std::to_string(RT_TREE_MAX_DEPTH));
        rt::FlatScene flat;
        const std::string err = rt::flatten_scene(*scene, flat, (upload_flags & RT_UPLOAD_REFERENCE_MESH_TREES) != 0u);
        if (!err.empty()) return rt::set_error(RT_ERR_INVALID, err);
        RT_HIP(hipSetDevice(be->device));

        RtDeviceScene *ds = new RtDeviceScene();
        ds->backend = be;
        ds->device = be->device;
        auto fail = [&](int rc) {
            free_all(ds->allocs);
            delete ds;
            return rc;
        };
        auto up = [&](const void *src, size_t bytes, const void **dst) -> int {
            void *p = nullptr;
            RT_HIP(hipMalloc(&p, std::max<size_t>(bytes, 16)));
            ds->allocs.push_back(p);
            if (bytes) RT_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
            *dst = p;
            return RT_OK;
        };
        RtSceneView &V = ds->view;
        std::vector<int32_t> point_lights;
        for (uint32_t i = 0; i < scene->n_lights; ++i)
            if (scene->lights[i].type == RT_LIGHT_POINT) point_lights.push_back((int32_t)i);
        int rc;
        if ((rc = up(flat.nodes.data(), flat.nodes.size() * sizeof(RtNode), (const void **)&V.nodes))) return fail(rc);
        if ((rc = up(flat.tprims.data(), flat.tprims.size() * sizeof(RtTPrim), (const void **)&V.tprims))) return fail(rc);
        if ((rc = up(flat.tris.data(), flat.tris.size() * sizeof(RtTri), (const void **)&V.tris))) return fail(rc);
        if ((rc = up(flat.tri_xf.data(), flat.tri_xf.size() * sizeof(RtTriXf), (const void **)&V.tri_xf))) return fail(rc);
        if ((rc = up(flat.tprim_xf.data(), flat.tprim_xf.size() * sizeof(float), (const void **)&V.tprim_xf))) return fail(rc);
        if ((rc = up(scene->shapes, scene->n_shapes * sizeof(RtShape), (const void **)&V.shapes))) return fail(rc);
        if ((rc = up(scene->protos, scene->n_protos * sizeof(RtShape), (const void **)&V.protos))) return fail(rc);
        if ((rc = up(flat.verts64.data(), flat.verts64.size() * sizeof(double), (const void **)&V.verts64))) return fail(rc);
        if ((rc = up(flat.mesh_tri_base.data(), flat.mesh_tri_base.size() * sizeof(uint64_t), (const void **)&V.mesh_tri_base))) return fail(rc);
        if ((rc = up(scene->shaders, scene->n_shaders * sizeof(RtShader), (const void **)&V.shaders))) return fail(rc);
        if ((rc = up(scene->lights, scene->n_lights * sizeof(RtLight), (const void **)&V.lights))) return fail(rc);
        if ((rc = up(point_lights.data(), point_lights.size() * sizeof(int32_t), (const void **)&V.point_lights))) return fail(rc);
        V.camera = scene->camera;
        std::memcpy(V.background, scene->background, sizeof(V.background));
        V.n_lights = (int32_t)scene->n_lights;
        V.n_point_lights = (int32_t)point_lights.size();
        V.tlas_root = flat.tlas_root;
        V.recursion_depth = scene->recursion_depth;
        V.disable_shadows = scene->disable_shadows;
        V.stack_depth = flat.stack_depth;
        ds->view_fast = V;
        ds->view_fast.tlas_root = flat.tlas_root_fast;
        ds->view_fast.stack_depth = flat.stack_depth_fast;
        ds->width = scene->image_width;
        ds->height = scene->image_height;
        ds->n_point_lights = (uint32_t)point_lights.size();
        for (uint32_t s = 0; s < scene->n_shapes; ++s) {
            const int32_t t = scene->shaders[scene->shapes[s].shader].type;
            ds->has_mirror = ds->has_mirror || t == RT_SHADER_PERFECT_MIRROR;
            ds->has_lit = ds->has_lit || t == RT_SHADER_LAMBERTIAN || t == RT_SHADER_BLINN_PHONG;
        }
        ds->has_ggx = uses_ggx;
        *out = ds;
        return RT_OK;
    });
}

extern "C" void rt_scene_free(RtDeviceScene *ds) {
    if (!ds) return;
    (void)hipSetDevice(ds->device);
    (void)hipDeviceSynchronize();
    free_all(ds->allocs);
    delete ds;
}

extern "C" int rt_collect_timings(RtBackend *be, RtStats *out) {
    return rt::guarded([&]() -> int {
        if (!be || !out) return rt::set_error(RT_ERR_INVALID, "null argument");
        std::memset(out, 0, sizeof(*out));
        if (be->deferred.empty()) return RT_OK;
        RT_HIP(hipSetDevice(be->device));
        RT_HIP(hipStreamSynchronize(be->deferred_stream));
        for (auto &p : be->deferred) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                out->kernel_ms[p.klass] += ms;
                out->kernel_launches[p.klass] += 1;
            }
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
        be->deferred.clear();
        return RT_OK;
    });
}

extern "C" uint32_t rt_scene_point_light_count(const RtDeviceScene *ds) { return ds ? ds->n_point_lights : 0; }

extern "C" int rt_render_device(RtBackend *be, const RtDeviceScene *ds, const RtRenderArgs *args, float *fb_rgb_dev,
                                const RtRayIds *ids_dev, void *hip_stream, RtStats *stats_out) {
    return rt::render_frame(be, ds, args, fb_rgb_dev, ids_dev, (hipStream_t)hip_stream, stats_out, false, false);
}

extern "C" int rt_render_device_static(RtBackend *be, const RtDeviceScene *ds, const RtRenderArgs *args, float *fb_rgb_dev,
                                       const RtRayIds *ids_dev, void *hip_stream) {
    return rt::render_frame(be, ds, args, fb_rgb_dev, ids_dev, (hipStream_t)hip_stream, nullptr, false, true);
}

extern "C" int rt_render_async(RtBackend *be, const RtDeviceScene *ds, const RtRenderArgs *args, float *fb_rgb_pinned) {
    return rt::guarded([&]() -> int {
        if (!be || !ds || !args || !fb_rgb_pinned) return rt::set_error(RT_ERR_INVALID, "null argument");
        if (args->part_count > 1) return rt::set_error(RT_ERR_INVALID, "rt_render_async renders whole frames (part_count <= 1)");
        RT_HIP(hipSetDevice(be->device));
        if (!rt::is_pinned_host(fb_rgb_pinned))
            return rt::set_error(RT_ERR_INVALID, "rt_render_async needs page-locked host memory (rt_host_alloc)");
        Window win;
        if (int bad = resolve_window(args, ds->width, ds->height, win)) return bad;
        const size_t fb_floats = (size_t)win.w * win.h * 3;
        if (int grown = rt::ensure_device_fb(be, fb_floats)) return grown;
        if (int rc = rt::render_frame(be, ds, args, be->fb, nullptr, be->stream, nullptr, false, false)) return rc;
        RT_HIP(hipMemcpyAsync(fb_rgb_pinned, be->fb, fb_floats * sizeof(float), hipMemcpyDeviceToHost, be->stream));
        return RT_OK;
    });
}

extern "C" int rt_render(RtBackend *be, const RtDeviceScene *ds, const RtRenderArgs *args, float *fb_rgb,
                         const RtRayIds *ids_host, RtStats *stats_out) {
    return rt::guarded([&]() -> int {
        if (!be || !ds || !args || !fb_rgb) return rt::set_error(RT_ERR_INVALID, "null argument");
        RT_HIP(hipSetDevice(be->device));
        Window win;
        if (int bad = resolve_window(args, ds->width, ds->height, win)) return bad;
        const size_t W = win.w, H = win.h;  // the rendered pixels: what the outputs cover
        const size_t n = args->sqrt_rays_per_pixel;
        const size_t samples = W * H * n * n;
        const size_t fb_floats = W * H * 3;
        const uint32_t parts = std::max<uint32_t>(args->part_count, 1);
        const uint32_t tw = args->tile_w ? args->tile_w : 32, th = args->tile_h ? args->tile_h : 32;
        uint32_t tiles_x, tiles_y;
        if (int bad = tile_grid((uint32_t)W, (uint32_t)H, tw, th, tiles_x, tiles_y)) return bad;
        // a partitioned render lands tile-major (edge tiles padded), so size for whole tiles
        size_t dev_floats = std::max(fb_floats, (size_t)tiles_x * tiles_y * tw * th * 3);
        if (parts > 1 && args->part_span > 1) {  // the span's blocks, a pitch apart
            const std::vector<uint32_t> counts = part_tile_counts(tiles_x, tiles_y, parts, args->part0_weight <= 8 ? args->part0_weight : 0);
            const size_t pitch = args->part_span_pitch_floats ? (size_t)args->part_span_pitch_floats
                                                              : (size_t)*std::max_element(counts.begin(), counts.end()) * tw * th * 3;
            dev_floats = std::max(dev_floats, pitch * args->part_span);
        }
        if (int grown = rt::ensure_device_fb(be, dev_floats)) return grown;

        // device mirrors of the requested id arrays
        std::vector<void *> tmp;
        RtRayIds dev{};
        struct Copy {
            int32_t *host, *dev;
            size_t count;
        };
        std::vector<Copy> copies;
        auto cleanup = [&]() { free_all(tmp); };
        if (ids_host) {
            const size_t npl = ds->n_point_lights;
            struct Want {
                int32_t *host;
                int32_t **dev;
                size_t count;
            } wants[] = {{ids_host->primary_shape, &dev.primary_shape, samples}, {ids_host->primary_tri, &dev.primary_tri, samples},
                         {ids_host->final_shape, &dev.final_shape, samples},     {ids_host->final_tri, &dev.final_tri, samples},
                         {ids_host->bounces, &dev.bounces, samples},             {ids_host->shadow_shape, &dev.shadow_shape, samples * npl},
                         {ids_host->shadow_tri, &dev.shadow_tri, samples * npl}};
            for (auto &w : wants) {
                if (!w.host || w.count == 0) continue;
                void *p = nullptr;
                hipError_t e = hipMalloc(&p, w.count * sizeof(int32_t));
                if (e != hipSuccess) {
                    cleanup();
                    return rt::set_error(RT_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
                }
                tmp.push_back(p);
                *w.dev = (int32_t *)p;
                (void)hipMemsetAsync(p, 0xFF, w.count * sizeof(int32_t), be->stream);  // -1
                copies.push_back({w.host, (int32_t *)p, w.count});
            }
        }
        // The kernels are only enqueued here (statistics and a progress callback excepted, which make
        // render_frame wait); the copy of the frame follows on the same stream, and ONE wait ends the call.
        int rc = rt::render_frame(be, ds, args, be->fb, ids_host ? &dev : nullptr, be->stream, stats_out, true, false);
        if (rc) {
            cleanup();
            return rc;
        }
        auto fail = [&](hipError_t e, const char *what) {
            cleanup();
            return rt::set_error(RT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
        };
        // what to bring over: the frame, or the owned tiles of a part (tile-major, scattered below)
        // (with part_span the parts' blocks lie a pitch apart: up to the last place that holds a tile)
        size_t places = be->tile_list_host.size();
        for (int32_t at : be->tile_place_host) places = std::max(places, (size_t)at + 1);
        const size_t copy_floats = parts > 1 ? places * tw * th * 3 : fb_floats;
        const bool direct = parts <= 1 && rt::is_pinned_host(fb_rgb);  // the device can write the caller's memory itself
        float *landing = fb_rgb;
        if (!direct) {
            if (int grown = rt::ensure_pinned(be, std::max<size_t>(copy_floats, 1))) {
                cleanup();
                return grown;
            }
            landing = be->pinned;
        }
        if (copy_floats) {
            hipError_t e = hipMemcpyAsync(landing, be->fb, copy_floats * sizeof(float), hipMemcpyDeviceToHost, be->stream);
            if (e != hipSuccess) return fail(e, "hipMemcpyAsync");
        }
        for (auto &c : copies) {  // pageable destinations: the runtime stages these itself
            hipError_t e = hipMemcpyAsync(c.host, c.dev, c.count * sizeof(int32_t), hipMemcpyDeviceToHost, be->stream);
            if (e != hipSuccess) return fail(e, "hipMemcpyAsync");
        }
        if ((rc = rt::finish_stream(be, be->stream))) {
            cleanup();
            return rc;
        }
        if (parts > 1) {
            // compact tile-major -> full layout, owned pixels only
            for (size_t s = 0; s < be->tile_list_host.size(); ++s) {
                const uint32_t tile = (uint32_t)be->tile_list_host[s];
                const uint32_t x0 = (tile % tiles_x) * tw, y0 = (tile / tiles_x) * th;
                const uint32_t w_in = std::min<uint32_t>(tw, (uint32_t)W - x0), h_in = std::min<uint32_t>(th, (uint32_t)H - y0);
                const size_t at = be->tile_place_host.empty() ? s : (size_t)be->tile_place_host[s];
                for (uint32_t y = 0; y < h_in; ++y)
                    std::memcpy(fb_rgb + ((size_t)x0 + (size_t)(y0 + y) * W) * 3, landing + (at * tw * th + (size_t)y * tw) * 3,
                                (size_t)w_in * 3 * sizeof(float));
            }
        } else if (!direct) {
            std::memcpy(fb_rgb, landing, fb_floats * sizeof(float));
        }
        cleanup();
        return RT_OK;
    });
}

extern "C" int rt_partition_info(uint32_t width, uint32_t height, const RtRenderArgs *args, uint32_t *tiles_total,
                                 uint32_t *tiles_owned, uint32_t *tiles_owned_max) {
    return rt::guarded([&]() -> int {
        if (!args) return rt::set_error(RT_ERR_INVALID, "null argument");
        const uint32_t tw = args->tile_w ? args->tile_w : 32, th = args->tile_h ? args->tile_h : 32;
        const uint32_t parts = std::max<uint32_t>(args->part_count, 1);
        if (int bad = check_partition(args)) return bad;
        uint32_t tiles_x, tiles_y;
        if (int bad = tile_grid(width, height, tw, th, tiles_x, tiles_y)) return bad;
        const std::vector<uint32_t> per_part = part_tile_counts(tiles_x, tiles_y, parts, args->part0_weight);
        if (tiles_total) *tiles_total = tiles_x * tiles_y;
        if (tiles_owned) *tiles_owned = per_part[args->part_index];
        if (tiles_owned_max) *tiles_owned_max = *std::max_element(per_part.begin(), per_part.end());
        return RT_OK;
    });
}

extern "C" int rt_partition_tiles(uint32_t width, uint32_t height, const RtRenderArgs *args, int32_t *tiles_out,
                                  uint32_t capacity, uint32_t *count_out) {
    return rt::guarded([&]() -> int {
        if (!args || !count_out) return rt::set_error(RT_ERR_INVALID, "null argument");
        const uint32_t tw = args->tile_w ? args->tile_w : 32, th = args->tile_h ? args->tile_h : 32;
        const uint32_t parts = std::max<uint32_t>(args->part_count, 1);
        if (int bad = check_partition(args)) return bad;
        std::vector<int32_t> tiles;
        uint32_t tiles_x, tiles_y;
        if (int bad = tile_grid(width, height, tw, th, tiles_x, tiles_y)) return bad;
        owned_tiles(tiles_x, tiles_y, args->part_index, parts, args->part0_weight, tiles);
        *count_out = (uint32_t)tiles.size();
        if (tiles_out) {
            if (capacity < tiles.size()) return rt::set_error(RT_ERR_INVALID, "tile buffer too small");
            std::memcpy(tiles_out, tiles.data(), tiles.size() * sizeof(int32_t));
        }
        return RT_OK;
    });
}

extern "C" int rt_untile_device(RtBackend *be, uint32_t width, uint32_t height, const RtRenderArgs *args,
                                const float *gathered_dev, float *fb_rgb_dev, void *hip_stream) {
    return rt::guarded([&]() -> int {
        return rt_untile_device_pitched(be, width, height, args, gathered_dev, 0, fb_rgb_dev, hip_stream);
    });
}

extern "C" int rt_untile_device_pitched(RtBackend *be, uint32_t width, uint32_t height, const RtRenderArgs *args,
                                        const float *gathered_dev, size_t part_pitch_floats, float *fb_rgb_dev,
                                        void *hip_stream) {
    return rt::guarded([&]() -> int {
        if (!be || !args || !gathered_dev || !fb_rgb_dev) return rt::set_error(RT_ERR_INVALID, "null argument");
        const uint32_t tw = args->tile_w ? args->tile_w : 32, th = args->tile_h ? args->tile_h : 32;
        const uint32_t parts = std::max<uint32_t>(args->part_count, 1);
        uint32_t tiles_x, tiles_y;
        if (int bad = tile_grid(width, height, tw, th, tiles_x, tiles_y)) return bad;
        uint32_t stride = 0;
        int rc = rt_partition_info(width, height, args, nullptr, nullptr, &stride);
        if (rc) return rc;
        const size_t block_floats = (size_t)stride * tw * th * 3;
        if (part_pitch_floats == 0) part_pitch_floats = block_floats;  // end to end
        if (part_pitch_floats < block_floats) return rt::set_error(RT_ERR_INVALID, "part pitch smaller than a part's block of tiles");
        if (parts > 4096 || stride >= (1u << 20)) return rt::set_error(RT_ERR_INVALID, "too many parts or tiles per part for the un-tile table");
        if (args->part_rotation >= parts) return rt::set_error(RT_ERR_INVALID, "part_rotation must be below part_count");
        RT_HIP(hipSetDevice(be->device));
        hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : be->stream;
        const uint32_t key[5] = {width, height, tw, th, parts | (args->part0_weight << 24)};
        if (std::memcmp(key, be->untile_key, sizeof(key)) != 0 || !be->tile_where) {
            std::vector<uint32_t> where((size_t)tiles_x * tiles_y), next(parts, 0);
            const RtTileMap map = rt_tile_map(tiles_x, tiles_y, parts, args->part0_weight);
            for (uint32_t ty = 0; ty < tiles_y; ++ty)
                for (uint32_t tx = 0; tx < tiles_x; ++tx) {
                    const uint32_t k = ty * tiles_x + tx, p = rt_tile_map_owner(map, tx, ty);
                    where[k] = (p << 20) | next[p]++;
                }
            if (be->tile_where) (void)hipFree(be->tile_where);
            be->tile_where = nullptr;
            RT_HIP(hipMalloc((void **)&be->tile_where, where.size() * sizeof(uint32_t)));
            RT_HIP(hipMemcpy(be->tile_where, where.data(), where.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            std::memcpy(be->untile_key, key, sizeof(key));
        }
        const uint64_t untile_blocks = ((uint64_t)width * height + RT_UNTILE_BLOCK - 1) / RT_UNTILE_BLOCK;
        hipLaunchKernelGGL(k_untile, dim3((uint32_t)std::min<uint64_t>(untile_blocks, 4u * RT_MAX_GRID)), dim3(RT_UNTILE_BLOCK), 0, stream,
                           gathered_dev, part_pitch_floats, fb_rgb_dev, be->tile_where, width, height, tw, th, tiles_x, parts,
                           (parts - args->part_rotation) % parts);
        RT_HIP(hipGetLastError());
        return RT_OK;
    });
}

extern "C" int rt_trace_rays(RtBackend *be, const RtDeviceScene *ds, uint64_t n, const double *origins,
                             const double *directions, const double *tmin, const double *tmax, int32_t mode,
                             float *t_out, int32_t *shape_out, int32_t *tri_out) {
    return rt::guarded([&]() -> int {
        if (!be || !ds || !origins || !directions || !tmin || !tmax || !t_out || !shape_out || !tri_out)
            return rt::set_error(RT_ERR_INVALID, "null argument");
        if (n == 0) return RT_OK;
        RT_HIP(hipSetDevice(be->device));
        std::vector<void *> tmp;
        double *d_o = nullptr, *d_d = nullptr, *d_tmin = nullptr, *d_tmax = nullptr;
        float *d_t = nullptr;
        int32_t *d_shape = nullptr, *d_tri = nullptr;
        uint32_t *d_over = nullptr;
        int rc;
        auto bail = [&](int code) {
            free_all(tmp);
            return code;
        };
        if ((rc = alloc_into(tmp, &d_o, n * 3)) || (rc = alloc_into(tmp, &d_d, n * 3)) || (rc = alloc_into(tmp, &d_tmin, n)) ||
            (rc = alloc_into(tmp, &d_tmax, n)) || (rc = alloc_into(tmp, &d_t, n)) || (rc = alloc_into(tmp, &d_shape, n)) ||
            (rc = alloc_into(tmp, &d_tri, n)) || (rc = alloc_into(tmp, &d_over, 1)))
            return bail(rc);
        if (hipMemcpy(d_o, origins, n * 3 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_d, directions, n * 3 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_tmin, tmin, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_tmax, tmax, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(d_over, 0, sizeof(uint32_t)) != hipSuccess)
            return bail(rt::set_error(RT_ERR_DEVICE, "hipMemcpy to device failed"));
        const int mode_stack_depth = mode == RT_MODE_REFERENCE_ORDER ? ds->view.stack_depth : ds->view_fast.stack_depth;
        const size_t lds_bytes = (size_t)(std::max(mode_stack_depth, 1) + RT_LDS_AUX_ROWS) * RT_BLOCK * sizeof(int32_t);
        if (lds_bytes > 64 * 1024) return bail(rt::set_error(RT_ERR_UNSUPPORTED, "scene trees too deep for the LDS traversal stack"));
        if (mode == RT_MODE_REFERENCE_ORDER)
            hipLaunchKernelGGL((k_trace_user<false, true>), dim3(grid_for(n)), dim3(RT_BLOCK), lds_bytes, be->stream, ds->view, n, d_o, d_d, d_tmin, d_tmax, d_t, d_shape, d_tri, d_over);
        else
            hipLaunchKernelGGL((k_trace_user<true, false>), dim3(grid_for(n)), dim3(RT_BLOCK), lds_bytes, be->stream, ds->view_fast, n, d_o, d_d, d_tmin, d_tmax, d_t, d_shape, d_tri, d_over);
        uint32_t over = 0;
        if (hipStreamSynchronize(be->stream) != hipSuccess || hipGetLastError() != hipSuccess ||
            hipMemcpy(t_out, d_t, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(shape_out, d_shape, n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(tri_out, d_tri, n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&over, d_over, sizeof(over), hipMemcpyDeviceToHost) != hipSuccess)
            return bail(rt::set_error(RT_ERR_DEVICE, "trace kernel or copy back failed"));
        free_all(tmp);
        if (over) return rt::set_error(RT_ERR_DEVICE, "traversal stack overflow (internal error)");
        return RT_OK;
    });
}
