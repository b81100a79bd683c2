// rt_backend_internal.h -- what the two host-side translation units of librt_hip.so share:
// the backend and device-scene objects behind the opaque handles of include/rt_hip.h, and the
// few internal entry points the multi-device code (rt_multi.cpp) builds on.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_error.h"
#include "rt_pipeline.h"

#define RT_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return rt::set_error(RT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct RtEventPair {
    hipEvent_t a, b;
    int klass;
};

struct RtBackend {
    int device = 0;
    hipStream_t stream = nullptr;
    // Recorded behind every frame that is enqueued on a stream of the caller's: rt_backend_sync and
    // rt_backend_destroy wait for this event, never for the caller's stream, whose handle may be gone by then.
    hipEvent_t foreign_done = nullptr;
    bool foreign_pending = false;
    // event pairs of frames enqueued with RtRenderArgs::timed_classes, until rt_collect_timings
    std::vector<RtEventPair> deferred;
    hipStream_t deferred_stream = nullptr;
    // batch buffers, grown on demand
    RtBuffers buf{};
    std::vector<void *> buf_allocs;
    uint32_t buf_capacity = 0;
    uint32_t buf_point_lights = 0;
    // leading words of each of buf's two counter sets that may be non-zero (0: the set is clean, ~0u: all of it)
    uint32_t counter_dirty[2] = {~0u, ~0u};
    // "a traversal stack overflowed" (RtBuffers::overflow): a device word that kernels only ever set,
    // and the page-locked host word it is copied to when the host looks
    uint32_t *sticky = nullptr;
    uint32_t *sticky_host = nullptr;
    // device address of this device's static work array (rt_render_device_static), from hipGetSymbolAddress
    unsigned char *scratch = nullptr;
    // Ray trees traced level by level (scenes with a GGXMirror shader): grown on demand, see ensure_tree_buffers
    struct TreeLevels {
        std::vector<void *> allocs;
        uint32_t capacity = 0;
        // the GGX hits that the rays of a level leave (RtBuffers::g_*), two sets: a level's rays are made from the
        // hits of the level above while they leave their own
        double *g_p[2][3] = {}, *g_n[2][3] = {}, *g_in[2][3] = {};
        int32_t *g_shape[2] = {}, *g_tri[2] = {}, *g_mark[2] = {};
        uint32_t *off = nullptr;         // [capacity] where a node's children start inside its chunk's run of children
        uint32_t *chunk_base = nullptr;  // [chunks + 1] children per chunk of nodes, then their exclusive sums; last: all
        uint32_t *parent = nullptr;      // [capacity] the node that each node of the NEXT level hangs under
        uint64_t *root[2] = {}, *path[2] = {};  // [capacity] per node: its tree's frame-wide sample index, its ray's path
        // per level (level 0 is the batch's samples, whose colours are the batch's own colour array)
        std::vector<float *> color;      // [3 * capacity]
        std::vector<uint32_t *> base, count;  // [capacity] children of each node: where they start in the next level, how many
        // how many rays the next level has, read by the host between levels: page-locked, so that the copy is only
        // enqueued, and an event behind it, so that the host need not wait for what is enqueued behind the copy
        uint32_t *host_words = nullptr;
        hipEvent_t copied = nullptr;
    } tree;
    // tile list of the last partition
    int32_t *tile_list = nullptr;
    std::vector<int32_t> tile_list_host;
    std::vector<int32_t> tile_place_host;  // with RtRenderArgs::part_span > 1: where in the framebuffer each listed tile goes
    uint32_t tile_key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // tile -> position lookup of rt_untile_device
    uint32_t *tile_where = nullptr;
    uint32_t untile_key[5] = {0, 0, 0, 0, 0};
    // device framebuffer and page-locked staging buffer of rt_render / rt_render_async
    float *fb = nullptr;
    size_t fb_floats = 0;
    float *pinned = nullptr;
    size_t pinned_floats = 0;
};

struct RtDeviceScene {
    RtBackend *backend = nullptr;  // identity check only; may already be destroyed when the scene is freed
    int device = 0;
    RtSceneView view{};       // reference-order mode: the reference's scene-level tree
    RtSceneView view_fast{};  // fast mode: the same arrays, entered through the fast mode's scene-level tree
    std::vector<void *> allocs;
    uint32_t width = 0, height = 0;
    uint32_t n_point_lights = 0;
    bool has_mirror = false;
    bool has_lit = false;
    bool has_ggx = false;  // a shape uses GGXMirror: frames go through the ray-tree kernel
};

namespace rt {

// Enqueues one frame (or one part of it) on `stream`.  `stats` non-null: every kernel class is timed,
// the call waits for the frame and checks the overflow word.  `blocking`: the caller is going to wait
// for the frame anyway (and calls finish_stream itself), so the progress callback may be served,
// which takes a wait per batch.
// `static_memory`: the batch buffers are laid out in the device's static work array (RtBackend::scratch)
// and the call makes no allocation of any kind: whole frames only, no statistics, no timing events.
int render_frame(RtBackend *be, const RtDeviceScene *ds, const RtRenderArgs *args, float *fb_dev, const RtRayIds *ids,
                 hipStream_t stream, RtStats *stats, bool blocking, bool static_memory);

// Waits for `stream` and turns a set overflow word into an error (clearing it).
int finish_stream(RtBackend *be, hipStream_t stream);

// Grow-only device framebuffer / page-locked staging buffer of a backend.
int ensure_device_fb(RtBackend *be, size_t floats);
int ensure_pinned(RtBackend *be, size_t floats);

// True when `p` is host memory the device can write directly (hipHostMalloc / hipHostRegister).
bool is_pinned_host(const void *p);

}  // namespace rt
