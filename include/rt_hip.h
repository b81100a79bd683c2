/*
 * rt_hip.h -- C ABI of librt_hip.so, the MI355X (gfx950) render backend.
 *
 * The library replaces one call of the reference:
 *
 *     raytracer_lib::render(&Scene, sqrt_rays_per_pixel, AntialiasMethod, cb) -> Framebuffer
 *         (raytracer-lib/src/render.rs:7-23, called from raytracer-cli/src/main.rs:100)
 *
 * and mirrors the one call that must run before it so that a `Scene` exists at
 * all without a Rust toolchain:
 *
 *     raytracer_lib::parse_scene(json, data_path, w, h, aspect, depth, shadows, normals)
 *         (raytracer-lib/src/scene/mod.rs:378-715)
 *
 * Every function returns RT_OK (0) or a negative RT_ERR_* code and never
 * throws or aborts; rt_last_error() returns the message of the calling
 * thread's last failure.  All pointers are plain host pointers unless a
 * parameter is documented as a device pointer.  No function retains a caller
 * pointer after it returns.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)     /* bad argument, or a scene the reference rejects with Err */
#define RT_ERR_PARSE (-2)       /* malformed JSON / OBJ (scene/mod.rs:388) */
#define RT_ERR_UNSUPPORTED (-3) /* the reference panics here (unimplemented!()), or a limit of this
                                   library (GGXMirror beyond recursion depth 16, trees too deep
                                   for the LDS traversal stack, an image format other than PNG) */
#define RT_ERR_DEVICE (-4)      /* HIP or RCCL runtime failure */
#define RT_ERR_IO (-5)          /* file could not be read */

/* ------------------------------------------------------------------ */
/* Scene construction on the host (mirror of parse_scene)              */
/* ------------------------------------------------------------------ */

/* The Option<> arguments of parse_scene (scene/mod.rs:378-387); a zero /
 * negative value stands for None. */
typedef struct RtParseArgs {
    uint32_t image_width;    /* 0 -> DEFAULT_IMAGE_WIDTH 360  (prelude.rs:54) */
    uint32_t image_height;   /* 0 -> DEFAULT_IMAGE_HEIGHT 360 (prelude.rs:55) */
    double aspect_ratio;     /* <= 0 -> width / height        (scene/mod.rs:400) */
    int32_t recursion_depth; /* < 0 -> DEFAULT_RECURSION_DEPTH 3 (prelude.rs:57) */
    uint8_t disable_shadows;
    uint8_t render_normals;
    uint8_t _pad[2];
} RtParseArgs;

/* parse_scene(scene_json, scene_data_path, ...) -- scene/mod.rs:378.
 * `scene_data_path` is the directory mesh `file` entries are relative to. */
int rt_parse_scene(const char *scene_json, const char *scene_data_path,
                   const RtParseArgs *args, RtScene **scene_out);

/* The CLI's read_to_string + parent() + parse_scene sequence -- main.rs:51-67. */
int rt_parse_scene_file(const char *scene_path, const RtParseArgs *args, RtScene **scene_out);

/* Frees a scene returned by rt_parse_scene / rt_parse_scene_file. */
void rt_scene_destroy(RtScene *scene);

/* ------------------------------------------------------------------ */
/* Backend and device-resident scene                                    */
/* ------------------------------------------------------------------ */

typedef struct RtBackend RtBackend;
typedef struct RtDeviceScene RtDeviceScene;

/* A backend owns the ray queues and work buffers of ONE frame in flight on one GPU
 * (`device_ordinal` is the HIP device index).  Calls on one backend must not overlap
 * (one host thread at a time, and the frames it enqueues run one after the other); to keep
 * several frames in flight on a GPU create several backends and give each its own stream. */
int rt_backend_create(int device_ordinal, RtBackend **backend_out);
void rt_backend_destroy(RtBackend *backend);

/* Flattens the scene into HBM: builds the top-level and per-mesh BVHs with the
 * reference's split rule (geometry/bvh.rs:36-87), linearises them, and uploads
 * fp32 traversal records plus the f64 originals.  Replaces the tree build that
 * parse_scene performs at scene/mod.rs:699-700 and geometry/mesh.rs:59.
 * RT_MODE_REFERENCE_ORDER walks exactly these trees.  RT_MODE_FAST, whose results do
 * not depend on the visiting order, walks surface-area trees of its own, built here
 * over the shapes and triangles the reference's trees can reach (a leaf whose box
 * has zero thickness is never entered, geometry/bbox.rs:89,104,119).
 * The scene lies on `backend`'s GPU and is only ever read: it may be rendered through ANY backend of
 * that GPU, also by several at once.  Frames in flight on several backends should share one uploaded
 * scene -- the caches then hold one copy of the trees instead of one per backend (four frames in flight
 * on the 1000 x 1000 bunny frame: 0.640 -> 0.632 ms per frame, 0.625 -> 0.602 left on the device).
 * Free it after the last frame that reads it has finished (rt_scene_free waits for the device). */
int rt_scene_upload(RtBackend *backend, const RtScene *scene, RtDeviceScene **dev_scene_out);
void rt_scene_free(RtDeviceScene *dev_scene);

/* rt_scene_upload with build options (NULL or all zero = rt_scene_upload).
 * RT_UPLOAD_REFERENCE_MESH_TREES: RT_MODE_FAST walks the reference's mesh trees
 * (geometry/mesh.rs:59) instead of surface-area trees of its own; same hits, for
 * comparing the two. */
enum { RT_UPLOAD_REFERENCE_MESH_TREES = 1 };
typedef struct RtUploadOptions {
    uint32_t flags; /* RT_UPLOAD_* */
    uint32_t _reserved[3];
} RtUploadOptions;
int rt_scene_upload_with(RtBackend *backend, const RtScene *scene, const RtUploadOptions *options,
                         RtDeviceScene **dev_scene_out);

/* ------------------------------------------------------------------ */
/* render                                                               */
/* ------------------------------------------------------------------ */

/* antialias.rs:6-10, 38-57.  Offsets (di, dj) of sample (p, q) inside its pixel:
 *   NORMAL    ((p + 0.5) / n, (q + 0.5) / n)
 *   JITTERED  ((p + r1) / n, (q + r2) / n)
 *   RANDOM    (r1, r2)
 * with r1, r2 uniform in [0, 1).  The reference draws them from rand::thread_rng(), which
 * is seeded by the operating system, so no two of its runs agree; here they come from a
 * counter-based generator, Philox4x32-10 with key = RtRenderArgs.seed (low word, high
 * word) and counter = (sample index low word, high word, 0, 0), sample index =
 * (i + j * width) * n * n + p * n + q.  Of the four output words x0..x3,
 * r1 = ((x0 << 32 | x1) >> 11) * 2^-53 and r2 likewise from x2, x3 (53 random bits, like
 * rand's f64).  A frame therefore depends only on (scene, n, method, seed): not on the
 * tiling, the partition over GPUs or the traversal mode.
 * GGXMirror (ggx_mirror.rs:33-34, two uniform numbers per reflected ray) draws from the same
 * generator, keyed by the ray's PATH in its sample's ray tree: the camera ray has path 0, the
 * k-th ray (k = 0, 1, ...) spawned at the GGX hit of a ray of path P has path (P << 8) + k + 1
 * (64 bits, wrapping), a perfect mirror passes its ray's path on; the pair that shapes a ray has
 * counter (sample index low word, high word, path low word, path high word).  A path is never 0,
 * so these counters never meet the sample offsets'.  A ray's numbers thus do not depend on what
 * the other rays of its tree hit: the tree may be traced in any order (rand::thread_rng() makes
 * the order of the reference's recursion unobservable). */
enum { RT_AA_NORMAL = 0, RT_AA_JITTERED = 1, RT_AA_RANDOM = 2 };

/* Traversal policy.  Both produce the same closest hits except at exact ties.
 *  FAST:            children visited near-first; shadow rays stop at the first
 *                   blocker (only "occluded or not" is defined for them).
 *  REFERENCE_ORDER: children visited left-then-right as geometry/bvh.rs:112-122
 *                   does, shadow rays run to the closest blocker as
 *                   light/point.rs:35 does, so blocker ids are comparable. */
enum { RT_MODE_FAST = 0, RT_MODE_REFERENCE_ORDER = 1 };

typedef struct RtRenderArgs {
    uint16_t sqrt_rays_per_pixel; /* render.rs:9 */
    /* Bit k set: time the kernels of class RT_K_<k> of this frame with HIP events WITHOUT waiting
     * for them; rt_collect_timings adds them up later.  Only looked at when the call is given no
     * stats_out (which times every class and waits).  0 = no events at all. */
    uint16_t timed_classes;
    int32_t antialias_method;     /* render.rs:10 */
    int32_t mode;                 /* RT_MODE_* */
    /* Image partition for multi-GPU rendering.  The image is cut into
     * tile_w x tile_h tiles; tile (tx, ty) belongs to part
     * (tx + skew * ty) % part_count, where skew is chosen per tile grid and part
     * count so that part sizes are as equal as a lattice allows (they differ by
     * at most a tile or two) and a part's tiles are spread evenly over the
     * image in both directions (8 parts: skew 3, nearest tiles of a part 2.8
     * tiles apart).  rt_partition_tiles lists a part's tiles (row-major tile
     * numbers, ascending); rt_partition_info gives the counts.
     * part_count <= 1 renders everything.
     * part0_weight = w in 1..7 gives part 0 -- the device that also receives the other parts'
     * tiles and restores the image layout -- a share w/8 of what each other part gets: the
     * lattice then deals the tiles to w + 8 * (part_count - 1) lanes, of which part 0 owns w,
     * evenly spaced, and every other part eight.  0 or 8: equal shares.  All parts of a frame,
     * and rt_untile_device, must be given the same value. */
    uint32_t tile_w, tile_h;      /* 0 -> 32 */
    uint32_t part_index, part_count;
    uint32_t part0_weight;        /* 0 */
    uint64_t seed;                /* key of the sample-offset generator (JITTERED / RANDOM) */
    /* render.rs:11, 80-82: the reference calls per_pixel_cb once for every finished pixel, on the
     * calling thread (the CLI advances a progress bar with it, main.rs:86-100).  A frame here
     * finishes in batches of tiles, so the callback gets counts instead: NULL, or a function that
     * a BLOCKING render (rt_render; rt_render_device with stats_out) calls on the calling thread
     * after every finished batch with the pixels of this part finished so far and their total.
     * The last call has pixels_done == pixels_total.  Never called by a render that only
     * enqueues. */
    void (*progress)(uint64_t pixels_done, uint64_t pixels_total, void *user);
    void *progress_user;
    /* Tuning; 0 = the library's choice everywhere.  None of them changes a pixel (RT_RENDER_TREE_DEPTH_FIRST, for
     * scenes with a GGXMirror shader, changes colours in the last places of an f32: see there).
     *  batch_samples  most samples (pixels x rays per pixel) worked on at a time; a larger frame is
     *                 rendered in several batches of whole tiles (0 -> 32 Mi, about 8.5 GB of work buffers; at most 2^30)
     *  xcd_chunk      consecutive 256-sample blocks handed to one XCD (one L2) at a time (rounded down to a power of two)
     *  flags          RT_RENDER_* below */
    uint32_t batch_samples;
    uint32_t xcd_chunk;
    uint32_t flags;
    /* rt_untile_device(_pitched) only, with part_count > 1: block b of the gathered buffer holds the tiles
     * of part (b + part_rotation) % part_count (0: block p holds part p).  For a stream of frames whose
     * parts rotate over the GPUs -- GPU r renders part (r + k) % N of frame k -- so that a part that costs
     * more than the others does not always fall on the same GPU. */
    uint32_t part_rotation;
    /* render_pixel(scene, i, j, ...) -- render.rs:53-84, the reference's entry point for "these pixels"
     * (lib.rs:21; its incremental caller renders the next n pixels with it).  A window of window_w x
     * window_h pixels whose lower left pixel is (window_x0, window_y0) of the frame (row 0 = bottom, as in
     * the framebuffer); window_w = 0 or window_h = 0: the whole frame.  Only the window's pixels are
     * rendered, and every output then covers the window instead of the frame: the framebuffer holds
     * window_w * window_h * 3 floats, pixel (i, j) of the frame at ((i - x0) + (j - y0) * window_w) * 3,
     * and sample s of RtRayIds is ((i - x0) + (j - y0) * window_w) * n*n + p * n + q.  A pixel's value does
     * not depend on the window it is rendered in (the stochastic sample offsets are keyed by the sample's
     * index in the FRAME); a 1 x 1 window is render_pixel.  Whole windows only: part_count <= 1. */
    uint32_t window_x0, window_y0, window_w, window_h;
    /* Several parts of the frame in ONE set of launches (part_count > 1; 0 or 1: the one part part_index).
     * The call renders the parts part_index, part_index + 1, ... (mod part_count), part_span of them; part
     * (part_index + j) % part_count leaves its compact block of tiles (as it would alone) at
     * part_span_pitch_floats * j floats into the framebuffer (0: tiles_owned_max * tile_w * tile_h * 3, the blocks
     * end to end as a gather buffer holds them; else a whole number of tiles, at least that).  The tiles of all
     * these parts are worked on together in image order.  For a GPU whose share of a frame is so small that the
     * fixed cost of a frame's launches -- a kernel lasts as long as its slowest wave -- rivals the share itself:
     * what it owes a group of frames it can then render in fewer, larger launches.  No counterpart in the
     * reference.  rt_render scatters the pixels of all these parts; not for rt_render_async / _static. */
    uint32_t part_span;
    uint32_t _reserved0;
    uint64_t part_span_pitch_floats;
} RtRenderArgs;

enum {
    RT_RENDER_NO_PACKETS = 1,       /* RT_MODE_FAST: one ray per lane for camera and shadow rays too */
    RT_RENDER_ROW_MAJOR_PIXELS = 2, /* work the pixels of a tile row by row instead of in small blocks */
    RT_RENDER_UNFUSED_SHADOWS = 4,  /* RT_MODE_FAST: one shadow-ray launch per light and a separate launch for the
                                       light sums, instead of one kernel for both */
    RT_RENDER_UNFUSED_CAMERA = 8,   /* RT_MODE_FAST: the camera rays and the stage that decides what happens to their
                                       hits as two launches (the hit records go through memory) instead of one, and
                                       likewise two launches per mirror bounce instead of one for all bounces */
    RT_RENDER_CLOSEST_BLOCKERS = 32, /* RT_MODE_FAST: shadow rays run to the CLOSEST blocker as light/point.rs:35 does (the
                                       mode's own shadow rays stop at the first one they find), through the separate
                                       shadow kernel: RtRayIds.shadow_shape / shadow_tri then compare with the reference's
                                       one to one, as they do in RT_MODE_REFERENCE_ORDER, at several times that mode's
                                       speed.  Pixels are the same as without it */
    RT_RENDER_TREE_DEPTH_FIRST = 16 /* scenes with a GGXMirror shader: one thread walks a sample's whole ray tree, depth
                                       first.  Without it RT_MODE_FAST traces the trees level by level through the
                                       packet kernels, four times as fast, but every call -- the enqueue-only ones
                                       too -- then waits for the device once per level (it reads how many rays the
                                       next level has); with it rt_render_async and rt_render_device only enqueue,
                                       as they do for every other scene.  Same rays and random numbers either way;
                                       colours agree to a few units in the last place of an f32 */
};

/* Optional per-sample outputs.  Sample s of pixel (i, j) is
 *   s = (i + j * width) * n*n + p * n + q     (render.rs:64-65 loop order).
 * Any pointer may be NULL.  Shape ids index RtScene.shapes; tri ids index the
 * mesh's triangles in file order (-1 when the hit is not inside a mesh).
 * -1 in a shape id means "no hit".
 * Scenes that use GGXMirror (shader/ggx_mirror.rs) trace a TREE of rays per sample: only
 * primary_shape / primary_tri are defined for them (final_* = -1, bounces = 0, shadow_* = -2). */
typedef struct RtRayIds {
    int32_t *primary_shape; /* first hit of the camera ray */
    int32_t *primary_tri;
    int32_t *final_shape;   /* hit that was finally shaded after mirror bounces */
    int32_t *final_tri;
    int32_t *bounces;       /* mirror bounces traced for this sample */
    /* n_point_lights entries per sample, in RtScene.lights order restricted to
     * point lights: blocking shape id, -1 = light visible, -2 = no shadow ray. */
    int32_t *shadow_shape;
    int32_t *shadow_tri;
} RtRayIds;

enum {
    RT_K_RAYGEN = 0,
    RT_K_TRACE_CLOSEST = 1,
    RT_K_SHADE = 2,
    RT_K_TRACE_SHADOW = 3,
    RT_K_ACCUMULATE = 4,
    RT_K_RESOLVE = 5,
    RT_K_COUNT = 8
};

typedef struct RtStats {
    uint64_t primary_rays;   /* width * height * n*n of the rendered part */
    uint64_t secondary_rays; /* mirror bounces traced (shader/perfect_mirror.rs:27) */
    uint64_t shadow_rays;    /* shaded hits x point lights (light/point.rs:35) */
    uint64_t shaded_hits;
    double render_ms;               /* device time, first kernel start to last kernel end */
    double kernel_ms[RT_K_COUNT];   /* summed per kernel class, HIP events on the render stream */
    uint32_t kernel_launches[RT_K_COUNT];
} RtStats;

/* render(&scene, sqrt_rpp, aa, cb) -> Framebuffer  -- render.rs:7-23.
 * `fb_rgb` is caller-owned host memory, width*height*3 floats, laid out as
 * Framebuffer::pixels (framebuffer.rs:21-31): index (i + j*width)*3, row j = 0
 * is the bottom of the image (with RtRenderArgs.window_*: the window's pixels, see there).  With part_count > 1 only the owned tiles'
 * pixels are written.  Blocking. */
int rt_render(RtBackend *backend, const RtDeviceScene *dev_scene, const RtRenderArgs *args,
              float *fb_rgb, const RtRayIds *ids_host, RtStats *stats_out);

/* Page-locked host memory for framebuffers (no counterpart in the reference, whose Framebuffer is
 * a Vec): the device writes a frame into such memory directly, at the speed of the bus.  rt_render
 * recognises it (and any other memory registered with the HIP runtime) and skips its staging copy;
 * rt_render_async requires it. */
int rt_host_alloc(size_t bytes, void **ptr_out);
void rt_host_free(void *ptr);

/* render (render.rs:7-23) without waiting: enqueues the frame and its copy into `fb_rgb_pinned`
 * (page-locked host memory, width*height*3 floats, whole frames only: part_count <= 1) on the
 * backend's stream and returns.  rt_backend_sync waits for it.  With one backend per frame in
 * flight the copy of frame k over the bus overlaps the kernels of frame k+1. */
int rt_render_async(RtBackend *backend, const RtDeviceScene *dev_scene, const RtRenderArgs *args,
                    float *fb_rgb_pinned);

/* Waits for everything enqueued through this backend (rt_render_async, rt_render_device without
 * stats_out) and reports what only the device could notice: RT_ERR_DEVICE if a kernel failed or a
 * frame since the last check was cut short (traversal stack overflow, an internal error). */
int rt_backend_sync(RtBackend *backend);

/* Same, but `fb_rgb_dev` and the `ids_dev` arrays are DEVICE pointers and the
 * work is enqueued on `hip_stream` (a hipStream_t, NULL = the backend's own
 * stream, which is created non-blocking: work on it is NOT ordered with the
 * caller's null-stream work, so pass the stream that produces / consumes the
 * buffers).  Returns after enqueueing when stats_out is NULL; with stats_out it
 * waits for the stream and fills the counters and timings.  The stream may be destroyed once the
 * call has returned: rt_backend_sync and rt_backend_destroy wait for an event recorded behind the
 * frame, not for the stream. */
int rt_render_device(RtBackend *backend, const RtDeviceScene *dev_scene, const RtRenderArgs *args,
                     float *fb_rgb_dev, const RtRayIds *ids_dev, void *hip_stream,
                     RtStats *stats_out);

/* rt_render_device for callers that cannot allocate -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE (whose render allocates
 * its framebuffer and, per hit, its light lists).  The call makes no allocation of any kind, on the
 * device or through the HIP runtime: the ray queues, hit records, lit table and counters of a batch
 * are laid out in a static array of RT_STATIC_SCRATCH_BYTES that is part of the library's code
 * object (every GPU the library runs on holds its own copy from load until the process ends; its
 * device address is looked up once, in rt_backend_create).  Same pixels as rt_render_device: a frame
 * larger than the array is rendered in more batches of fewer tiles.  Limits: whole frames only
 * (part_count <= 1); no statistics, timed_classes is ignored, progress is not called; ONE tile
 * (tile_w x tile_h x rays per pixel samples, about 240 + 12 x point lights bytes each) must fit the
 * array, else RT_ERR_INVALID.  All calls on one GPU share that GPU's array, whatever backend they go
 * through, and the array keeps the last frame's contents (each frame clears the counters it needs).  Frames
 * in the array therefore never overlap: the library orders them itself -- an event is recorded behind every
 * such frame and the next one on that GPU, whatever backend and stream it comes through, begins with a
 * wait for it on the device -- so two streams may both be used but gain nothing over one.  Two host
 * threads inside this call for one GPU at the same time would interleave their enqueues: the second gets
 * RT_ERR_INVALID. */
#define RT_STATIC_SCRATCH_BYTES (64u << 20)
int rt_render_device_static(RtBackend *backend, const RtDeviceScene *dev_scene, const RtRenderArgs *args,
                            float *fb_rgb_dev, const RtRayIds *ids_dev, void *hip_stream);

/* Waits for the frames enqueued with RtRenderArgs::timed_classes since the last call and fills
 * kernel_ms / kernel_launches of `out` with their sums (all other fields 0).  Lets a caller keep
 * many frames in flight and still learn how long a kernel took on the device: every event pair
 * costs a few microseconds of pipeline drain, so timing one class disturbs a frame far less than
 * stats_out does. */
int rt_collect_timings(RtBackend *backend, RtStats *out);

/* Tile bookkeeping of a partitioned render (no counterpart in the reference, which has
 * one thread and one image): how many tiles the image has, how many belong to
 * args->part_index, and the largest count any part owns (= the per-part stride of
 * a gathered buffer).  A part's compact framebuffer holds its tiles in row-major
 * tile order, each tile tile_w*tile_h*3 floats, rows bottom-up, padded at the image
 * edge. */
int rt_partition_info(uint32_t width, uint32_t height, const RtRenderArgs *args, uint32_t *tiles_total,
                      uint32_t *tiles_owned, uint32_t *tiles_owned_max);

/* The row-major tile indices owned by args->part_index, ascending: slot s of that part's
 * compact framebuffer holds tile tiles_out[s].  tiles_out may be NULL to query the count. */
int rt_partition_tiles(uint32_t width, uint32_t height, const RtRenderArgs *args, int32_t *tiles_out,
                       uint32_t capacity, uint32_t *count_out);

/* Rebuilds the full framebuffer (framebuffer.rs:21-31 layout) on the device from the
 * compact framebuffers of all parts laid end to end: `gathered_dev` holds
 * part_count blocks of tiles_owned_max*tile_w*tile_h*3 floats, block p written by the
 * render with part_index = p.  Enqueued on `hip_stream` (NULL = backend stream). */
int rt_untile_device(RtBackend *backend, uint32_t width, uint32_t height, const RtRenderArgs *args,
                     const float *gathered_dev, float *fb_rgb_dev, void *hip_stream);

/* The same with the parts' blocks `part_pitch_floats` apart instead of end to end (the pitch
 * must be at least tiles_owned_max*tile_w*tile_h*3).  For a gather that brings several frames
 * at once: each rank sends its blocks of F frames as one buffer, rank 0 receives
 * [rank][frame][block], and frame f is un-tiled from gathered_dev + f*block with pitch
 * F*block.  No counterpart in the reference (its render writes the framebuffer directly). */
int rt_untile_device_pitched(RtBackend *backend, uint32_t width, uint32_t height, const RtRenderArgs *args,
                             const float *gathered_dev, size_t part_pitch_floats, float *fb_rgb_dev,
                             void *hip_stream);

/* render (render.rs:7-23) with the frame split over several backends in ONE process -- what a
 * front end with a `--gpus N` flag calls.  An RtMulti binds n backends and the same scene uploaded
 * through each (backends[k] / dev_scenes[k] render part k of n); the backends may sit on
 * different devices or share one.  It owns what a frame needs beyond the backends: one host thread
 * per part that enqueues the part's kernels, the gather buffers on backends[0]'s device, and an
 * RCCL communicator with one rank per distinct device (ncclCommInitAll; librccl is loaded when the
 * first RtMulti or RtComm is created, not with this library).
 * rt_render_multi: every part renders its tiles (part 0 straight into the gather buffer), ONE
 * grouped ncclSend / ncclRecv brings the other parts' tiles to backends[0]'s device over xGMI,
 * a kernel there restores the image layout, and the frame is returned in host memory like
 * rt_render's.  args->part_index / part_count are ignored (set per part), part0_weight is
 * honoured, progress is not called.  stats_out (may be NULL): ray counts and kernel times summed
 * over the parts (this waits for every part), render_ms = wall time of the call. */
typedef struct RtMulti RtMulti;
int rt_multi_create(RtBackend *const *backends, RtDeviceScene *const *dev_scenes, uint32_t n, RtMulti **multi_out);
void rt_multi_destroy(RtMulti *multi);
int rt_render_multi(RtMulti *multi, const RtRenderArgs *args, float *fb_rgb, RtStats *stats_out);

/* The same gather for a launcher with ONE PROCESS PER GPU (MPI, torch.distributed): rank 0 makes
 * an id and hands it to the other ranks by whatever means the launcher has, every rank creates its
 * end of the communicator (ncclCommInitRank) on its backend's device, and rt_comm_gather enqueues
 * on `hip_stream`: this rank's `floats` floats at `part_dev` go to rank `root`, which receives
 * rank r's at gathered_dev + r * floats (gathered_dev is ignored on the other ranks).  Collective:
 * every rank must call it, with the same `floats` and `root`.  A caller that renders a stream of
 * frames can name a different root for every frame (or group of frames): receiving, restoring the
 * image layout and the copy to the host over that GPU's own PCIe link are then shared by all GPUs
 * instead of falling on one. */
#define RT_COMM_ID_BYTES 128
typedef struct RtComm RtComm;
int rt_comm_id(uint8_t id_out[RT_COMM_ID_BYTES]);
/* The version code of the RCCL library the communicators run on (ncclGetVersion: major * 10000 + minor * 100 +
 * patch), which loads it if no communicator has yet.  For a benchmark's record; no counterpart in the reference. */
int rt_comm_rccl_version(int *version_out);
int rt_comm_create(RtBackend *backend, const uint8_t id[RT_COMM_ID_BYTES], uint32_t rank, uint32_t world,
                   RtComm **comm_out);
void rt_comm_destroy(RtComm *comm);
int rt_comm_gather(RtComm *comm, const float *part_dev, size_t floats, float *gathered_dev, uint32_t root,
                   void *hip_stream);

/* A GROUP of frames in one exchange, the root rotating from frame to frame: frame f of the group
 * (f = 0 .. count-1) is gathered to rank (first_root + f) % world, so every GPU receives, un-tiles and
 * sends home its share of the frames over its own PCIe link, and every xGMI link carries tiles in both
 * directions at once instead of seven links converging on one GPU.  One call posts ONE send per
 * (rank, root) pair and its receive, all inside one group.
 * Layout, the same on every rank, in blocks of `floats` floats: with p = f % world and s = f / world,
 * this rank's part of frame f lies at block p * slots + s of `parts_dev` (the frames of one root are
 * adjacent, so they travel as one message), and the frame's root receives rank r's part of it at block
 * r * slots + s of `gathered_dev`: rt_untile_device_pitched with gathered_dev + s * floats and a part
 * pitch of slots * floats restores frame f there.  `slots` >= ceil(count / world) is the caller's choice
 * (a fixed number lets a last, shorter group use the same buffers); both buffers hold world * slots
 * blocks; gathered_dev may be NULL on a rank that is no frame's root.  Collective: every rank calls it
 * with the same floats, count, first_root and slots.  rt_comm_gather is the case count = 1, slots = 1,
 * first_root = root.  No counterpart in the reference. */
int rt_comm_gather_frames(RtComm *comm, const float *parts_dev, size_t floats, uint32_t count, uint32_t first_root,
                          uint32_t slots, float *gathered_dev, void *hip_stream);

/* BVH::closest_hit on caller-supplied rays -- geometry/bvh.rs:140-142 with a
 * Hit record initialised as {t = tmax[r], t_min = tmin[r]} (shader/hit_struct.rs:17-39).
 * origins/directions: n x 3 doubles (host).  Outputs (host): t (f32, unchanged
 * tmax when nothing is hit), shape id, tri id. */
int rt_trace_rays(RtBackend *backend, const RtDeviceScene *dev_scene, uint64_t n,
                  const double *origins, const double *directions, const double *tmin,
                  const double *tmax, int32_t mode, float *t_out, int32_t *shape_out,
                  int32_t *tri_out);

/* save_to_png's pixel conversion -- raytracer-cli/src/output/png_export.rs:4-10:
 * out(x, y) = round(clamp(fb(x, height-1-y), 0, 1) * 255) in f32, rows top-down.
 * `rgb8_out` holds width*height*3 bytes. */
int rt_framebuffer_to_rgb8(const float *fb_rgb, uint32_t width, uint32_t height, uint8_t *rgb8_out);

/* output::save -- raytracer-cli/src/output/mod.rs:7-17 with save_to_png
 * (png_export.rs:1-14): writes the framebuffer as an 8-bit RGB PNG.  A path whose
 * extension is not "png" is RT_ERR_UNSUPPORTED (the reference panics with
 * unimplemented!); a path without extension writes nothing, as there. */
int rt_save_png(const char *path, const float *fb_rgb, uint32_t width, uint32_t height);

/* Number of point lights of an uploaded scene (the stride of RtRayIds.shadow_*). */
uint32_t rt_scene_point_light_count(const RtDeviceScene *dev_scene);

/* Message of the calling thread's most recent failure ("" if none). */
const char *rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
