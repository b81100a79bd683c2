//! Raw bindings: one `#[repr(C)]` struct per struct of `include/rt_scene.h` and `include/rt_hip.h`,
//! one declaration per exported `rt_*` function.  Field order, names and types follow the headers
//! line by line (`tests/test_rust_bindings.py` in the back end's repository checks exactly that).
//!
//! NOT COMPILED: the image this file was written in has no Rust toolchain.
#![allow(non_camel_case_types, non_upper_case_globals)]

use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_PARSE: c_int = -2;
pub const RT_ERR_UNSUPPORTED: c_int = -3;
pub const RT_ERR_DEVICE: c_int = -4;
pub const RT_ERR_IO: c_int = -5;

pub const RT_CAMERA_PERSPECTIVE: i32 = 0;
pub const RT_CAMERA_ORTHOGRAPHIC: i32 = 1;

pub const RT_SHADER_LAMBERTIAN: i32 = 0;
pub const RT_SHADER_BLINN_PHONG: i32 = 1;
pub const RT_SHADER_PERFECT_MIRROR: i32 = 2;
pub const RT_SHADER_NORMAL: i32 = 3;
pub const RT_SHADER_NULL: i32 = 4;
pub const RT_SHADER_GGX_MIRROR: i32 = 5;

pub const RT_LIGHT_POINT: i32 = 0;
pub const RT_LIGHT_AMBIENT: i32 = 1;

pub const RT_SHAPE_SPHERE: i32 = 0;
pub const RT_SHAPE_BOX: i32 = 1;
pub const RT_SHAPE_TRIANGLE: i32 = 2;
pub const RT_SHAPE_MESH: i32 = 3;
pub const RT_SHAPE_INSTANCE: i32 = 4;

pub const RT_AA_NORMAL: i32 = 0;
pub const RT_AA_JITTERED: i32 = 1;
pub const RT_AA_RANDOM: i32 = 2;

pub const RT_MODE_FAST: i32 = 0;
pub const RT_MODE_REFERENCE_ORDER: i32 = 1;

pub const RT_RENDER_NO_PACKETS: u32 = 1;
pub const RT_RENDER_ROW_MAJOR_PIXELS: u32 = 2;
pub const RT_RENDER_UNFUSED_SHADOWS: u32 = 4;
pub const RT_RENDER_UNFUSED_CAMERA: u32 = 8;
pub const RT_RENDER_TREE_DEPTH_FIRST: u32 = 16;
pub const RT_RENDER_CLOSEST_BLOCKERS: u32 = 32;

pub const RT_UPLOAD_REFERENCE_MESH_TREES: u32 = 1;

// kernel classes: indices into RtStats::kernel_ms / kernel_launches
pub const RT_K_RAYGEN: usize = 0;
pub const RT_K_TRACE_CLOSEST: usize = 1;
pub const RT_K_SHADE: usize = 2;
pub const RT_K_TRACE_SHADOW: usize = 3;
pub const RT_K_ACCUMULATE: usize = 4;
pub const RT_K_RESOLVE: usize = 5;
pub const RT_K_COUNT: usize = 8;
pub const RT_COMM_ID_BYTES: usize = 128;
pub const RT_STATIC_SCRATCH_BYTES: usize = 64 << 20;

// ---------------------------------------------------------------- include/rt_scene.h

#[repr(C)]
#[derive(Debug, Clone, Copy, Default, PartialEq)]
pub struct RtVec3 {
    pub x: f64,
    pub y: f64,
    pub z: f64,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct RtCamera {
    pub r#type: i32,
    pub pixels_x: u32,
    pub pixels_y: u32,
    pub _pad: u32,
    pub position: RtVec3,
    pub u: RtVec3,
    pub v: RtVec3,
    pub w: RtVec3,
    pub left: f64,
    pub right: f64,
    pub top: f64,
    pub bottom: f64,
    pub focal_length: f64,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct RtShader {
    pub r#type: i32,
    pub diffuse: [f32; 3],
    pub specular: [f32; 3],
    pub shininess: f32,
    pub roughness: f64,
    pub samples: u32,
    pub _pad: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct RtLight {
    pub r#type: i32,
    pub intensity: [f32; 3],
    pub position: RtVec3,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct RtShape {
    pub r#type: i32,
    pub shader: i32,
    pub mesh: i32,
    pub proto: i32,
    pub p0: RtVec3,
    pub p1: RtVec3,
    pub p2: RtVec3,
    pub radius: f64,
    pub inv_transform: [f64; 16],
    pub normal_matrix: [f64; 16],
    pub bbox_min: RtVec3,
    pub bbox_max: RtVec3,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct RtMesh {
    pub n_triangles: u64,
    pub vertices: *const f64,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct RtScene {
    pub camera: RtCamera,
    pub background: [f32; 3],
    pub recursion_depth: u16,
    pub disable_shadows: u8,
    pub _pad0: u8,
    pub image_width: u32,
    pub image_height: u32,
    pub n_shapes: u32,
    pub n_protos: u32,
    pub n_meshes: u32,
    pub n_shaders: u32,
    pub n_lights: u32,
    pub _pad1: u32,
    pub shapes: *const RtShape,
    pub protos: *const RtShape,
    pub meshes: *const RtMesh,
    pub shaders: *const RtShader,
    pub lights: *const RtLight,
}

// ---------------------------------------------------------------- include/rt_hip.h

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct RtParseArgs {
    pub image_width: u32,
    pub image_height: u32,
    pub aspect_ratio: f64,
    pub recursion_depth: i32,
    pub disable_shadows: u8,
    pub render_normals: u8,
    pub _pad: [u8; 2],
}

#[repr(C)]
pub struct RtBackend {
    _private: [u8; 0],
}

#[repr(C)]
pub struct RtDeviceScene {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct RtUploadOptions {
    pub flags: u32,
    pub _reserved: [u32; 3],
}

pub type RtProgressFn = Option<unsafe extern "C" fn(pixels_done: u64, pixels_total: u64, user: *mut c_void)>;

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct RtRenderArgs {
    pub sqrt_rays_per_pixel: u16,
    pub timed_classes: u16,
    pub antialias_method: i32,
    pub mode: i32,
    pub tile_w: u32,
    pub tile_h: u32,
    pub part_index: u32,
    pub part_count: u32,
    pub part0_weight: u32,
    pub seed: u64,
    pub progress: RtProgressFn,
    pub progress_user: *mut c_void,
    pub batch_samples: u32,
    pub xcd_chunk: u32,
    pub flags: u32,
    pub part_rotation: u32,
    /// render_pixel's counterpart (render.rs:53-84): only the pixels of this window, outputs indexed inside it.
    pub window_x0: u32,
    pub window_y0: u32,
    pub window_w: u32,
    pub window_h: u32,
    /// Several parts of the frame in one set of launches (multi-GPU streams of frames).
    pub part_span: u32,
    pub _reserved0: u32,
    pub part_span_pitch_floats: u64,
}

impl RtRenderArgs {
    /// All zero: the library's defaults everywhere, no callback.
    pub fn zeroed() -> Self {
        RtRenderArgs {
            sqrt_rays_per_pixel: 0,
            timed_classes: 0,
            antialias_method: RT_AA_NORMAL,
            mode: RT_MODE_FAST,
            tile_w: 0,
            tile_h: 0,
            part_index: 0,
            part_count: 0,
            part0_weight: 0,
            seed: 0,
            progress: None,
            progress_user: std::ptr::null_mut(),
            batch_samples: 0,
            xcd_chunk: 0,
            flags: 0,
            part_rotation: 0,
            window_x0: 0,
            window_y0: 0,
            window_w: 0,
            window_h: 0,
            part_span: 0,
            _reserved0: 0,
            part_span_pitch_floats: 0,
        }
    }
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct RtRayIds {
    pub primary_shape: *mut i32,
    pub primary_tri: *mut i32,
    pub final_shape: *mut i32,
    pub final_tri: *mut i32,
    pub bounces: *mut i32,
    pub shadow_shape: *mut i32,
    pub shadow_tri: *mut i32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct RtStats {
    pub primary_rays: u64,
    pub secondary_rays: u64,
    pub shadow_rays: u64,
    pub shaded_hits: u64,
    pub render_ms: f64,
    pub kernel_ms: [f64; RT_K_COUNT],
    pub kernel_launches: [u32; RT_K_COUNT],
}

#[repr(C)]
pub struct RtMulti {
    _private: [u8; 0],
}

#[repr(C)]
pub struct RtComm {
    _private: [u8; 0],
}

extern "C" {
    pub fn rt_parse_scene(scene_json: *const c_char, scene_data_path: *const c_char, args: *const RtParseArgs,
                          scene_out: *mut *mut RtScene) -> c_int;
    pub fn rt_parse_scene_file(scene_path: *const c_char, args: *const RtParseArgs, scene_out: *mut *mut RtScene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut RtScene);
    pub fn rt_backend_create(device_ordinal: c_int, backend_out: *mut *mut RtBackend) -> c_int;
    pub fn rt_backend_destroy(backend: *mut RtBackend);
    pub fn rt_scene_upload(backend: *mut RtBackend, scene: *const RtScene, dev_scene_out: *mut *mut RtDeviceScene) -> c_int;
    pub fn rt_scene_free(dev_scene: *mut RtDeviceScene);
    pub fn rt_scene_upload_with(backend: *mut RtBackend, scene: *const RtScene, options: *const RtUploadOptions,
                                dev_scene_out: *mut *mut RtDeviceScene) -> c_int;
    pub fn rt_render(backend: *mut RtBackend, dev_scene: *const RtDeviceScene, args: *const RtRenderArgs, fb_rgb: *mut f32,
                     ids_host: *const RtRayIds, stats_out: *mut RtStats) -> c_int;
    pub fn rt_host_alloc(bytes: usize, ptr_out: *mut *mut c_void) -> c_int;
    pub fn rt_host_free(ptr: *mut c_void);
    pub fn rt_render_async(backend: *mut RtBackend, dev_scene: *const RtDeviceScene, args: *const RtRenderArgs,
                           fb_rgb_pinned: *mut f32) -> c_int;
    pub fn rt_backend_sync(backend: *mut RtBackend) -> c_int;
    pub fn rt_render_device(backend: *mut RtBackend, dev_scene: *const RtDeviceScene, args: *const RtRenderArgs,
                            fb_rgb_dev: *mut f32, ids_dev: *const RtRayIds, hip_stream: *mut c_void, stats_out: *mut RtStats) -> c_int;
    pub fn rt_render_device_static(backend: *mut RtBackend, dev_scene: *const RtDeviceScene, args: *const RtRenderArgs,
                                   fb_rgb_dev: *mut f32, ids_dev: *const RtRayIds, hip_stream: *mut c_void) -> c_int;
    pub fn rt_collect_timings(backend: *mut RtBackend, out: *mut RtStats) -> c_int;
    pub fn rt_partition_info(width: u32, height: u32, args: *const RtRenderArgs, tiles_total: *mut u32, tiles_owned: *mut u32,
                             tiles_owned_max: *mut u32) -> c_int;
    pub fn rt_partition_tiles(width: u32, height: u32, args: *const RtRenderArgs, tiles_out: *mut i32, capacity: u32,
                              count_out: *mut u32) -> c_int;
    pub fn rt_untile_device(backend: *mut RtBackend, width: u32, height: u32, args: *const RtRenderArgs, gathered_dev: *const f32,
                            fb_rgb_dev: *mut f32, hip_stream: *mut c_void) -> c_int;
    pub fn rt_untile_device_pitched(backend: *mut RtBackend, width: u32, height: u32, args: *const RtRenderArgs,
                                    gathered_dev: *const f32, part_pitch_floats: usize, fb_rgb_dev: *mut f32,
                                    hip_stream: *mut c_void) -> c_int;
    pub fn rt_multi_create(backends: *const *mut RtBackend, dev_scenes: *const *mut RtDeviceScene, n: u32,
                           multi_out: *mut *mut RtMulti) -> c_int;
    pub fn rt_multi_destroy(multi: *mut RtMulti);
    pub fn rt_render_multi(multi: *mut RtMulti, args: *const RtRenderArgs, fb_rgb: *mut f32, stats_out: *mut RtStats) -> c_int;
    pub fn rt_comm_id(id_out: *mut u8) -> c_int;
    pub fn rt_comm_rccl_version(version_out: *mut c_int) -> c_int;
    pub fn rt_comm_create(backend: *mut RtBackend, id: *const u8, rank: u32, world: u32, comm_out: *mut *mut RtComm) -> c_int;
    pub fn rt_comm_destroy(comm: *mut RtComm);
    pub fn rt_comm_gather(comm: *mut RtComm, part_dev: *const f32, floats: usize, gathered_dev: *mut f32, root: u32,
                          hip_stream: *mut c_void) -> c_int;
    pub fn rt_comm_gather_frames(comm: *mut RtComm, parts_dev: *const f32, floats: usize, count: u32, first_root: u32,
                                 slots: u32, gathered_dev: *mut f32, hip_stream: *mut c_void) -> c_int;
    pub fn rt_trace_rays(backend: *mut RtBackend, dev_scene: *const RtDeviceScene, n: u64, origins: *const f64,
                         directions: *const f64, tmin: *const f64, tmax: *const f64, mode: i32, t_out: *mut f32,
                         shape_out: *mut i32, tri_out: *mut i32) -> c_int;
    pub fn rt_framebuffer_to_rgb8(fb_rgb: *const f32, width: u32, height: u32, rgb8_out: *mut u8) -> c_int;
    pub fn rt_save_png(path: *const c_char, fb_rgb: *const f32, width: u32, height: u32) -> c_int;
    pub fn rt_scene_point_light_count(dev_scene: *const RtDeviceScene) -> u32;
    pub fn rt_last_error() -> *const c_char;
}
