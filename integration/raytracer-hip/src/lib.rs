//! `raytracer-hip`: the render of `raytracer-lib` on AMD MI355X GPUs.
//!
//! The crate keeps the shape of `raytracer_lib::render` (render.rs:7-12):
//!
//! ```ignore
//! let flat = scene.flatten();                                   // raytracer-lib, flat.rs
//! let fb = raytracer_hip::render(&flat, sqrt_rpp, aa, Some(&cb), 1)?;   // -> raytracer_lib::Framebuffer
//! ```
//!
//! and forwards to `librt_hip.so` (hand-written HIP for gfx950 behind the C ABI of
//! `include/rt_hip.h`).  `Backend` / `DeviceScene` keep a scene resident on the GPU across frames.
//!
//! NOT COMPILED: the image this file was written in has no Rust toolchain.

pub mod sys;

use std::ffi::CStr;
use std::os::raw::c_void;

use raytracer_lib::flat::{FlatCameraKind, FlatGeometry, FlatLight, FlatScene, FlatShader, FlatShape};
use raytracer_lib::{AntialiasMethod, Framebuffer};

/// An error of the back end: the `RT_ERR_*` code and the message of `rt_last_error`.
#[derive(Debug, Clone)]
pub struct HipError {
    pub code: i32,
    pub message: String,
}

impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "rt error {}: {}", self.code, self.message)
    }
}

impl std::error::Error for HipError {}

fn check(code: i32) -> Result<(), HipError> {
    if code == sys::RT_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(sys::rt_last_error()) }.to_string_lossy().into_owned();
    Err(HipError { code, message })
}

fn v3(p: &raytracer_lib::flat::FlatCamera, which: u8) -> sys::RtVec3 {
    let (x, y, z) = match which {
        0 => (p.position.x, p.position.y, p.position.z),
        1 => (p.u.x, p.u.y, p.u.z),
        2 => (p.v.x, p.v.y, p.v.z),
        _ => (p.w.x, p.w.y, p.w.z),
    };
    sys::RtVec3 { x, y, z }
}

fn color(c: &nalgebra::Vector3<f32>) -> [f32; 3] {
    [c.x, c.y, c.z]
}

const ZERO3: sys::RtVec3 = sys::RtVec3 { x: 0.0, y: 0.0, z: 0.0 };

/// A `FlatScene` laid out as the C structs of `rt_scene.h`.  Owns every array `scene` points into.
pub struct SceneBuffers {
    shapes: Vec<sys::RtShape>,
    protos: Vec<sys::RtShape>,
    mesh_vertices: Vec<Vec<f64>>,
    meshes: Vec<sys::RtMesh>,
    shaders: Vec<sys::RtShader>,
    lights: Vec<sys::RtLight>,
    scene: sys::RtScene,
}

impl SceneBuffers {
    pub fn new(flat: &FlatScene) -> Self {
        let shape = |s: &FlatShape| -> sys::RtShape {
            let mut out = sys::RtShape {
                r#type: 0,
                shader: s.shader as i32,
                mesh: -1,
                proto: -1,
                p0: ZERO3,
                p1: ZERO3,
                p2: ZERO3,
                radius: 0.0,
                inv_transform: [0.0; 16],
                normal_matrix: [0.0; 16],
                bbox_min: ZERO3,
                bbox_max: ZERO3,
            };
            let p = |q: &nalgebra::Point3<f64>| sys::RtVec3 { x: q.x, y: q.y, z: q.z };
            match &s.geometry {
                FlatGeometry::Sphere { center, radius } => {
                    out.r#type = sys::RT_SHAPE_SPHERE;
                    out.p0 = p(center);
                    out.radius = *radius;
                }
                FlatGeometry::Cuboid { min, max } => {
                    out.r#type = sys::RT_SHAPE_BOX;
                    out.p0 = p(min);
                    out.p1 = p(max);
                }
                FlatGeometry::Triangle { a, b, c } => {
                    out.r#type = sys::RT_SHAPE_TRIANGLE;
                    out.p0 = p(a);
                    out.p1 = p(b);
                    out.p2 = p(c);
                }
                FlatGeometry::Mesh { mesh } => {
                    out.r#type = sys::RT_SHAPE_MESH;
                    out.mesh = *mesh as i32;
                }
                FlatGeometry::Instance { prototype, inv_transform, normal_matrix, bbox_min, bbox_max } => {
                    out.r#type = sys::RT_SHAPE_INSTANCE;
                    out.proto = *prototype as i32;
                    out.inv_transform = *inv_transform;
                    out.normal_matrix = *normal_matrix;
                    out.bbox_min = p(bbox_min);
                    out.bbox_max = p(bbox_max);
                }
            }
            out
        };
        let shapes: Vec<sys::RtShape> = flat.shapes.iter().map(shape).collect();
        let protos: Vec<sys::RtShape> = flat.prototypes.iter().map(shape).collect();
        // mesh.rs:39-58: one triangle per index triple, 9 doubles each (a, b, c)
        let mesh_vertices: Vec<Vec<f64>> = flat
            .meshes
            .iter()
            .map(|m| m.iter().flat_map(|t| t.iter().flat_map(|q| [q.x, q.y, q.z])).collect())
            .collect();
        let meshes: Vec<sys::RtMesh> = mesh_vertices
            .iter()
            .map(|v| sys::RtMesh { n_triangles: (v.len() / 9) as u64, vertices: v.as_ptr() })
            .collect();
        let shaders: Vec<sys::RtShader> = flat
            .shaders
            .iter()
            .map(|s| {
                let mut out = sys::RtShader::default();
                match s {
                    FlatShader::Lambertian { diffuse } => {
                        out.r#type = sys::RT_SHADER_LAMBERTIAN;
                        out.diffuse = color(diffuse);
                    }
                    FlatShader::BlinnPhong { diffuse, specular, shininess } => {
                        out.r#type = sys::RT_SHADER_BLINN_PHONG;
                        out.diffuse = color(diffuse);
                        out.specular = color(specular);
                        out.shininess = *shininess;
                    }
                    FlatShader::PerfectMirror => out.r#type = sys::RT_SHADER_PERFECT_MIRROR,
                    FlatShader::Normal => out.r#type = sys::RT_SHADER_NORMAL,
                    FlatShader::Null => out.r#type = sys::RT_SHADER_NULL,
                    FlatShader::GGXMirror { roughness, samples } => {
                        out.r#type = sys::RT_SHADER_GGX_MIRROR;
                        out.roughness = *roughness;
                        out.samples = *samples;
                    }
                }
                out
            })
            .collect();
        let lights: Vec<sys::RtLight> = flat
            .lights
            .iter()
            .map(|l| match l {
                FlatLight::Point { position, intensity } => sys::RtLight {
                    r#type: sys::RT_LIGHT_POINT,
                    intensity: color(intensity),
                    position: sys::RtVec3 { x: position.x, y: position.y, z: position.z },
                },
                FlatLight::Ambient { intensity } => sys::RtLight { r#type: sys::RT_LIGHT_AMBIENT, intensity: color(intensity), position: ZERO3 },
            })
            .collect();
        let cam = &flat.camera;
        let camera = sys::RtCamera {
            r#type: match cam.kind {
                FlatCameraKind::Perspective { .. } => sys::RT_CAMERA_PERSPECTIVE,
                FlatCameraKind::Orthographic => sys::RT_CAMERA_ORTHOGRAPHIC,
            },
            pixels_x: cam.pixels_x,
            pixels_y: cam.pixels_y,
            _pad: 0,
            position: v3(cam, 0),
            u: v3(cam, 1),
            v: v3(cam, 2),
            w: v3(cam, 3),
            left: cam.left,
            right: cam.right,
            top: cam.top,
            bottom: cam.bottom,
            focal_length: match cam.kind {
                FlatCameraKind::Perspective { focal_length } => focal_length,
                FlatCameraKind::Orthographic => 0.0,
            },
        };
        let scene = sys::RtScene {
            camera,
            background: color(&flat.background_color),
            recursion_depth: flat.recursion_depth,
            disable_shadows: flat.disable_shadows as u8,
            _pad0: 0,
            image_width: flat.image_width,
            image_height: flat.image_height,
            n_shapes: shapes.len() as u32,
            n_protos: protos.len() as u32,
            n_meshes: meshes.len() as u32,
            n_shaders: shaders.len() as u32,
            n_lights: lights.len() as u32,
            _pad1: 0,
            shapes: shapes.as_ptr(),
            protos: protos.as_ptr(),
            meshes: meshes.as_ptr(),
            shaders: shaders.as_ptr(),
            lights: lights.as_ptr(),
        };
        // the Vecs are moved into the struct below; moving a Vec does not move its heap buffer,
        // so the pointers taken above stay valid for as long as `self` lives
        SceneBuffers { shapes, protos, mesh_vertices, meshes, shaders, lights, scene }
    }

    pub fn as_ptr(&self) -> *const sys::RtScene {
        &self.scene
    }

    pub fn image_size(&self) -> (u32, u32) {
        (self.scene.image_width, self.scene.image_height)
    }

    pub fn counts(&self) -> (usize, usize, usize, usize, usize) {
        (self.shapes.len(), self.protos.len(), self.mesh_vertices.len(), self.shaders.len(), self.lights.len())
    }
}

/// One GPU (`rt_backend_create`): owns the ray queues of one frame in flight.
pub struct Backend {
    raw: *mut sys::RtBackend,
}

impl Backend {
    pub fn new(device: i32) -> Result<Self, HipError> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::rt_backend_create(device, &mut raw) })?;
        Ok(Backend { raw })
    }

    /// Builds the trees (the reference's split rule, bvh.rs:36-87) and uploads the scene.
    pub fn upload(&self, flat: &FlatScene) -> Result<DeviceScene<'_>, HipError> {
        let buffers = SceneBuffers::new(flat);
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::rt_scene_upload(self.raw, buffers.as_ptr(), &mut raw) })?;
        // the library copies what it needs during the upload: `buffers` may go
        Ok(DeviceScene { backend: self, raw, width: flat.image_width, height: flat.image_height })
    }
}

impl Drop for Backend {
    fn drop(&mut self) {
        unsafe { sys::rt_backend_destroy(self.raw) }
    }
}

/// A scene resident in HBM (`rt_scene_upload`); frames are rendered from it without further uploads.
pub struct DeviceScene<'b> {
    backend: &'b Backend,
    raw: *mut sys::RtDeviceScene,
    width: u32,
    height: u32,
}

struct Progress<'a> {
    cb: &'a dyn Fn(),
    done: u64,
}

unsafe extern "C" fn progress_trampoline(pixels_done: u64, _pixels_total: u64, user: *mut c_void) {
    // render.rs:80-82 calls the callback once per finished pixel; a GPU frame finishes in batches of
    // tiles, so the calls for a batch's pixels are made when the batch is done
    let state = &mut *(user as *mut Progress);
    while state.done < pixels_done {
        (state.cb)();
        state.done += 1;
    }
}

impl<'b> DeviceScene<'b> {
    fn args(&self, sqrt_rays_per_pixel: u16, antialias_method: AntialiasMethod) -> sys::RtRenderArgs {
        let mut args = sys::RtRenderArgs::zeroed();
        args.sqrt_rays_per_pixel = sqrt_rays_per_pixel;
        args.antialias_method = match antialias_method {
            AntialiasMethod::Normal => sys::RT_AA_NORMAL,
            AntialiasMethod::Jittered => sys::RT_AA_JITTERED,
            AntialiasMethod::Random => sys::RT_AA_RANDOM,
        };
        // antialias.rs:45-57 draws from thread_rng(), so no two runs of the reference agree; the back
        // end's offsets come from a counter-based generator keyed by `seed`: fresh entropy per frame
        // keeps that behaviour
        args.seed = rand::random::<u64>();
        args
    }

    /// `render_mut` (render.rs:25-51): fills a caller-owned framebuffer.
    pub fn render_mut(&self, fb: &mut Framebuffer, sqrt_rays_per_pixel: u16, antialias_method: AntialiasMethod,
                      per_pixel_cb: Option<&dyn Fn()>) -> Result<sys::RtStats, HipError> {
        assert!(fb.width == self.width && fb.pixels.len() >= (self.width * self.height) as usize);
        let mut args = self.args(sqrt_rays_per_pixel, antialias_method);
        let mut state = per_pixel_cb.map(|cb| Progress { cb, done: 0 });
        if let Some(s) = state.as_mut() {
            args.progress = Some(progress_trampoline);
            args.progress_user = s as *mut Progress as *mut c_void;
        }
        let mut stats = sys::RtStats::default();
        // Framebuffer::pixels is Vec<[f32; 3]>: the same bytes as the width*height*3 floats rt_render writes,
        // row 0 at the bottom (framebuffer.rs:21-31)
        check(unsafe {
            sys::rt_render(self.backend.raw, self.raw, &args, fb.pixels.as_mut_ptr() as *mut f32, std::ptr::null(), &mut stats)
        })?;
        Ok(stats)
    }

    /// The pixels of a window of the frame (`RtRenderArgs::window_*`): `width * height` colours, row 0 at the
    /// bottom of the window.  A pixel's value is the one it has in the whole frame.
    pub fn render_window(&self, x0: u32, y0: u32, width: u32, height: u32, sqrt_rays_per_pixel: u16,
                         antialias_method: AntialiasMethod, seed: u64) -> Result<Vec<[f32; 3]>, HipError> {
        let mut args = self.args(sqrt_rays_per_pixel, antialias_method);
        args.seed = seed; // a frame rendered piece by piece needs one seed for all its pieces
        args.window_x0 = x0;
        args.window_y0 = y0;
        args.window_w = width;
        args.window_h = height;
        let mut pixels = vec![[0.0f32; 3]; (width as usize) * (height as usize)];
        check(unsafe {
            sys::rt_render(self.backend.raw, self.raw, &args, pixels.as_mut_ptr() as *mut f32, std::ptr::null(), std::ptr::null_mut())
        })?;
        Ok(pixels)
    }

    /// `render_pixel` (render.rs:53-84): one pixel into the caller's framebuffer -- a 1 x 1 window.
    pub fn render_pixel(&self, fb: &mut Framebuffer, sqrt_rays_per_pixel: u16, antialias_method: AntialiasMethod, i: u32,
                        j: u32, seed: u64, per_pixel_cb: Option<&dyn Fn()>) -> Result<(), HipError> {
        let px = self.render_window(i, j, 1, 1, sqrt_rays_per_pixel, antialias_method, seed)?;
        if let Some(cb) = per_pixel_cb {
            cb();
        }
        fb.pixels[(i + j * fb.width) as usize] = px[0]; // Framebuffer::set_pixel, framebuffer.rs:21-31
        Ok(())
    }

    /// `render` (render.rs:7-23).  The reference allocates width x width pixels (render.rs:13); the
    /// first width x height of them are the image, and those are what the back end fills.
    pub fn render(&self, sqrt_rays_per_pixel: u16, antialias_method: AntialiasMethod, per_pixel_cb: Option<&dyn Fn()>)
                  -> Result<Framebuffer, HipError> {
        let mut fb = Framebuffer::new(self.width, self.width.max(self.height));
        fb.height = self.height;
        self.render_mut(&mut fb, sqrt_rays_per_pixel, antialias_method, per_pixel_cb)?;
        Ok(fb)
    }
}

impl Drop for DeviceScene<'_> {
    fn drop(&mut self) {
        unsafe { sys::rt_scene_free(self.raw) }
    }
}

/// Drop-in for `raytracer_lib::render` on a flattened scene; `gpus` > 1 splits the frame over that many
/// devices of this machine (interleaved tiles, one RCCL gather: `rt_render_multi`).
pub fn render(flat: &FlatScene, sqrt_rays_per_pixel: u16, antialias_method: AntialiasMethod, per_pixel_cb: Option<&dyn Fn()>,
              gpus: u32) -> Result<Framebuffer, HipError> {
    if gpus <= 1 {
        let backend = Backend::new(0)?;
        let scene = backend.upload(flat)?;
        return scene.render(sqrt_rays_per_pixel, antialias_method, per_pixel_cb);
    }
    let backends = (0..gpus as i32).map(Backend::new).collect::<Result<Vec<_>, _>>()?;
    let scenes = backends.iter().map(|b| b.upload(flat)).collect::<Result<Vec<_>, _>>()?;
    let raw_backends: Vec<*mut sys::RtBackend> = backends.iter().map(|b| b.raw).collect();
    let raw_scenes: Vec<*mut sys::RtDeviceScene> = scenes.iter().map(|s| s.raw).collect();
    let mut multi = std::ptr::null_mut();
    check(unsafe { sys::rt_multi_create(raw_backends.as_ptr(), raw_scenes.as_ptr(), gpus, &mut multi) })?;
    let mut args = scenes[0].args(sqrt_rays_per_pixel, antialias_method);
    args.tile_w = 32;
    args.tile_h = 32;
    args.part0_weight = if gpus > 4 { 7 } else { 0 }; // the first device also collects the tiles
    let (w, h) = (flat.image_width, flat.image_height);
    let mut fb = Framebuffer::new(w, w.max(h));
    fb.height = h;
    let rc = unsafe { sys::rt_render_multi(multi, &args, fb.pixels.as_mut_ptr() as *mut f32, std::ptr::null_mut()) };
    unsafe { sys::rt_multi_destroy(multi) };
    check(rc)?;
    if let Some(cb) = per_pixel_cb {
        for _ in 0..(w as u64 * h as u64) {
            cb();
        }
    }
    Ok(fb)
}
