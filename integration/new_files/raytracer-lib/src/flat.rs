//! A plain-data description of a parsed `Scene`, for render back ends that live outside this crate.
//!
//! NOT COMPILED where this file was written (no Rust toolchain in that build image).
//!
//! The render path works on trait objects whose fields are private (`Sphere::center`,
//! `BlinnPhongShader::shininess`, `Instance::inv_transform`, ...), so another crate cannot read a
//! `Scene`.  `Scene::flatten()` hands out exactly the values the constructors stored, in scene
//! order, with the trait objects replaced by enums: what `raytracer-hip` converts into the C structs
//! of `include/rt_scene.h`.  Derived data (bounding boxes of plain shapes, triangle normals, the
//! BVHs) is not carried: a back end rebuilds it with the rules of `geometry/bvh.rs`.

use std::collections::HashMap;

use crate::prelude::*;

#[derive(Debug, Clone, Copy, PartialEq)]
pub enum FlatCameraKind {
    Perspective { focal_length: Real },
    Orthographic,
}

/// `camera::CameraBase` plus the concrete camera's own field.
#[derive(Debug, Clone, PartialEq)]
pub struct FlatCamera {
    pub kind: FlatCameraKind,
    pub pixels_x: u32,
    pub pixels_y: u32,
    pub position: P3,
    pub u: V3,
    pub v: V3,
    pub w: V3,
    pub left: Real,
    pub right: Real,
    pub top: Real,
    pub bottom: Real,
}

#[derive(Debug, Clone, PartialEq)]
pub enum FlatShader {
    Lambertian { diffuse: Color },
    BlinnPhong { diffuse: Color, specular: Color, shininess: f32 },
    PerfectMirror,
    Normal,
    Null,
    GGXMirror { roughness: Real, samples: u32 },
}

#[derive(Debug, Clone, PartialEq)]
pub enum FlatLight {
    Point { position: P3, intensity: Color },
    Ambient { intensity: Color },
}

#[derive(Debug, Clone, PartialEq)]
pub enum FlatGeometry {
    Sphere { center: P3, radius: Real },
    Cuboid { min: P3, max: P3 },
    Triangle { a: P3, b: P3, c: P3 },
    /// index into `FlatScene::meshes`
    Mesh { mesh: usize },
    /// `Instance`: index into `FlatScene::prototypes`, the two matrices `Instance::new` computed
    /// (row-major) and the world-space box it stored
    Instance {
        prototype: usize,
        inv_transform: [Real; 16],
        normal_matrix: [Real; 16],
        bbox_min: P3,
        bbox_max: P3,
    },
}

#[derive(Debug, Clone, PartialEq)]
pub struct FlatShape {
    pub name: String,
    /// index into `FlatScene::shaders`
    pub shader: usize,
    pub geometry: FlatGeometry,
}

/// One triangle per index triple of the OBJ file, in file order (`Mesh::new`).
pub type FlatMesh = Vec<[P3; 3]>;

#[derive(Debug, Clone)]
pub struct FlatScene {
    pub camera: FlatCamera,
    pub background_color: Color,
    pub recursion_depth: u16,
    pub disable_shadows: bool,
    pub image_width: u32,
    pub image_height: u32,
    /// top-level shapes, in `Scene::shapes` order
    pub shapes: Vec<FlatShape>,
    /// shapes referenced by instances
    pub prototypes: Vec<FlatShape>,
    pub meshes: Vec<FlatMesh>,
    pub shaders: Vec<FlatShader>,
    /// in `Scene::lights` order
    pub lights: Vec<FlatLight>,
}

/// Collects the shared objects (shaders, instance prototypes, meshes) once each while the shapes
/// of a scene flatten themselves.  Identity is the address of the `Arc`'s target.
#[derive(Debug, Default)]
pub struct Flattener {
    pub(crate) shaders: Vec<FlatShader>,
    pub(crate) prototypes: Vec<FlatShape>,
    pub(crate) meshes: Vec<FlatMesh>,
    shader_index: HashMap<usize, usize>,
    prototype_index: HashMap<usize, usize>,
}

impl Flattener {
    /// Index of `shader` in `FlatScene::shaders`, flattening it on first sight.
    pub fn shader(&mut self, shader: &std::sync::Arc<dyn crate::shader::Shader>) -> usize {
        let key = std::sync::Arc::as_ptr(shader) as *const () as usize;
        if let Some(&index) = self.shader_index.get(&key) {
            return index;
        }
        self.shaders.push(shader.flatten());
        self.shader_index.insert(key, self.shaders.len() - 1);
        self.shaders.len() - 1
    }

    /// Index of `shape` in `FlatScene::prototypes`, flattening it on first sight.
    pub fn prototype(&mut self, shape: &std::sync::Arc<dyn crate::geometry::Shape>) -> usize {
        let key = std::sync::Arc::as_ptr(shape) as *const () as usize;
        if let Some(&index) = self.prototype_index.get(&key) {
            return index;
        }
        let flat = shape.flatten(self);
        self.prototypes.push(flat);
        self.prototype_index.insert(key, self.prototypes.len() - 1);
        self.prototypes.len() - 1
    }

    pub fn mesh(&mut self, triangles: FlatMesh) -> usize {
        self.meshes.push(triangles);
        self.meshes.len() - 1
    }
}

/// Row-major copy of a homogeneous matrix.
pub(crate) fn row_major(m: &na::Matrix4<Real>) -> [Real; 16] {
    let mut out = [0.0; 16];
    for r in 0..4 {
        for c in 0..4 {
            out[r * 4 + c] = m[(r, c)];
        }
    }
    out
}

impl crate::scene::Scene {
    /// Everything `render` reads from this scene, as plain data.
    pub fn flatten(&self) -> FlatScene {
        let mut f = Flattener::default();
        let shapes = self.shapes.iter().map(|s| s.flatten(&mut f)).collect();
        FlatScene {
            camera: self.camera.flatten(),
            background_color: self.background_color,
            recursion_depth: self.recursion_depth,
            disable_shadows: self.disable_shadows,
            image_width: self.image_width,
            image_height: self.image_height,
            shapes,
            prototypes: f.prototypes,
            meshes: f.meshes,
            shaders: f.shaders,
            lights: self.lights.iter().map(|l| l.flatten()).collect(),
        }
    }
}
