#!/usr/bin/env python3
"""Regenerates integration/patches/*.patch from a checkout of the reference.

    python3 integration/make_patches.py /path/to/rust-raytracer

The edits are spelled out below as (file, anchor text, replacement) so that the intent is readable
here; the patch files are `diff -urN` of the checkout before and after.  Only needed when the
reference moves: the patches themselves are committed.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))

FLATTEN_SHAPE = {
    "raytracer-lib/src/geometry/sphere.rs": '''
    fn flatten(&self, out: &mut crate::flat::Flattener) -> crate::flat::FlatShape {
        crate::flat::FlatShape {
            name: self.name.to_string(),
            shader: out.shader(&self.shader),
            geometry: crate::flat::FlatGeometry::Sphere { center: self.center, radius: self.radius },
        }
    }
''',
    "raytracer-lib/src/geometry/cuboid.rs": '''
    fn flatten(&self, out: &mut crate::flat::Flattener) -> crate::flat::FlatShape {
        crate::flat::FlatShape {
            name: self.name.to_string(),
            shader: out.shader(&self.shader),
            geometry: crate::flat::FlatGeometry::Cuboid { min: self.bbox.min, max: self.bbox.max },
        }
    }
''',
    "raytracer-lib/src/geometry/triangle.rs": '''
    fn flatten(&self, out: &mut crate::flat::Flattener) -> crate::flat::FlatShape {
        crate::flat::FlatShape {
            name: self.name.to_string(),
            shader: out.shader(&self.shader),
            geometry: crate::flat::FlatGeometry::Triangle { a: self.a, b: self.b, c: self.c },
        }
    }
''',
    "raytracer-lib/src/geometry/mesh.rs": '''
    fn flatten(&self, out: &mut crate::flat::Flattener) -> crate::flat::FlatShape {
        let mesh = out.mesh(self.triangles.clone());
        crate::flat::FlatShape {
            name: self.name.to_string(),
            shader: out.shader(&self.shader),
            geometry: crate::flat::FlatGeometry::Mesh { mesh },
        }
    }
''',
    "raytracer-lib/src/geometry/instance.rs": '''
    fn flatten(&self, out: &mut crate::flat::Flattener) -> crate::flat::FlatShape {
        let prototype = out.prototype(&self.shape);
        crate::flat::FlatShape {
            name: self.name.to_string(),
            shader: out.shader(&self.shader),
            geometry: crate::flat::FlatGeometry::Instance {
                prototype,
                inv_transform: crate::flat::row_major(&self.inv_transform),
                normal_matrix: crate::flat::row_major(&self.normal_matrix),
                bbox_min: self.bbox.min,
                bbox_max: self.bbox.max,
            },
        }
    }
''',
}

FLATTEN_SHADER = {
    "raytracer-lib/src/shader/lambertian.rs": ("impl Shader for LambertianShader {\n",
                                               "        crate::flat::FlatShader::Lambertian { diffuse: self.diffuse }\n"),
    "raytracer-lib/src/shader/blinn_phong.rs": ("impl Shader for BlinnPhongShader {\n",
                                                "        crate::flat::FlatShader::BlinnPhong { diffuse: self.diffuse, specular: self.specular, shininess: self.shininess }\n"),
    "raytracer-lib/src/shader/perfect_mirror.rs": ("impl Shader for PerfectMirrorShader {\n", "        crate::flat::FlatShader::PerfectMirror\n"),
    "raytracer-lib/src/shader/normal.rs": ("impl Shader for NormalShader {\n", "        crate::flat::FlatShader::Normal\n"),
    "raytracer-lib/src/shader/null.rs": ("impl super::Shader for NullShader {\n", "        crate::flat::FlatShader::Null\n"),
    "raytracer-lib/src/shader/ggx_mirror.rs": ("impl Shader for GGXMirrorShader {\n",
                                               "        crate::flat::FlatShader::GGXMirror { roughness: self.roughness, samples: self.samples }\n"),
}


def edit(root, rel, old, new, count=1):
    path = os.path.join(root, rel)
    text = open(path).read()
    if text.count(old) < 1:
        raise SystemExit(f"{rel}: anchor not found: {old[:60]!r}")
    open(path, "w").write(text.replace(old, new, count))


def apply_lib(root):
    shutil.copy(os.path.join(HERE, "new_files", "raytracer-lib", "src", "flat.rs"), os.path.join(root, "raytracer-lib", "src", "flat.rs"))
    edit(root, "raytracer-lib/src/lib.rs", "mod camera;\n", "mod camera;\npub mod flat;\n")
    edit(root, "raytracer-lib/src/lib.rs", "pub use antialias::AntialiasMethod;\n",
         "pub use antialias::AntialiasMethod;\npub use flat::FlatScene;\n")
    # Shape: one more required method, implemented next to the private fields
    edit(root, "raytracer-lib/src/geometry/mod.rs",
         "    fn closest_hit<'hit>(&'hit self, hit: &mut crate::shader::Hit<'hit>) -> bool;\n}",
         "    fn closest_hit<'hit>(&'hit self, hit: &mut crate::shader::Hit<'hit>) -> bool;\n"
         "    /// This shape as plain data (see `crate::flat`); shared shaders, prototypes and meshes go through `out`.\n"
         "    fn flatten(&self, out: &mut crate::flat::Flattener) -> crate::flat::FlatShape;\n}")
    for rel, body in FLATTEN_SHAPE.items():
        text = open(os.path.join(root, rel)).read()
        at = text.rindex("\n}")  # end of `impl Shape for ...`, the last item of each file
        open(os.path.join(root, rel), "w").write(text[:at] + "\n" + body.rstrip("\n") + text[at:])
    # BBox::min / max are read by Cuboid and Instance above
    bbox = open(os.path.join(root, "raytracer-lib/src/geometry/bbox.rs")).read()
    if "pub min" not in bbox:
        edit(root, "raytracer-lib/src/geometry/bbox.rs", "    min: P3,\n    max: P3,\n", "    pub min: P3,\n    pub max: P3,\n")
    # Mesh keeps its triangles in file order (they are otherwise only reachable inside its BVH)
    edit(root, "raytracer-lib/src/geometry/mesh.rs", "pub struct Mesh {\n    bvh: BVH,\n", "pub struct Mesh {\n    bvh: BVH,\n    triangles: crate::flat::FlatMesh,\n")
    edit(root, "raytracer-lib/src/geometry/mesh.rs", "        let triangles = model\n", "        let shapes = model\n")
    edit(root, "raytracer-lib/src/geometry/mesh.rs", "        let bvh = BVH::new(triangles);\n",
         "        let triangles = model\n            .mesh\n            .indices\n            .chunks(3)\n"
         "            .map(|i| [positions[i[0] as usize], positions[i[1] as usize], positions[i[2] as usize]])\n"
         "            .collect::<crate::flat::FlatMesh>();\n        let bvh = BVH::new(shapes);\n")
    edit(root, "raytracer-lib/src/geometry/mesh.rs", "        Self {\n            bvh,\n", "        Self {\n            bvh,\n            triangles,\n")
    # Shader
    edit(root, "raytracer-lib/src/shader/mod.rs", "    fn apply(&self, hit: &Hit) -> Color;\n}",
         "    fn apply(&self, hit: &Hit) -> Color;\n    /// This shader's parameters as plain data (see `crate::flat`).\n    fn flatten(&self) -> crate::flat::FlatShader;\n}")
    for rel, (anchor, body) in FLATTEN_SHADER.items():
        edit(root, rel, anchor, anchor + "    fn flatten(&self) -> crate::flat::FlatShader {\n" + body + "    }\n\n")
    # Light
    edit(root, "raytracer-lib/src/light/mod.rs", "    fn illuminates(&self, hit: &crate::shader::Hit) -> Option<V3>;\n}",
         "    fn illuminates(&self, hit: &crate::shader::Hit) -> Option<V3>;\n    fn flatten(&self) -> crate::flat::FlatLight;\n}")
    edit(root, "raytracer-lib/src/light/point.rs", "impl Light for PointLight {\n",
         "impl Light for PointLight {\n    fn flatten(&self) -> crate::flat::FlatLight {\n"
         "        crate::flat::FlatLight::Point { position: self.position, intensity: self.intensity }\n    }\n\n")
    edit(root, "raytracer-lib/src/light/ambient.rs", "impl Light for AmbientLight {\n",
         "impl Light for AmbientLight {\n    fn flatten(&self) -> crate::flat::FlatLight {\n"
         "        crate::flat::FlatLight::Ambient { intensity: self.intensity }\n    }\n\n")
    # Camera: the image-plane rectangle is private to CameraBase
    edit(root, "raytracer-lib/src/camera/mod.rs", "    fn camera_base_mut(&mut self) -> &mut CameraBase;\n}",
         "    fn camera_base_mut(&mut self) -> &mut CameraBase;\n\n    /// This camera as plain data (see `crate::flat`).\n    fn flatten(&self) -> crate::flat::FlatCamera;\n}")
    edit(root, "raytracer-lib/src/camera/mod.rs", "    pub fn get_uv(&self,",
         "    pub(crate) fn flatten(&self, kind: crate::flat::FlatCameraKind) -> crate::flat::FlatCamera {\n"
         "        crate::flat::FlatCamera {\n            kind,\n            pixels_x: self.pixels_x,\n            pixels_y: self.pixels_y,\n"
         "            position: self.basis.position,\n            u: self.basis.u,\n            v: self.basis.v,\n            w: self.basis.w,\n"
         "            left: self.left,\n            right: self.right,\n            top: self.top,\n            bottom: self.bottom,\n        }\n    }\n\n"
         "    pub fn get_uv(&self,")
    edit(root, "raytracer-lib/src/camera/perspective.rs", "    fn camera_base(&self) -> &CameraBase {\n",
         "    fn flatten(&self) -> crate::flat::FlatCamera {\n"
         "        self.base.flatten(crate::flat::FlatCameraKind::Perspective { focal_length: self.focal_length })\n    }\n\n"
         "    fn camera_base(&self) -> &CameraBase {\n")
    edit(root, "raytracer-lib/src/camera/orthographic.rs", "    fn camera_base(&self) -> &CameraBase {\n",
         "    fn flatten(&self) -> crate::flat::FlatCamera {\n"
         "        self.base.flatten(crate::flat::FlatCameraKind::Orthographic)\n    }\n\n"
         "    fn camera_base(&self) -> &CameraBase {\n")


def apply_cli(root):
    edit(root, "raytracer-cli/Cargo.toml", 'image = "0.24"\n',
         'image = "0.24"\n# the MI355X back end (librt_hip.so behind a C ABI); only linked with `--features hip`\n'
         'raytracer-hip = { path = "../raytracer-hip", optional = true }\n\n[features]\nhip = ["dep:raytracer-hip"]\n')
    edit(root, "Cargo.toml", '"raytracer-cli", ', '"raytracer-cli", "raytracer-hip", ')
    edit(root, "raytracer-cli/src/main.rs", "#[derive(Parser, Debug)]\n",
         "/// Where the frame is rendered: on the host by raytracer-lib, or on an AMD GPU by librt_hip.so.\n"
         "#[derive(Debug, Clone, Copy, PartialEq, ValueEnum)]\nenum Backend {\n    Cpu,\n    Hip,\n}\n\n#[derive(Parser, Debug)]\n")
    edit(root, "raytracer-cli/src/main.rs", "    antialias_method: Option<AntialiasMethod>,\n}",
         "    antialias_method: Option<AntialiasMethod>,\n"
         "    #[arg(long = \"backend\", value_enum, default_value = \"cpu\")]\n    backend: Backend,\n"
         "    /// GPUs to split the frame over (hip back end)\n    #[arg(long = \"gpus\", default_value_t = 1)]\n    gpus: u32,\n}")
    edit(root, "raytracer-cli/src/main.rs",
         "    let fb = render(&scene, sqrt_rays_per_pixel, aa_method, Some(&per_pixel_cb));\n",
         "    let fb = match args.backend {\n"
         "        Backend::Cpu => render(&scene, sqrt_rays_per_pixel, aa_method, Some(&per_pixel_cb)),\n"
         "        Backend::Hip => render_hip(&scene, sqrt_rays_per_pixel, aa_method, &per_pixel_cb, args.gpus)\n"
         "            .unwrap_or_else(|e| {\n"
         "                // unsupported scene features (RT_ERR_UNSUPPORTED) and missing GPUs end up here\n"
         "                eprintln!(\"hip backend: {e}; rendering on the cpu\");\n"
         "                render(&scene, sqrt_rays_per_pixel, aa_method, Some(&per_pixel_cb))\n"
         "            }),\n    };\n")
    edit(root, "raytracer-cli/src/main.rs", "fn main() -> Result<(), Box<dyn std::error::Error>> {\n",
         "#[cfg(feature = \"hip\")]\nfn render_hip(\n    scene: &raytracer_lib::Scene,\n    sqrt_rays_per_pixel: u16,\n"
         "    aa_method: raytracer_lib::AntialiasMethod,\n    per_pixel_cb: &dyn Fn(),\n    gpus: u32,\n"
         ") -> Result<raytracer_lib::Framebuffer, String> {\n"
         "    let flat = scene.flatten();\n"
         "    raytracer_hip::render(&flat, sqrt_rays_per_pixel, aa_method, Some(per_pixel_cb), gpus).map_err(|e| e.to_string())\n}\n\n"
         "#[cfg(not(feature = \"hip\"))]\nfn render_hip(\n    _scene: &raytracer_lib::Scene,\n    _sqrt_rays_per_pixel: u16,\n"
         "    _aa_method: raytracer_lib::AntialiasMethod,\n    _per_pixel_cb: &dyn Fn(),\n    _gpus: u32,\n"
         ") -> Result<raytracer_lib::Framebuffer, String> {\n"
         "    Err(\"this binary was built without `--features hip`\".to_string())\n}\n\n"
         "fn main() -> Result<(), Box<dyn std::error::Error>> {\n")


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    work = tempfile.mkdtemp()
    for side in ("a", "b"):
        for sub in ("raytracer-lib", "raytracer-cli"):
            shutil.copytree(os.path.join(ref, sub), os.path.join(work, side, sub), ignore=shutil.ignore_patterns("target", "*.lock"))
        shutil.copy(os.path.join(ref, "Cargo.toml"), os.path.join(work, side, "Cargo.toml"))
    apply_lib(os.path.join(work, "b"))
    apply_cli(os.path.join(work, "b"))
    head = ("# NOT COMPILED: the image these patches were written in has no Rust toolchain.\n"
            "# Apply from the root of reecelikesramen/rust-raytracer with `patch -p1 < this file`.\n")
    for name, subs in (("raytracer-lib-flatten.patch", ["raytracer-lib"]), ("raytracer-cli-backend.patch", ["raytracer-cli", "Cargo.toml"])):
        text = head
        for sub in subs:
            r = subprocess.run(["diff", "-urN", os.path.join("a", sub), os.path.join("b", sub)], cwd=work, capture_output=True, text=True)
            text += r.stdout
        open(os.path.join(HERE, "patches", name), "w").write(text)
        print(name, text.count("\n@@"), "hunks")
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
