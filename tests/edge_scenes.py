"""Small hand-written scenes for corner cases of the render path (shared by the CPU and GPU parity tests).
Each entry: name -> (scene dict, width, height, sqrt_rpp)."""
import json
import os

CAM = [{"_name": "main", "_type": "perspective", "position": "0.013 0.021 6", "viewDir": "0.002 -0.003 -1", "focalLength": 1.0}]
SHADERS = [
    {"_name": "lam", "_type": "Lambertian", "diffuse": "0.7 0.3 0.2"},
    {"_name": "bp", "_type": "BlinnPhong", "diffuse": "0.2 0.5 0.7", "specular": "1 1 1", "phongExp": 24.0},
    {"_name": "mir", "_type": "Mirror"},
]
POINT = {"_type": "point", "position": "3.1 4.3 5.2", "intensity": "1 1 1"}
AMBIENT = {"_type": "ambient", "intensity": "0.15 0.15 0.2"}
BALLS = [
    {"_name": "a", "_type": "sphere", "shader": "lam", "center": "-1.1 0 0", "radius": 1.0},
    {"_name": "b", "_type": "sphere", "shader": "bp", "center": "1.2 0.1 -0.5", "radius": 0.8},
    {"_name": "floor", "_type": "box", "shader": "lam", "minPt": "-6 -1.5 -6", "maxPt": "6 -1.0 6"},
]


def _scene(**kw):
    s = {"camera": CAM, "shader": SHADERS}
    s.update(kw)
    return {"scene": s}


def degenerate_obj(path):
    """A quad of two triangles, one zero-area triangle (the reference never hits it: det A == 0) and a sliver."""
    with open(path, "w") as f:
        f.write("o m\nv -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nv 0.5 0.5 0.5\nv 0.5 0.5 0.5\nv 0.25 0.25 0.5\n"
                "v -0.5 0.9 0.4\nv 0.5 0.9000001 0.4\nv 0 0.9 0.45\n"
                "f 1 2 3\nf 1 3 4\nf 5 6 7\nf 8 9 10\n")


def half_flat_obj(path):
    """Sixteen triangles: eight in the plane z = 0 at x < 0, eight tilted ones at x > 0.  The reference's
    tree (median split on x, then y, leaves of four) puts the coplanar ones into leaves whose boxes have
    no thickness, and such a box is never entered (bbox.rs:89,104,119): that half of the mesh is
    invisible in the reference, and must be in every mode here."""
    v, f = [], []
    for half, (x0, tilt) in enumerate([(-1.6, 0.0), (0.4, 0.35)]):
        for q in range(4):
            xa, ya = x0 + 0.6 * (q % 2), -0.6 + 0.6 * (q // 2)
            base = len(v)
            v += [(xa, ya, 0.0), (xa + 0.55, ya, tilt * 0.5), (xa + 0.55, ya + 0.55, tilt), (xa, ya + 0.55, tilt * 0.5)]
            f += [(base + 1, base + 2, base + 3), (base + 1, base + 3, base + 4)]
    with open(path, "w") as out:
        out.write("o m\n" + "".join(f"v {a} {b} {c}\n" for a, b, c in v) + "".join(f"f {a} {b} {c}\n" for a, b, c in f))


def scenes(tmp_dir):
    obj = os.path.join(tmp_dir, "degenerate.obj")
    degenerate_obj(obj)
    half_flat_obj(os.path.join(tmp_dir, "half_flat.obj"))
    out = {
        "no_lights": (_scene(shape=BALLS), 64, 48, 2),
        "ambient_only": (_scene(shape=BALLS, light=[AMBIENT]), 64, 48, 2),
        "one_shape": (_scene(shape=BALLS[:1], light=[POINT]), 48, 48, 1),
        "one_pixel": (_scene(shape=BALLS, light=[POINT, AMBIENT]), 1, 1, 3),
        "wide_aspect": (_scene(shape=BALLS, light=[POINT]), 150, 20, 2),
        "odd_size_many_samples": (_scene(shape=BALLS, light=[POINT, AMBIENT]), 37, 23, 5),
        "camera_inside_box": (_scene(shape=BALLS + [{"_name": "room", "_type": "box", "shader": "bp", "minPt": "-8 -8 -8",
                                                     "maxPt": "8 8 8"}], light=[POINT]), 64, 48, 2),
        "mirror_pair": (_scene(shape=[{"_name": "m1", "_type": "sphere", "shader": "mir", "center": "-1.05 0 0", "radius": 1.0},
                                      {"_name": "m2", "_type": "sphere", "shader": "mir", "center": "1.05 0 0", "radius": 1.0},
                                      BALLS[2]], light=[POINT]), 80, 60, 2),
        "degenerate_mesh": (_scene(shape=[{"_name": "mesh", "_type": "mesh", "shader": "bp", "file": "degenerate.obj"}, BALLS[2]],
                                   light=[POINT, AMBIENT]), 96, 72, 2),
        "half_flat_mesh": (_scene(shape=[{"_name": "mesh", "_type": "mesh", "shader": "bp", "file": "half_flat.obj"}, BALLS[2]],
                                  light=[POINT, AMBIENT]), 96, 72, 2),
        "scaled_instances": (_scene(
            instance=[{"_name": "proto", "_type": "mesh", "shader": "lam", "file": "degenerate.obj"},
                      {"_name": "ball", "_type": "sphere", "shader": "lam", "center": "0 0 0", "radius": 1.0}],
            shape=[{"_name": "i1", "_type": "instance", "shader": "bp", "_id": "proto",
                    "xform": [{"type": "scale", "amount": "1.5 0.7 1"}, {"type": "rotate", "axis": "Y", "amount": 35},
                              {"type": "translate", "amount": "-0.8 0 0"}]},
                   {"_name": "i2", "_type": "instance", "shader": "lam", "_id": "ball",
                    "xform": [{"type": "scale", "amount": "0.4 1.3 0.6"}, {"type": "translate", "amount": "1.4 0 0.5"}]},
                   BALLS[2]], light=[POINT]), 96, 72, 2),
        # Boxes lit at grazing angles: two lights a hair above the plane of the floor's top face (cosines of 1e-6 and
        # 1e-8 on that face, either side of the margin below which the shade stage runs the slab test), one well
        # above; plus an instanced, scaled and rotated box.  (The boxes reach into the floor: faces in one plane
        # with rays that skim it would make exact ties, which an f32 tree search and the f64 reference break apart.)  The shade stage
        # decides "can this shadow ray meet its own box again" without the slab test when the ray clearly leaves
        # the face, and with it otherwise: every answer must be the reference's.
        "boxes_grazing_lights": (_scene(
            instance=[{"_name": "unit", "_type": "box", "shader": "lam", "minPt": "-0.5 -0.5 -0.5", "maxPt": "0.5 0.5 0.5"}],
            shape=[BALLS[2],
                   {"_name": "crate", "_type": "box", "shader": "bp", "minPt": "-1.6 -1.2 -0.9", "maxPt": "-0.4 0.2 0.3"},
                   {"_name": "slab", "_type": "instance", "shader": "lam", "_id": "unit",
                    "xform": [{"type": "scale", "amount": "1.8 0.4 1.1"}, {"type": "rotate", "axis": "Y", "amount": 25},
                              {"type": "translate", "amount": "1.1 -0.9 0.4"}]}],
            light=[{"_type": "point", "position": "5.0 -0.99999 2.0", "intensity": "0.6 0.6 0.6"},
                   {"_type": "point", "position": "-4.0 -0.9999999 3.0", "intensity": "0.5 0.5 0.5"},
                   POINT]), 96, 72, 2),
        # More point lights than the fused shadow kernel has bits for (32): the fast mode then goes through the separate
        # shadow and light-sum kernels; with a mirror in view, so that lit hits behind a bounce take that way too.
        "forty_lights": (_scene(shape=BALLS + [{"_name": "m", "_type": "sphere", "shader": "mir", "center": "0.2 1.4 -1.0", "radius": 0.7}],
                                light=[{"_type": "point", "position": f"{-5 + 0.25 * k:.2f} {3 + 0.05 * (k % 7):.2f} {4 - 0.2 * (k % 5):.2f}",
                                        "intensity": "0.03 0.03 0.03"} for k in range(40)] + [AMBIENT]), 64, 48, 2),
    }
    return {k: (json.dumps(v[0]), tmp_dir) + v[1:] for k, v in out.items()}


NAMES = ["no_lights", "ambient_only", "one_shape", "one_pixel", "wide_aspect", "odd_size_many_samples", "camera_inside_box",
         "mirror_pair", "degenerate_mesh", "half_flat_mesh", "scaled_instances", "boxes_grazing_lights", "forty_lights"]


def ggx_scene(roughness=0.25, samples=3, mirror_type=None, point_lights=None):
    """Two rough mirrors (GGXMirror, ggx_mirror.rs) facing each other and a perfect mirror, over a lit floor
    and a lit ball: GGX hits under GGX hits, GGX under a mirror, lit hits at every depth.  With
    `mirror_type` the rough shader is replaced by that type (for comparisons); `point_lights` replaces the two
    point lights."""
    rough = ({"_name": "rough", "_type": "GGXMirror", "roughness": roughness, "samples": samples} if mirror_type is None
             else {"_name": "rough", "_type": mirror_type})
    shaders = SHADERS + [rough]
    shapes = [
        {"_name": "g1", "_type": "sphere", "shader": "rough", "center": "-1.3 0 0", "radius": 1.0},
        {"_name": "g2", "_type": "box", "shader": "rough", "minPt": "0.9 -1.0 -1.6", "maxPt": "1.1 1.4 0.9"},
        {"_name": "m", "_type": "sphere", "shader": "mir", "center": "0.1 -0.45 1.4", "radius": 0.5},
        {"_name": "ball", "_type": "sphere", "shader": "bp", "center": "0 1.3 -0.8", "radius": 0.6},
        {"_name": "tri", "_type": "triangle", "shader": "lam", "v0": "-3 -1 -3", "v1": "3 -1 -3", "v2": "0 3.5 -3.4"},
        BALLS[2],
    ]
    return json.dumps({"scene": {"camera": CAM, "shader": shaders, "shape": shapes,
                                 "light": (point_lights or [POINT, {"_type": "point", "position": "-4 2.5 3", "intensity": "0.5 0.5 0.6"}]) + [AMBIENT],
                                 "sceneParameters": {"bgColor": "0.2 0.25 0.35"}}})


def two_sphere_scene_of_the_reference_test():
    """The scene of the reference's own (commented-out) BVH test, geometry/bvh.rs:149-183: unit spheres
    at (0, 0, -5) and (2, 0, -5)."""
    import json

    shapes = [{"_name": "sphere1", "_type": "sphere", "shader": "m", "center": "0 0 -5", "radius": 1.0},
              {"_name": "sphere2", "_type": "sphere", "shader": "m", "center": "2 0 -5", "radius": 1.0}]
    return json.dumps({"scene": {
        "camera": [{"_name": "main", "_type": "perspective", "position": "0 0 0", "viewDir": "0 0 -1", "focalLength": 1.0}],
        "shader": [{"_name": "m", "_type": "Lambertian", "diffuse": "1 1 1"}], "shape": shapes}})
