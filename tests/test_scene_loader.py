"""Host-side scene construction (rt_parse_scene) against the reference's schema and
parse_scene rules (raytracer-lib/src/scene/mod.rs:27-715, scene/parse_vec3.rs:31-76)."""
import json
import math
import os

import numpy as np
import pytest


def minimal_scene(**over):
    s = {
        "camera": [{"_name": "main", "_type": "perspective", "position": "0 0 5", "viewDir": "0 0 -1", "focalLength": 1.0}],
        "shader": [{"_name": "m", "_type": "Lambertian", "diffuse": "0.5 0.5 0.5"}],
        "shape": [{"_name": "s", "_type": "sphere", "shader": "m", "center": "0 0 0", "radius": 1.0}],
    }
    s.update(over)
    return json.dumps({"scene": s})


def v3(v):
    return np.array([v.x, v.y, v.z])


def test_defaults(rt):
    """DEFAULT_IMAGE_* 360, DEFAULT_RECURSION_DEPTH 3, DEFAULT_BACKGROUND_COLOR 0.198 (prelude.rs:41-58)."""
    sc = rt.parse_scene(minimal_scene(), "")
    c = sc.c
    assert (c.image_width, c.image_height, c.recursion_depth, c.disable_shadows) == (360, 360, 3, 0)
    assert np.allclose(list(c.background), [0.198] * 3)
    cam = c.camera
    # CameraBase::new: plane width 0.5, height 0.5 / aspect (camera/mod.rs:37-50)
    assert (cam.left, cam.right, cam.bottom, cam.top) == (-0.25, 0.25, -0.25, 0.25)
    assert (cam.pixels_x, cam.pixels_y) == (360, 360)
    assert np.allclose(v3(cam.u), [1, 0, 0]) and np.allclose(v3(cam.v), [0, 1, 0]) and np.allclose(v3(cam.w), [0, 0, 1])
    # implicit Null and Normal shaders are appended after the file's shaders
    assert c.n_shaders == 3


def test_aspect_ratio_and_overrides(rt):
    sc = rt.parse_scene(minimal_scene(), "", image_width=200, image_height=100, recursion_depth=6, disable_shadows=True)
    cam = sc.c.camera
    assert (cam.top, cam.bottom) == (0.125, -0.125)  # 0.5 / (200/100) / 2
    assert sc.c.recursion_depth == 6 and sc.c.disable_shadows == 1
    sc = rt.parse_scene(minimal_scene(), "", image_width=200, image_height=100, aspect_ratio=1.0)
    assert sc.c.camera.top == 0.25


def test_vec3_forms_and_aliases(rt):
    scene = minimal_scene(
        sceneParameters={"_bgColor": [0.25, 0.5, 0.75]},
        shader=[{"_name": "bp", "_type": "BlinnPhong", "diffuse": [0.1, 0.2, 0.3], "specular": "1 1 1", "phongExp": 12.5},
                {"_name": "mir", "_type": "Mirror"}, {"_name": "d", "_type": "Dielectric"}],
        shape=[
            {"_name": "a", "_type": "box", "_shader": {"_ref": "bp"}, "minPt": "-1 -1 -1", "maxPt": [1, 1, 1]},
            {"_name": "b", "_type": "box", "shader": "mir", "center": "0 0 0", "extent": "2 4 6"},
            {"_name": "c", "_type": "triangle", "shader": "d", "v0": "0 0 0", "v1": "1 0 0", "v2": "0 1 0"},
        ],
        light=[{"_type": "ambient", "intensity": "0.1 0.1 0.1"}, {"_type": "point", "position": [1, 2, 3], "intensity": "1 1 1"}],
    )
    sc = rt.parse_scene(scene, "")
    c = sc.c
    assert np.allclose(list(c.background), [0.25, 0.5, 0.75])
    assert c.shaders[0].type == 1 and c.shaders[0].shininess == 12.5  # BlinnPhong
    assert c.shaders[1].type == 2  # Mirror alias of PerfectMirror
    assert c.shaders[2].type == 4  # unimplemented shader kinds fall back to NullShader (scene/mod.rs:491)
    assert np.allclose(v3(c.shapes[1].p0), [-1, -2, -3]) and np.allclose(v3(c.shapes[1].p1), [1, 2, 3])
    assert c.n_lights == 2 and sc.n_point_lights == 1
    assert np.allclose(v3(c.lights[1].position), [1, 2, 3])


def test_colour_strings_parse_as_f32(rt):
    sc = rt.parse_scene(minimal_scene(shader=[{"_name": "m", "_type": "Lambertian", "diffuse": "0.1 0.2 0.3"}]), "")
    assert list(sc.c.shaders[0].diffuse) == [np.float32(0.1), np.float32(0.2), np.float32(0.3)]


def test_camera_selection(rt):
    cams = [
        {"_name": "a", "_type": "perspective", "position": "0 0 5", "viewDir": "0 0 -1", "focalLength": 1.0},
        {"_name": "main", "_type": "orthographic", "position": "1 2 3", "lookatPoint": "1 2 0"},
    ]
    sc = rt.parse_scene(minimal_scene(camera=cams), "")
    assert sc.c.camera.type == 1 and np.allclose(v3(sc.c.camera.position), [1, 2, 3])  # DEFAULT_CAMERA "main"
    sc = rt.parse_scene(minimal_scene(camera=cams, sceneParameters={"camera": "a"}), "")
    assert sc.c.camera.type == 0
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene(minimal_scene(camera=cams, sceneParameters={"camera": "zzz"}), "")
    assert e.value.code == rt.RT_ERR_INVALID and "camera zzz not found" in e.value.message


def test_errors_follow_parse_scene(rt):
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene("{ not json", "")
    assert e.value.code == rt.RT_ERR_PARSE
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene(minimal_scene(camera=[]), "")
    assert "at least one camera" in e.value.message
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene(minimal_scene(shape=[{"_name": "s", "_type": "sphere", "shader": "nope", "center": "0 0 0", "radius": 1}]), "")
    assert "non-existent shader" in e.value.message
    two = [{"_name": "s", "_type": "sphere", "shader": "m", "center": "0 0 0", "radius": 1}] * 2
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene(minimal_scene(shape=two), "")
    assert "unique" in e.value.message
    # the reference panics (unimplemented!) on these: reported as UNSUPPORTED
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene(minimal_scene(light=[{"_type": "area", "position": "0 0 0", "intensity": "1 1 1", "normal": "0 1 0", "radius": 1}]), "")
    assert e.value.code == rt.RT_ERR_UNSUPPORTED
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene(minimal_scene(sceneParameters={"envMapPrefix": "x"}), "")
    assert e.value.code == rt.RT_ERR_UNSUPPORTED
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene(minimal_scene(shader=[{"_name": "m", "_type": "Lambertian", "diffuse": {"tex": "t", "data": "1 1 1"}}]), "")
    assert e.value.code == rt.RT_ERR_UNSUPPORTED
    with pytest.raises(rt.RtError):
        rt.parse_scene(minimal_scene(shape=[{"_name": "s", "_type": "sphere", "shader": "m", "center": "0 0", "radius": 1}]), "")


def test_render_normals_replaces_shaders_and_background(rt):
    sc = rt.parse_scene(minimal_scene(sceneParameters={"bgColor": "1 0 0"}), "", render_normals=True)
    assert list(sc.c.background) == [0, 0, 0]
    assert sc.c.shaders[sc.c.shapes[0].shader].type == 3  # NormalShader


def test_instance_matrices(rt, scene_path):
    """Instance::new (geometry/instance.rs:29-42) with the xform accumulation of scene/mod.rs:626-654:
    translations add, scales multiply, a later rotation about the same axis replaces, R = Rz * Ry * Rx."""
    xf = [
        {"type": "translate", "amount": "1 2 3"}, {"type": "rotate", "axis": "X", "amount": 90},
        {"type": "scale", "amount": "2 2 2"}, {"type": "rotation", "axis": "y", "degrees": 30},
        {"type": "rotate", "axis": "X", "amount": 20}, {"type": "translate", "amount": "0 0 1"},
        {"type": "scale", "amount": "1 0.5 1"}, {"type": "rotate", "axis": "Z", "amount": -45},
    ]
    scene = minimal_scene(
        instance=[{"_name": "proto", "_type": "box", "shader": "m", "minPt": "-1 -1 -1", "maxPt": "1 2 3"}],
        shape=[{"_name": "i", "_type": "instance", "shader": "m", "_id": "proto", "xform": xf}],
    )
    sc = rt.parse_scene(scene, "")
    sh = sc.c.shapes[0]
    assert sh.type == 4 and sh.proto == 0

    def rot(axis, deg):
        a = math.pi * deg / 180.0
        c, s = math.cos(a), math.sin(a)
        return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
                "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]

    R = rot("z", -45) @ rot("y", 30) @ rot("x", 20)
    T, S = np.array([1, 2, 4.0]), np.array([2, 1, 2.0])
    M = np.eye(4)
    M[:3, :3] = R * S[None, :]
    M[:3, 3] = T
    inv = np.array(list(sh.inv_transform)).reshape(4, 4)
    nm = np.array(list(sh.normal_matrix)).reshape(4, 4)
    assert np.allclose(inv @ M, np.eye(4), atol=1e-12)
    assert np.allclose(nm[:3, :3], np.linalg.inv(M[:3, :3]).T, atol=1e-12)
    corners = np.array([[x, y, z, 1.0] for x in (-1, 1) for y in (-1, 2) for z in (-1, 3)]) @ M.T
    assert np.allclose(v3(sh.bbox_min), corners[:, :3].min(axis=0), atol=1e-12)
    assert np.allclose(v3(sh.bbox_max), corners[:, :3].max(axis=0), atol=1e-12)


def test_obj_mesh_loading(rt, tmp_path):
    """geometry/mesh.rs:18-58: f32 positions, fan triangulation, negative indices, v/vt/vn tokens, one model only."""
    obj = tmp_path / "m.obj"
    obj.write_text("o thing\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.1 0.2 0.3\nvn 0 0 1\n"
                   "f 1/1/1 2/1/1 3/1/1 4/1/1\nf -1 1 2\n")
    scene = minimal_scene(shape=[{"_name": "mesh", "_type": "mesh", "shader": "m", "file": "m.obj"}])
    sc = rt.parse_scene(scene, str(tmp_path))
    assert sc.c.n_meshes == 1 and sc.c.meshes[0].n_triangles == 3
    v = np.ctypeslib.as_array(sc.c.meshes[0].vertices, shape=(27,)).reshape(3, 3, 3)
    assert np.array_equal(v[0], [[0, 0, 0], [1, 0, 0], [1, 1, 0]])
    assert np.array_equal(v[1], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])
    f = np.float32
    assert np.array_equal(v[2][0], [float(f(0.1)), float(f(0.2)), float(f(0.3))])  # widened from f32
    two = tmp_path / "two.obj"
    two.write_text("o a\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\no b\nf 1 2 3\n")
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene(minimal_scene(shape=[{"_name": "mesh", "_type": "mesh", "shader": "m", "file": "two.obj"}]), str(tmp_path))
    assert "exactly one model" in e.value.message
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene(minimal_scene(shape=[{"_name": "mesh", "_type": "mesh", "shader": "m", "file": "missing.obj"}]), str(tmp_path))
    assert e.value.code == rt.RT_ERR_IO


def test_every_shipped_scene_parses(rt, scene_path):
    """Mirror of raytracer-cli/test_scene_parsing.sh: every scene file must load."""
    for name in ["simple_sphere_scene", "spheres_and_triangles", "cornell_room", "spheres_1K", "bunny", "bunny_small",
                 "bunny_medium", "mixed_features", "mixed_ortho"]:
        sc = rt.parse_scene_file(scene_path(name))
        assert sc.c.n_shapes > 0
    sc = rt.parse_scene_file(scene_path("spheres_1K"))
    assert sc.c.n_shapes == 1000
    sc = rt.parse_scene_file(scene_path("bunny"))
    assert sc.c.meshes[0].n_triangles == 81920
