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


def test_real_bunny_branch_puts_the_mesh_where_the_stand_in_lies(rt):
    """scenes/gen_scenes.py renders scenes/models/bunny.obj instead of the procedural mesh when that file exists
    (geometry/mesh.rs:18-37 semantics come from the scene loader).  Checked with a file that IS the small
    stand-in, scaled and shifted the way a scan's coordinates are: after the normalising transform the scene is
    the stand-in's scene, so the oracle renders the same frame and casts the same rays."""
    import importlib
    import json

    import oracle_lib

    gen = importlib.import_module("scenes.gen_scenes")
    gen.ensure_models(("small",))
    src = os.path.join(gen.MODELS, "bunny_class_1280.obj")
    dst = os.path.join(gen.MODELS, "bunny_test_%d.obj" % os.getpid())
    try:
        with open(src) as f, open(dst, "w") as g:
            for line in f:
                t = line.split()
                if t and t[0] == "v":
                    p = np.array([float(x) for x in t[1:4]]) * 0.09 + np.array([-0.02, 0.11, 0.004])
                    g.write("v %.9g %.9g %.9g\n" % tuple(np.float32(p)))
                elif not (t and t[0] == "o"):
                    g.write(line)
        lo, hi, tris = gen.obj_bounds(dst)
        assert tris == 1280 and (hi - lo).max() < 0.3
        real = gen.bunny_scene_real(os.path.relpath(dst, gen.HERE))
        a = rt.parse_scene(json.dumps({"scene": real}), gen.HERE, image_width=120, image_height=90, recursion_depth=3)
        b = rt.parse_scene_file(os.path.join(gen.HERE, "bunny_small.json"), image_width=120, image_height=90, recursion_depth=3)
        fa, _, ca = oracle_lib.render(a, 2)
        fb, _, cb = oracle_lib.render(b, 2)
        assert ca["rays"] == cb["rays"]
        assert np.abs(fa - fb).max() < 1e-4
    finally:
        if os.path.exists(dst):
            os.remove(dst)


# ---- bit-for-bit restatements.  The render oracle takes camera basis, instance matrices and world boxes from the
# ---- loader under test, so it cannot notice an error in them; these follow the reference's operations in the
# ---- reference's order with Python floats (IEEE f64, no fused multiply-add) and demand identical bits.

def _dot(a, b):  # nalgebra dot of 3-vectors: (x x' + y y') + z z'
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def _normalize(a):  # v / norm, component by component
    n = math.sqrt(_dot(a, a))
    return [a[0] / n, a[1] / n, a[2] / n]


def _camera_basis(view):
    """CoordinateSystem::new, math/coordinate_system.rs:12-39."""
    w = [-c for c in _normalize(view)]
    up = [0.0, 1.0, 0.0]
    if abs(_dot(up, w)) > 0.999:
        up = list(w)
        x, y, z = abs(up[0]), abs(up[1]), abs(up[2])
        if x <= y and x <= z:
            up[0] = 1.0
        elif y <= x:
            up[1] = 1.0
        else:
            up[2] = 1.0
    u = _cross(up, w)
    v = _cross(w, u)  # from the un-normalised u
    return _normalize(u), _normalize(v), w


@pytest.mark.parametrize("cam", [
    {"position": "0 3.2 6.5", "viewDir": "0 -0.42 -1"},
    {"position": "7.5 6.5 13", "lookatPoint": "0 0 -4.5"},
    {"position": "0.3 9 0.1", "viewDir": "0 -1 0"},          # |up . w| > 0.999: x is the smallest component
    {"position": "0 9 0", "viewDir": "1e-4 -1 0"},           # ... z is
    {"position": "1 2 3", "viewDir": "0.02 0.9995 -0.01"},   # ... z again, looking up
    {"position": "-2.5 0.125 4", "lookatPoint": "3.3 0.7 -9.1"},
])
def test_camera_basis_bit_for_bit(rt, cam):
    entry = {"_name": "main", "_type": "perspective", "focalLength": 0.75}
    entry.update(cam)
    sc = rt.parse_scene(minimal_scene(camera=[entry]), "", image_width=640, image_height=360)
    pos = [float(x) for x in cam["position"].split()]
    if "viewDir" in cam:
        view = [float(x) for x in cam["viewDir"].split()]
    else:
        look = [float(x) for x in cam["lookatPoint"].split()]
        view = [look[k] - pos[k] for k in range(3)]  # scene/mod.rs:133
    u, v, w = _camera_basis(view)
    c = sc.c.camera
    assert [c.u.x, c.u.y, c.u.z] == u and [c.v.x, c.v.y, c.v.z] == v and [c.w.x, c.w.y, c.w.z] == w
    assert [c.position.x, c.position.y, c.position.z] == pos
    # CameraBase::new, camera/mod.rs:37-50 (aspect = W as Real / H as Real, scene/mod.rs:400)
    width = 0.5
    height = width / (640.0 / 360.0)
    assert (c.left, c.right, c.top, c.bottom) == (-width / 2.0, width / 2.0, height / 2.0, -height / 2.0)
    assert c.focal_length == 0.75 and (c.pixels_x, c.pixels_y) == (640, 360)


def _matmul(a, b):  # nalgebra's product: column j of the result is ((A[:,0] b0j + A[:,1] b1j) + A[:,2] b2j) + ...
    n = len(a)
    c = [[0.0] * n for _ in range(n)]
    for j in range(n):
        for i in range(n):
            acc = a[i][0] * b[0][j]
            for k in range(1, n):
                acc = acc + a[i][k] * b[k][j]
            c[i][j] = acc
    return c


def _axis_rotation(axis, degrees):
    """Rotation3::from_axis_angle(&V3::<axis>_axis(), PI * degrees / 180.0), scene/mod.rs:640-652: the axis-angle
    matrix with a unit basis vector put in for the axis."""
    angle = math.pi * degrees / 180.0
    ux, uy, uz = [1.0 if axis == k else 0.0 for k in "xyz"]
    s, c = math.sin(angle), math.cos(angle)
    omc = 1.0 - c
    sqx, sqy, sqz = ux * ux, uy * uy, uz * uz
    return [[sqx + (1.0 - sqx) * c, ux * uy * omc - uz * s, ux * uz * omc + uy * s],
            [ux * uy * omc + uz * s, sqy + (1.0 - sqy) * c, uy * uz * omc - ux * s],
            [ux * uz * omc - uy * s, uy * uz * omc + ux * s, sqz + (1.0 - sqz) * c]]


def _homogeneous(m3=None, t=None, s=None):
    m = [[1.0 if i == j else 0.0 for j in range(4)] for i in range(4)]
    if m3 is not None:
        for i in range(3):
            for j in range(3):
                m[i][j] = m3[i][j]
    if t is not None:
        for i in range(3):
            m[i][3] = t[i]
    if s is not None:
        for i in range(3):
            m[i][i] = s[i]
    return m


def test_instance_matrices_bit_for_bit(rt):
    """Instance::new (geometry/instance.rs:29-42) behind the xform accumulation of scene/mod.rs:626-654, and
    BBox::transform (geometry/bbox.rs:38-74): inverse transform, normal matrix and world box, bit for bit."""
    xf = [
        {"type": "translate", "amount": "1.25 -2.1 3.3"}, {"type": "rotate", "axis": "X", "amount": 90},
        {"type": "scale", "amount": "2 1.7 0.3"}, {"type": "rotation", "axis": "y", "degrees": 33.3},
        {"type": "rotate", "axis": "X", "amount": 20.5}, {"type": "translate", "amount": "0.1 0 1"},
        {"type": "scale", "amount": "1 0.5 1.1"}, {"type": "rotate", "axis": "Z", "amount": -47},
    ]
    lo, hi = [-1.0, -1.5, -0.25], [1.0, 2.0, 3.5]
    scene = minimal_scene(
        instance=[{"_name": "proto", "_type": "box", "shader": "m", "minPt": " ".join(map(str, lo)), "maxPt": " ".join(map(str, hi))}],
        shape=[{"_name": "i", "_type": "instance", "shader": "m", "_id": "proto", "xform": xf}],
    )
    sh = rt.parse_scene(scene, "").c.shapes[0]
    translate = [1.25 + 0.1, -2.1 + 0.0, 3.3 + 1.0]                     # translate += amount
    scale = [1.0 * 2 * 1, 1.0 * 1.7 * 0.5, 1.0 * 0.3 * 1.1]             # scale.component_mul_assign(amount)
    rx, ry, rz = _axis_rotation("x", 20.5), _axis_rotation("y", 33.3), _axis_rotation("z", -47)  # the later X wins
    rotation = _matmul(_matmul(rz, ry), rx)                              # rotate.2 * rotate.1 * rotate.0
    T, R, S = _homogeneous(t=translate), _homogeneous(m3=rotation), _homogeneous(s=scale)
    transform = _matmul(_matmul(T, R), S)
    inv_rotate = _homogeneous(m3=[[rotation[j][i] for j in range(3)] for i in range(3)])  # a rotation's inverse: transpose
    inv_scale = _homogeneous(s=[1.0 / c for c in scale])
    inv_translate = _homogeneous(t=[-c for c in translate])
    inv_scale_rot = _matmul(inv_scale, inv_rotate)
    inv_transform = _matmul(inv_scale_rot, inv_translate)
    normal_matrix = [[inv_scale_rot[j][i] for j in range(4)] for i in range(4)]
    assert list(sh.inv_transform) == [x for row in inv_transform for x in row]
    assert list(sh.normal_matrix) == [x for row in normal_matrix for x in row]
    # bbox.rs:38-74: the 8 corners through Matrix4::transform_point ((M3 p + t) / (0 p + m33)), then re-boxed
    pts = []
    for x in (lo[0], hi[0]):
        for y in (lo[1], hi[1]):
            for z in (lo[2], hi[2]):
                p = []
                for i in range(3):
                    acc = transform[i][0] * x
                    acc = acc + transform[i][1] * y
                    acc = acc + transform[i][2] * z
                    p.append(acc + transform[i][3])
                n = transform[3][0] * x + transform[3][1] * y + transform[3][2] * z + transform[3][3]
                pts.append([c / n for c in p])
    assert [sh.bbox_min.x, sh.bbox_min.y, sh.bbox_min.z] == [min(p[k] for p in pts) for k in range(3)]
    assert [sh.bbox_max.x, sh.bbox_max.y, sh.bbox_max.z] == [max(p[k] for p in pts) for k in range(3)]
