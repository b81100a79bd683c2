"""Known answers pushed through the DEVICE code, not only through the oracle.

The only executable test the reference holds for this path is
raytracer-lib/src/geometry/bbox.rs:128-173 (five ray / box cases, tests/golden/bbox_vectors.json).
Here each of them runs through librt_hip.so on one-box scenes, in both traversal modes and
through both kinds of kernels: rt_trace_rays (one ray per lane: the node test of the scene-level
tree, then Cuboid::closest_hit = BBox::hit on the box itself) and a 1 x 1 pixel render whose only
camera ray is the case's ray (the wave-packet kernels in fast mode).  The hand-derived sphere /
triangle / cuboid cases of tests/test_oracle_golden.py follow the same way.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
INF = float("inf")


def v3(a):
    return " ".join(repr(float(x)) for x in a)


def one_shape_scene(shape, eye=(0, 0, 0), view=(0, 0, -1), ballast=False):
    """A scene that holds `shape` alone, seen by a camera whose centre ray is exactly eye + t * view
    (perspective.rs:20-28: direction = -W * focal at u = v = 0, and W = -normalize(view)).
    `ballast` adds a small sphere far away (shape 1): the leaf that holds an axis-aligned triangle
    alone has a box without thickness, which the reference never enters (bbox.rs:119) -- such a
    triangle is invisible there, and here."""
    shape = dict(shape, _name="only", shader={"_ref": "s"})
    shapes = [shape]
    if ballast:
        shapes.append({"_name": "far", "_type": "sphere", "shader": {"_ref": "s"}, "center": "100 100 100", "radius": 0.5})
    return json.dumps({"scene": {
        "camera": [{"_name": "main", "_type": "perspective", "position": v3(eye), "viewDir": v3(view),
                    "focalLength": float(np.linalg.norm(np.asarray(view, np.float64))), "imagePlaneWidth": 0.5}],
        "shader": [{"_name": "s", "_type": "Lambertian", "diffuse": "1 1 1"}],
        "shape": shapes,
    }})


def box(mn, mx):
    return {"_type": "box", "minPt": v3(mn), "maxPt": v3(mx)}


def trace_one(rt, backend, text, o, d, tmin, tmax, mode):
    scene = rt.parse_scene(text, "", image_width=1, image_height=1, recursion_depth=0)
    dev = backend.upload(scene)
    t, shape, tri = dev.trace_rays([o], [d], tmin, tmax, mode=mode)
    dev.close()
    return float(t[0]), int(shape[0])


@pytest.mark.parametrize("mode", [0, 1])
def test_reference_bbox_vectors_on_the_device(rt, backend, mode):
    """geometry/bbox.rs:135-171: three hits, two misses, tmin = 1.0, tmax = inf."""
    with open(os.path.join(GOLDEN, "bbox_vectors.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == 5
    for c in cases:
        text = one_shape_scene(box(c["min"], c["max"]), eye=c["origin"], view=c["direction"])
        # (1) caller's ray through the per-lane traversal
        t, shape = trace_one(rt, backend, text, c["origin"], c["direction"], c["tmin"], INF, mode)
        assert (shape == 0) == c["hit"], (c, t, shape)
        if not c["hit"]:
            assert np.isinf(t)  # a miss leaves t at tmax
        # (2) the same ray as the only camera ray of a 1 x 1 frame: Hit::new gives it t_min = 1.0
        # (hit_struct.rs:20), the case's tmin; fast mode walks it as a wave packet
        scene = rt.parse_scene(text, "", image_width=1, image_height=1, recursion_depth=0)
        dev = backend.upload(scene)
        fb, ids, stats = dev.render(1, mode=mode, want_ids=True)
        assert (int(ids.primary_shape[0]) == 0) == c["hit"], (c, ids.primary_shape)
        assert stats.primary_rays == 1
        if not c["hit"]:
            assert fb.pixels[0, 0].tolist() == pytest.approx([0.198] * 3)  # DEFAULT_BACKGROUND_COLOR, prelude.rs:41
        dev.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_bbox_entry_values_on_the_device(rt, backend, mode):
    unit = box([-0.5] * 3, [0.5] * 3)
    # entry at t = 4.5 (bbox.rs:124: max(tn_z, tmin))
    t, shape = trace_one(rt, backend, one_shape_scene(unit), [0, 0, 5], [0, 0, -1], 1.0, INF, mode)
    assert shape == 0 and t == 4.5
    # origin inside the box: the reference reports t_min, not the exit point
    t, shape = trace_one(rt, backend, one_shape_scene(box([-1] * 3, [1] * 3)), [0, 0, 0], [0.3, 0.2, -1], 0.25, INF, mode)
    assert shape == 0 and t == 0.25
    # a flat box is never hit (bbox.rs:119)
    t, shape = trace_one(rt, backend, one_shape_scene(box([-1, -1, -3], [1, 1, -3])), [0, 0, 0], [0, 0, -1], 1e-6, INF, mode)
    assert shape == -1
    # half-open upper bound: entry exactly at tmax is a miss
    t, shape = trace_one(rt, backend, one_shape_scene(unit), [0, 0, 5], [0, 0, -1], 1.0, 4.5, mode)
    assert shape == -1
    # 0/0 = NaN compares false: a ray lying in the plane x = min.x is not rejected by the x slab
    t, shape = trace_one(rt, backend, one_shape_scene(box([0, -1, -1], [1, 1, 1])), [0, 0, 5], [0, 0, -1], 1.0, INF, mode)
    assert shape == 0 and t == 4.0


@pytest.mark.parametrize("mode", [0, 1])
def test_sphere_cases_on_the_device(rt, backend, mode):
    """sphere.rs:58-89 (values derived by hand, as in tests/test_oracle_golden.py)."""
    text = one_shape_scene({"_type": "sphere", "center": "0 0 0", "radius": 1.0})
    o, d = [0, 0, 5], [0, 0, -1]
    assert trace_one(rt, backend, text, o, d, 1.0, INF, mode) == (4.0, 0)
    assert trace_one(rt, backend, text, o, d, 4.5, INF, mode) == (6.0, 0)  # near root below t_min: the far one
    t, shape = trace_one(rt, backend, text, o, d, 1.0, 4.0, mode)         # half-open: a root equal to hit.t is no hit
    assert shape == -1 and t == 4.0
    assert trace_one(rt, backend, text, o, [0, 0, -2], 1.0, INF, mode) == (2.0, 0)  # t in units of |d|
    assert trace_one(rt, backend, text, [0, 2, 5], d, 1.0, INF, mode)[1] == -1


@pytest.mark.parametrize("mode", [0, 1])
def test_triangle_cases_on_the_device(rt, backend, mode):
    """triangle.rs:66-118."""
    tri = {"_type": "triangle", "v0": "0 0 0", "v1": "1 0 0", "v2": "0 1 0"}
    text = one_shape_scene(tri, ballast=True)
    down = [0, 0, -1]
    # alone in its leaf, the triangle in the plane z = 0 cannot be seen at all (flat leaf box)
    assert trace_one(rt, backend, one_shape_scene(tri), [0.25, 0.25, 2], down, 1.0, INF, mode)[1] == -1
    assert trace_one(rt, backend, text, [0.25, 0.25, 2], down, 1.0, INF, mode) == (2.0, 0)
    assert trace_one(rt, backend, text, [0.25, 0.25, -2], [0, 0, 1], 1.0, INF, mode) == (2.0, 0)  # from behind
    assert trace_one(rt, backend, text, [0.25, 0.25, 2], down, 1.0, 2.0, mode) == (2.0, 0)  # closed interval: t == hit.t replaces
    assert trace_one(rt, backend, text, [-1, 0.25, 0], [1, 0, 0], 1e-6, INF, mode)[1] == -1  # edge-on: |det A| < epsilon
    assert trace_one(rt, backend, text, [0.75, 0.75, 2], down, 1.0, INF, mode)[1] == -1      # beta > 1 - gamma
    assert trace_one(rt, backend, text, [0.5, 0.0, 2], down, 1.0, INF, mode)[1] == 0         # the edge gamma = 0 is inside
    assert trace_one(rt, backend, text, [0.0, 0.5, 2], down, 1.0, INF, mode)[1] == 0         # the edge beta = 0 is inside


@pytest.mark.parametrize("mode", [0, 1])
def test_cuboid_cases_on_the_device(rt, backend, mode):
    """cuboid.rs:63-73 = BBox::hit on the box's own extent."""
    text = one_shape_scene(box([-1, -2, -3], [1, 2, 3]))
    assert trace_one(rt, backend, text, [0, 0, 10], [0, 0, -1], 1.0, INF, mode) == (7.0, 0)
    assert trace_one(rt, backend, text, [5, 0.5, 0.2], [-1, 0, 0], 1.0, INF, mode) == (4.0, 0)
    assert trace_one(rt, backend, text, [0.2, -9, 0.3], [0, 1, 0], 1.0, INF, mode) == (7.0, 0)
    t, shape = trace_one(rt, backend, text, [0, 0, 0], [0, 0, -1], 0.5, INF, mode)  # from inside: t_min
    assert shape == 0 and t == 0.5


@pytest.mark.parametrize("mode", [0, 1])
def test_reference_bvh_construction_case_on_the_device(rt, backend, mode):
    """geometry/bvh.rs:149-183 (a test the reference keeps commented out): two unit spheres at (0, 0, -5)
    and (2, 0, -5), a ray from the origin along -z over [0, inf) must hit the tree's root box.  On the
    device the walk must go on to find sphere1 at t = 4, through rt_trace_rays and as the only camera ray
    of a 1 x 1 frame (t_min = 1, the same answer)."""
    from edge_scenes import two_sphere_scene_of_the_reference_test

    scene = rt.parse_scene(two_sphere_scene_of_the_reference_test(), "", image_width=1, image_height=1, recursion_depth=0)
    dev = backend.upload(scene)
    t, shape, tri = dev.trace_rays([(0, 0, 0)], [(0, 0, -1)], 0.0, INF, mode=mode)
    assert (float(t[0]), int(shape[0]), int(tri[0])) == (4.0, 0, -1)
    # from a point between the spheres' boxes and the second sphere: only sphere2
    t, shape, tri = dev.trace_rays([(2, 0, 0)], [(0, 0, -1)], 0.0, INF, mode=mode)
    assert (float(t[0]), int(shape[0])) == (4.0, 1)
    fb, ids, stats = dev.render(1, mode=mode, want_ids=True)
    assert int(ids.primary_shape[0]) == 0 and stats.primary_rays == 1
    dev.close()
