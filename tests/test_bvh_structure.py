"""The shipped BVH builder (csrc/bvh_build.cpp) must produce the reference's tree
(raytracer-lib/src/geometry/bvh.rs:21-138): same pre-order node sequence, same leaf
contents in the same order, same f64 boxes.  The oracle builds its tree independently,
recursively, the way BVH::new does; the product builds into flat arrays."""
import numpy as np
import pytest

import hostsim_lib as HS
import oracle_lib as O

SCENES = ["simple_sphere_scene", "spheres_and_triangles", "cornell_room", "spheres_1K", "bunny_small", "mixed_features",
          "mixed_ortho"]


@pytest.mark.parametrize("name", SCENES)
def test_scene_level_tree_matches(rt, scene_path, name):
    sc = rt.parse_scene_file(scene_path(name))
    ko, po, bo = O.bvh_dump(sc, -1)
    kp, pp, bp, depth = HS.bvh_dump(sc, -1)
    assert np.array_equal(ko, kp)
    assert np.array_equal(po, pp)
    assert np.array_equal(bo, bp)
    assert sorted(po.tolist()) == list(range(sc.c.n_shapes))  # every shape exactly once
    assert (ko[ko > 0] <= 4).all()  # BVH leaf size (bvh.rs:46)


@pytest.mark.parametrize("name", ["bunny_small", "bunny_medium", "mixed_features"])
def test_mesh_tree_matches(rt, scene_path, name):
    sc = rt.parse_scene_file(scene_path(name))
    for m in range(sc.c.n_meshes):
        ko, po, bo = O.bvh_dump(sc, m)
        kp, pp, bp, depth = HS.bvh_dump(sc, m)
        assert np.array_equal(ko, kp) and np.array_equal(po, pp) and np.array_equal(bo, bp)
        assert sorted(po.tolist()) == list(range(sc.c.meshes[m].n_triangles))


def test_split_axis_cycles_and_sort_is_stable(rt):
    """Eight unit spheres whose centroids tie on y and z: the x split must keep file order among
    equal keys (Vec::sort_by is stable, bvh.rs:75-82), and depth d splits on axis d % 3."""
    import json

    shapes = [{"_name": f"s{i}", "_type": "sphere", "shader": "m", "center": f"{i % 2} 0 0", "radius": 0.25} for i in range(8)]
    scene = json.dumps({"scene": {
        "camera": [{"_name": "main", "_type": "perspective", "position": "0 0 5", "viewDir": "0 0 -1", "focalLength": 1.0}],
        "shader": [{"_name": "m", "_type": "Lambertian", "diffuse": "1 1 1"}], "shape": shapes}})
    sc = rt.parse_scene(scene, "")
    ko, po, bo = O.bvh_dump(sc, -1)
    kp, pp, bp, _ = HS.bvh_dump(sc, -1)
    assert np.array_equal(ko, kp) and np.array_equal(po, pp) and np.array_equal(bo, bp)
    # root splits on x: the four x = 0 spheres (even file indices, in file order) go left
    assert po.tolist() == [0, 2, 4, 6, 1, 3, 5, 7]
    assert ko.tolist() == [-1, 4, 4]


def test_stack_depth_bounds_the_walk(rt, scene_path):
    """stack_depth must cover the deepest scene-level path plus the deepest mesh path."""
    for name in ["bunny_small", "spheres_1K", "mixed_features"]:
        sc = rt.parse_scene_file(scene_path(name))

        def depth_of(kinds):
            # pre-order walk: inner = -1 has two children
            best, stack, d = 0, [], 0
            it = iter(kinds.tolist())

            def rec(level):
                nonlocal best
                k = next(it)
                best = max(best, level)
                if k < 0:
                    rec(level + 1)
                    rec(level + 1)

            rec(0)
            return best

        import sys

        sys.setrecursionlimit(10000)
        k_top, _, _, stack_depth = HS.bvh_dump(sc, -1)
        k_fast, _, _, stack_depth_fast = HS.bvh_dump(sc, -2)  # the fast mode's own scene-level tree
        mesh = max([depth_of(HS.bvh_dump(sc, m)[0]) for m in range(sc.c.n_meshes)] or [0])
        assert stack_depth >= depth_of(k_top) + mesh + 1
        assert stack_depth <= depth_of(k_top) + mesh + 6  # and is not wastefully large (LDS stack rows)
        # the fast mode walks its own scene-level tree and, per mesh, its own tree if it has one
        need = max(depth_of(k_top), depth_of(k_fast))
        fast = [HS.bvh_dump(sc, -3 - m)[0] for m in range(sc.c.n_meshes)]
        mesh_fast = max([depth_of(k) for k in fast if len(k)] + [mesh])
        assert stack_depth_fast >= need + mesh_fast + 1
        assert stack_depth_fast <= need + mesh_fast + 6


def test_fast_mesh_trees_hold_the_reachable_triangles_once(rt, scene_path, tmp_path):
    """The fast mode's mesh trees are split by surface area; they must hold exactly the triangles the
    reference's tree can reach (those of its leaves whose box is not flat, bbox.rs:89,104,119), each
    once, in leaves of at most four, every box inside its parent's, and the leaf boxes must contain
    their triangles."""
    import sys

    import edge_scenes

    sys.setrecursionlimit(10000)
    text, data_dir, _, _, _ = edge_scenes.scenes(str(tmp_path))["degenerate_mesh"]
    for sc in [rt.parse_scene_file(scene_path("bunny_small")), rt.parse_scene_file(scene_path("mixed_features")),
               rt.parse_scene(text, data_dir)]:
        assert sc.c.n_meshes > 0
        for m in range(sc.c.n_meshes):
            kinds, prims, boxes, _ = HS.bvh_dump(sc, -3 - m)
            n_tri = int(sc.c.meshes[m].n_triangles)
            assert len(kinds) > 0, "these meshes are expected to get a tree of their own"
            k_ref, p_ref, b_ref, _ = HS.bvh_dump(sc, m)
            reachable, pos = [], 0
            for k, b in zip(k_ref.tolist(), b_ref):
                if k < 0:
                    continue
                if (b[3:] > b[:3]).all():
                    reachable += p_ref[pos:pos + k].tolist()
                pos += k
            assert sorted(prims.tolist()) == sorted(reachable)
            assert max(k for k in kinds.tolist() if k > 0) <= 4
            verts = np.ctypeslib.as_array(sc.c.meshes[m].vertices, shape=(n_tri, 3, 3))
            at = [0]

            def check(i):
                if kinds[i] >= 0:
                    tri = verts[prims[at[0]:at[0] + kinds[i]]].reshape(-1, 3)
                    at[0] += kinds[i]
                    assert (tri >= boxes[i][:3]).all() and (tri <= boxes[i][3:]).all()
                    return i + 1
                l = i + 1
                r = check(l)
                end = check(r)
                for c in (l, r):
                    assert (boxes[c][:3] >= boxes[i][:3]).all() and (boxes[c][3:] <= boxes[i][3:]).all()
                return end

            assert check(0) == len(kinds)


def test_fast_scene_tree_holds_the_reachable_shapes(rt, scene_path, tmp_path):
    """The fast mode's scene-level tree is split by surface area, not by the reference's median rule, so
    it must hold exactly the shapes the reference's tree can reach: all of them, except those in a
    reference leaf whose box has zero thickness (never entered, bbox.rs:89,104,119)."""
    for name in ["spheres_1K", "mixed_features", "cornell_room", "spheres_and_triangles"]:
        sc = rt.parse_scene_file(scene_path(name))
        kinds, prims, boxes, _ = HS.bvh_dump(sc, -1)
        reachable, at = [], 0
        for k, b in zip(kinds.tolist(), boxes):
            if k < 0:
                continue
            if (b[3:] > b[:3]).all():
                reachable += prims[at:at + k].tolist()
            at += k
        k_fast, p_fast, b_fast, _ = HS.bvh_dump(sc, -2)
        assert sorted(p_fast.tolist()) == sorted(reachable), name
        assert max(k for k in k_fast.tolist() if k > 0) <= 2
        # every node's box contains its children's boxes (pre-order: left child follows its parent)
        def check(i):
            if k_fast[i] >= 0:
                return i + 1
            l = i + 1
            r = check(l)
            end = check(r)
            for c in (l, r):
                assert (b_fast[c][:3] >= b_fast[i][:3]).all() and (b_fast[c][3:] <= b_fast[i][3:]).all()
            return end
        assert check(0) == len(k_fast)


def test_reference_bvh_construction_case(rt):
    """geometry/bvh.rs:156-182 asserts that the root box of that two-sphere tree is hit by the ray from the
    origin along -z over [0, inf).  Here: both builders give one leaf of two spheres whose box is
    [-1, 3] x [-1, 1] x [-6, -4] (BBox::combine of the spheres' boxes, sphere.rs:91-96), the oracle's
    BBox::hit accepts that ray, and the walk (oracle and host build of the device code) finds sphere1 at
    t = 4."""
    import edge_scenes

    sc = rt.parse_scene(edge_scenes.two_sphere_scene_of_the_reference_test(), "")
    ko, po, bo = O.bvh_dump(sc, -1)
    kp, pp, bp, _ = HS.bvh_dump(sc, -1)
    assert np.array_equal(ko, kp) and np.array_equal(po, pp) and np.array_equal(bo, bp)
    assert ko.tolist() == [2] and po.tolist() == [0, 1]
    assert bo.reshape(-1, 6)[0].tolist() == [-1.0, -1.0, -6.0, 3.0, 1.0, -4.0]
    assert O.bbox_hit((-1, -1, -6), (3, 1, -4), (0, 0, 0), (0, 0, -1), 0.0, float("inf")) == 4.0  # is_some(), entry t
    for mode in (rt.RT_MODE_FAST, rt.RT_MODE_REFERENCE_ORDER):
        t, shape, tri = HS.trace_rays(sc, [(0, 0, 0)], [(0, 0, -1)], 0.0, float("inf"), mode=mode)
        assert (float(t[0]), int(shape[0])) == (4.0, 0)
    t, shape, tri, normal = O.trace_rays(sc, [(0, 0, 0)], [(0, 0, -1)], 0.0, float("inf"))
    assert (float(t[0]), int(shape[0]), normal[0].tolist()) == (4.0, 0, [0.0, 0.0, 1.0])
