"""CPU rehearsal of the GPU parity tests: the backend's stage bodies (the same headers the
HIP kernels are compiled from, built for the host by tests/hostsim) against the oracle.
This proves the fp32-traversal / f64-refine split and the queue protocol without a GPU; the
`gpu`-marked tests repeat it through the C ABI on the device."""
import numpy as np
import pytest

import hostsim_lib as HS
import oracle_lib as O
import parity

SCENES = ["simple_sphere_scene", "spheres_and_triangles", "cornell_room", "spheres_1K", "bunny_small", "mixed_features",
          "mixed_ortho"]
W, H, N = 160, 120, 3


@pytest.fixture(scope="module")
def oracle_cache(rt, scene_path):
    cache = {}

    def get(name, **kw):
        key = (name, tuple(sorted(kw.items())))
        if key not in cache:
            sc = rt.parse_scene_file(scene_path(name), image_width=W, image_height=H, recursion_depth=6, **kw)
            cache[key] = (sc,) + O.render(sc, N)
        return cache[key]

    return get


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("mode", ["fast", "reference_order"])
def test_stage_bodies_match_oracle(rt, oracle_cache, name, mode):
    sc, fb_o, ids_o, cnt = oracle_cache(name)
    m = rt.RT_MODE_FAST if mode == "fast" else rt.RT_MODE_REFERENCE_ORDER
    fb, ids, stats = HS.render(sc, N, mode=m)
    parity.assert_parity(fb, ids, fb_o, ids_o, occlusion_only=(mode == "fast"))
    assert stats.primary_rays == cnt["rays"][0]
    # secondary / shadow counts may differ by the handful of rays whose hit differs
    assert abs(int(stats.secondary_rays) - cnt["rays"][1]) <= max(4, cnt["rays"][1] // 5000)
    assert abs(int(stats.shadow_rays) - cnt["rays"][2]) <= max(8, cnt["rays"][2] // 5000)


@pytest.mark.parametrize("aa", [1, 2])
@pytest.mark.parametrize("name", ["mixed_features", "bunny_small"])
def test_stochastic_antialiasing_matches_oracle(rt, scene_path, name, aa):
    """Jittered / random sample offsets (antialias.rs:45-57): both sides draw the same numbers, so the
    frames compare sample by sample like the regular grid's."""
    sc = rt.parse_scene_file(scene_path(name), image_width=96, image_height=72, recursion_depth=6)
    fb_o, ids_o, cnt = O.render(sc, 3, aa=aa, seed=0x1234567890ABCDEF)
    fb, ids, stats = HS.render(sc, 3, mode=rt.RT_MODE_FAST, aa=aa, seed=0x1234567890ABCDEF)
    parity.assert_parity(fb, ids, fb_o, ids_o, occlusion_only=True)
    assert stats.primary_rays == cnt["rays"][0]
    fb2, _, _ = HS.render(sc, 3, mode=rt.RT_MODE_FAST, aa=aa, seed=99, want_ids=False)
    assert not np.array_equal(fb, fb2)
    # the frame does not depend on how the image is tiled or split
    whole = np.zeros_like(fb)
    for p in range(2):
        part, _, _ = HS.render(sc, 3, mode=rt.RT_MODE_FAST, aa=aa, seed=0x1234567890ABCDEF, want_ids=False, tile=(16, 8),
                               part=(p, 2))
        whole += part
    assert np.array_equal(whole, fb)


@pytest.mark.parametrize("name", ["mixed_features", "cornell_room"])
def test_shadows_disabled_and_normals(rt, scene_path, name):
    for kw in ({"disable_shadows": True}, {"render_normals": True}):
        sc = rt.parse_scene_file(scene_path(name), image_width=96, image_height=72, recursion_depth=6, **kw)
        fb_o, ids_o, cnt = O.render(sc, 2)
        fb, ids, stats = HS.render(sc, 2, mode=rt.RT_MODE_FAST)
        parity.assert_parity(fb, ids, fb_o, ids_o, occlusion_only=True)
        if kw.get("disable_shadows"):
            assert stats.shadow_rays == 0 and cnt["rays"][2] == 0


def test_recursion_depth_limits_mirror_bounces(rt, scene_path):
    for depth in (0, 1, 3):
        sc = rt.parse_scene_file(scene_path("mixed_features"), image_width=96, image_height=72, recursion_depth=depth)
        fb_o, ids_o, _ = O.render(sc, 2)
        fb, ids, _ = HS.render(sc, 2, mode=rt.RT_MODE_REFERENCE_ORDER)
        parity.assert_parity(fb, ids, fb_o, ids_o, occlusion_only=False)
        assert ids.bounces.max() <= depth


def test_fast_mesh_trees_change_no_hit(rt, scene_path):
    """The fast mode's surface-area mesh trees against the reference's trees walked by the same fast
    code (RT_UPLOAD_REFERENCE_MESH_TREES at scene flattening): the same closest hits -- a tree only
    culls -- except where two triangles tie in fp32 along a shared edge, and the same image."""
    sc = rt.parse_scene_file(scene_path("bunny_small"), image_width=160, image_height=120, recursion_depth=4)
    fb_a, ids_a, st_a = HS.render(sc, 2, mode=rt.RT_MODE_FAST)
    fb_b, ids_b, st_b = HS.render(sc, 2, mode=rt.RT_MODE_FAST, upload_flags=rt.RT_UPLOAD_REFERENCE_MESH_TREES)
    same = (ids_a.primary_shape == ids_b.primary_shape) & (ids_a.primary_tri == ids_b.primary_tri)
    assert same.mean() >= 0.9999, same.mean()
    assert np.array_equal(ids_a.shadow_shape >= 0, ids_b.shadow_shape >= 0) or \
        ((ids_a.shadow_shape >= 0) == (ids_b.shadow_shape >= 0)).mean() >= 0.9999
    assert parity.image_metrics(fb_a, fb_b)["max_abs_diff"] <= 1
    assert st_a.total_rays() == st_b.total_rays()


@pytest.mark.parametrize("parts,weight", [(2, 0), (3, 0), (3, 5)])
def test_partitioned_render_tiles_the_image(rt, scene_path, parts, weight):
    """Each part renders only the tiles it owns; the parts are disjoint, cover the image, and their
    pixels are bit-identical with the unpartitioned frame -- also when part 0 takes a smaller share."""
    sc = rt.parse_scene_file(scene_path("mixed_features"), image_width=100, image_height=70, recursion_depth=6)
    tile = (16, 16)
    full, _, _ = HS.render(sc, 2, want_ids=False, tile=tile)
    tiles_x, tiles_y = (100 + 15) // 16, (70 + 15) // 16
    seen = np.zeros(tiles_x * tiles_y, np.int32)
    merged = np.zeros_like(full)
    for p in range(parts):
        a = rt.RtRenderArgs()
        a.sqrt_rays_per_pixel, a.tile_w, a.tile_h, a.part_index, a.part_count, a.part0_weight = 2, 16, 16, p, parts, weight
        owned = rt.partition_tiles(100, 70, a)  # the shipped library's ownership rule
        seen[owned] += 1
        fb, _, _ = HS.render(sc, 2, want_ids=False, tile=tile, part=(p, parts), part0_weight=weight)
        mask = np.zeros((70, 100), bool)
        for k in owned:
            ty, tx = divmod(int(k), tiles_x)
            mask[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
        assert np.array_equal(fb[mask], full[mask])
        assert not fb[~mask].any()
        merged += fb
    assert (seen == 1).all()
    assert np.array_equal(merged, full)


def test_trace_rays_matches_oracle(rt, scene_path):
    """BVH::closest_hit on arbitrary rays (not only camera rays): random origins on a shell around the scene."""
    sc = rt.parse_scene_file(scene_path("mixed_features"))
    rng = np.random.default_rng(5)
    n = 20000
    o = rng.normal(size=(n, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * 12.0
    d = rng.normal(size=(n, 3)) * 1.5 - o
    t_o, s_o, tri_o, _ = O.trace_rays(sc, o, d, 1e-6, np.inf)
    for mode in (rt.RT_MODE_FAST, rt.RT_MODE_REFERENCE_ORDER):
        t, s, tri = HS.trace_rays(sc, o, d, 1e-6, np.inf, mode=mode)
        agree = np.mean((s == s_o) & (tri == tri_o))
        assert agree >= 0.9999, agree
        both = (s == s_o) & (s >= 0)
        assert np.allclose(t[both], t_o[both], rtol=2e-5)


import edge_scenes


@pytest.mark.parametrize("name", edge_scenes.NAMES)
def test_corner_case_scenes(rt, tmp_path, name):
    """No lights, ambient only, one pixel, odd sizes, camera inside a box, facing mirrors, a mesh with
    a zero-area triangle, non-uniformly scaled instances."""
    text, data_dir, w, h, n = edge_scenes.scenes(str(tmp_path))[name]
    sc = rt.parse_scene(text, data_dir, image_width=w, image_height=h, recursion_depth=5)
    fb_o, ids_o, cnt = O.render(sc, n)
    for mode in (rt.RT_MODE_FAST, rt.RT_MODE_REFERENCE_ORDER):
        fb, ids, stats = HS.render(sc, n, mode=mode)
        # tiny frames: compare exactly (a single differing sample would already break the fraction bars)
        assert np.array_equal(ids.primary_shape, ids_o.primary_shape) and np.array_equal(ids.primary_tri, ids_o.primary_tri)
        assert np.array_equal(ids.final_shape, ids_o.final_shape) and np.array_equal(ids.bounces, ids_o.bounces)
        assert np.array_equal(ids.shadow_shape >= 0, ids_o.shadow_shape >= 0)
        assert parity.image_metrics(fb, fb_o)["max_abs_diff"] <= 1
        assert stats.total_rays() == sum(cnt["rays"])


# ---- GGXMirror (shader/ggx_mirror.rs): the ray-tree path

def _ggx_check(rt, fb, ids, stats, fb_o, ids_o, cnt):
    m = parity.image_metrics(fb, fb_o)
    assert m["lsb1_fraction"] >= parity.LSB_MATCH_MIN and m["psnr_db"] >= parity.PSNR_MIN_DB, m
    assert (ids.primary_shape == ids_o.primary_shape).mean() >= 0.9999
    got = [int(stats.primary_rays), int(stats.secondary_rays), int(stats.shadow_rays)]
    for g, r in zip(got, cnt["rays"]):
        assert abs(g - r) <= max(4, r // 2000), (got, cnt["rays"])


@pytest.mark.parametrize("mode", ["fast", "reference_order"])
def test_ggx_ray_trees_match_oracle(rt, mode):
    """Every GGX hit spawns `samples` reflected rays, recursively; both sides draw the same numbers in
    the order of the reference's recursion, so frames compare like deterministic ones."""
    sc = rt.parse_scene(edge_scenes.ggx_scene(), "", image_width=96, image_height=72, recursion_depth=3)
    fb_o, ids_o, cnt = O.render(sc, 2, seed=77)
    assert cnt["rays"][1] > 3 * 96 * 72  # the trees do fan out
    m = rt.RT_MODE_FAST if mode == "fast" else rt.RT_MODE_REFERENCE_ORDER
    fb, ids, stats = HS.render(sc, 2, mode=m, seed=77)
    _ggx_check(rt, fb, ids, stats, fb_o, ids_o, cnt)
    other, _, _ = HS.render(sc, 2, mode=m, seed=78, want_ids=False)
    assert not np.array_equal(other, fb)
    whole = np.zeros_like(fb)
    for p in range(2):
        part, _, _ = HS.render(sc, 2, mode=m, seed=77, want_ids=False, tile=(16, 16), part=(p, 2))
        whole += part
    assert np.array_equal(whole, fb)


def test_ggx_limits(rt):
    """Roughness 0 with one sample is a perfect mirror (the microfacet normal is the surface normal);
    recursion depth 0 shows the background on every rough surface; jittered offsets and GGX draws
    come from separate streams."""
    smooth = rt.parse_scene(edge_scenes.ggx_scene(roughness=0.0, samples=1), "", image_width=80, image_height=60, recursion_depth=4)
    mirror = rt.parse_scene(edge_scenes.ggx_scene(mirror_type="Mirror"), "", image_width=80, image_height=60, recursion_depth=4)
    a, _, _ = HS.render(smooth, 2, want_ids=False)
    b, _, _ = HS.render(mirror, 2, want_ids=False)
    assert parity.image_metrics(a, b)["lsb1_fraction"] >= 0.995
    flat = rt.parse_scene(edge_scenes.ggx_scene(), "", image_width=80, image_height=60, recursion_depth=0)
    fb, ids, stats = HS.render(flat, 1)
    fb_o, ids_o, cnt = O.render(flat, 1)
    assert stats.secondary_rays == 0 == cnt["rays"][1]
    assert parity.image_metrics(fb, fb_o)["max_abs_diff"] <= 1
    sc = rt.parse_scene(edge_scenes.ggx_scene(), "", image_width=64, image_height=48, recursion_depth=2)
    fb_o, ids_o, cnt = O.render(sc, 2, aa=1, seed=5)
    fb, ids, stats = HS.render(sc, 2, aa=1, seed=5)
    _ggx_check(rt, fb, ids, stats, fb_o, ids_o, cnt)
