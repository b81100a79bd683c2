"""Pins the CPU oracle: the reference's own known-answer test, hand-derived single-ray
cases, and what its published renders tell about constants.

The reference holds exactly one executable test for this path
(raytracer-lib/src/geometry/bbox.rs:128-173); its five vectors are transcribed in
tests/golden/bbox_vectors.json.  Everything else below is derived independently with
numpy float64 arithmetic, not by running the reference (no Rust toolchain exists here).
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import parity

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
INF = float("inf")


def test_reference_bbox_vectors():
    """geometry/bbox.rs:135-171: three hits, two misses, tmin = 1.0, tmax = inf."""
    with open(os.path.join(GOLDEN, "bbox_vectors.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == 5
    for c in cases:
        t = O.bbox_hit(c["min"], c["max"], c["origin"], c["direction"], c["tmin"], INF)
        assert (t is not None) == c["hit"], c


def test_bbox_entry_value_and_inside_origin():
    # unit box around the origin, ray from z = 5 looking down -z: entry at t = 4.5
    t = O.bbox_hit([-0.5] * 3, [0.5] * 3, [0, 0, 5], [0, 0, -1], 1.0, INF)
    assert t == 4.5
    # origin inside the box: the reference returns max(tn_z, tmin) = tmin, not the exit point (bbox.rs:124)
    t = O.bbox_hit([-1] * 3, [1] * 3, [0, 0, 0], [0.3, 0.2, -1], 1e-6, INF)
    assert t == 1e-6
    # a flat box is never hit: tmin_z >= tmax_z (bbox.rs:119)
    assert O.bbox_hit([-1, -1, -3], [1, 1, -3], [0, 0, 0], [0, 0, -1], 1e-6, INF) is None
    # half-open upper bound: entry exactly at tmax is a miss (tmin_k >= tmax)
    assert O.bbox_hit([-0.5] * 3, [0.5] * 3, [0, 0, 5], [0, 0, -1], 1.0, 4.5) is None
    # 0/0 = NaN compares false everywhere: a ray in the plane x = min.x with dir.x = 0 is not rejected by x
    t = O.bbox_hit([0, -1, -1], [1, 1, 1], [0, 0, 5], [0, 0, -1], 1.0, INF)
    assert t == 4.0


def test_sphere_cases():
    c, o, d = [0, 0, 0], [0, 0, 5], [0, 0, -1]
    hit, t, n = O.primitive_hit(0, c, c, c, 1.0, o, d, 1.0, INF)
    assert hit and t == 4.0 and np.allclose(n, [0, 0, 1])
    # t1 below t_min: the far root is taken (sphere.rs:78-81)
    hit, t, n = O.primitive_hit(0, c, c, c, 1.0, o, d, 4.5, INF)
    assert hit and t == 6.0 and np.allclose(n, [0, 0, -1])
    # half-open range: a root equal to hit.t is not accepted (sphere.rs:74)
    hit, t, _ = O.primitive_hit(0, c, c, c, 1.0, o, d, 1.0, 4.0)
    assert not hit and t == 4.0
    # un-normalised direction: t scales inversely (roots are in units of |d|)
    hit, t, _ = O.primitive_hit(0, c, c, c, 1.0, o, [0, 0, -2], 1.0, INF)
    assert hit and t == 2.0
    # miss
    hit, _, _ = O.primitive_hit(0, c, c, c, 1.0, [0, 2, 5], d, 1.0, INF)
    assert not hit
    # generic case against numpy
    rng = np.random.default_rng(7)
    for _ in range(50):
        c = rng.normal(size=3)
        r = rng.uniform(0.2, 2.0)
        o = rng.normal(size=3) * 5
        d = (c - o) + rng.normal(size=3) * 0.3
        oc = o - c
        disc = np.dot(oc, d) ** 2 - np.dot(d, d) * (np.dot(oc, oc) - r * r)
        hit, t, n = O.primitive_hit(0, c, c, c, r, o, d, 1e-6, INF)
        assert hit == (disc >= 0 and (-np.dot(oc, d) + np.sqrt(max(disc, 0))) / np.dot(d, d) >= 1e-6)
        if hit:
            p = o + d * t
            assert abs(np.linalg.norm(p - c) - r) < 1e-9
            assert np.allclose(n, (p - c) / np.linalg.norm(p - c))


def test_triangle_cases():
    a, b, c = [0, 0, 0], [1, 0, 0], [0, 1, 0]
    hit, t, n = O.primitive_hit(2, a, b, c, 0.0, [0.25, 0.25, 2], [0, 0, -1], 1.0, INF)
    assert hit and t == 2.0 and np.allclose(n, [0, 0, 1])
    # the stored face normal is never flipped toward the ray (triangle.rs:114)
    hit, t, n = O.primitive_hit(2, a, b, c, 0.0, [0.25, 0.25, -2], [0, 0, 1], 1.0, INF)
    assert hit and np.allclose(n, [0, 0, 1])
    # closed interval: t == hit.t is accepted and replaces (triangle.rs:90)
    hit, t, _ = O.primitive_hit(2, a, b, c, 0.0, [0.25, 0.25, 2], [0, 0, -1], 1.0, 2.0)
    assert hit and t == 2.0
    # edge-on ray: |det A| < f64::EPSILON (triangle.rs:80)
    hit, _, _ = O.primitive_hit(2, a, b, c, 0.0, [-1, 0.25, 0], [1, 0, 0], 1e-6, INF)
    assert not hit
    # outside the barycentric range
    hit, _, _ = O.primitive_hit(2, a, b, c, 0.0, [0.75, 0.75, 2], [0, 0, -1], 1.0, INF)
    assert not hit
    # the edges gamma = 0 and beta = 0 are inside
    hit, _, _ = O.primitive_hit(2, a, b, c, 0.0, [0.5, 0.0, 2], [0, 0, -1], 1.0, INF)
    assert hit
    rng = np.random.default_rng(3)
    for _ in range(50):
        A, B, C = rng.normal(size=(3, 3))
        w = rng.dirichlet([1, 1, 1])
        target = w[0] * A + w[1] * B + w[2] * C
        o = target + rng.normal(size=3) * 3
        d = (target - o) * rng.uniform(0.5, 2.0)
        hit, t, n = O.primitive_hit(2, A, B, C, 0.0, o, d, 1e-6, INF)
        assert hit
        assert np.allclose(o + d * t, target, atol=1e-9)
        nn = np.cross(B - A, C - A)
        assert np.allclose(n, nn / np.linalg.norm(nn))


def test_cuboid_cases():
    mn, mx = [-1, -2, -3], [1, 2, 3]
    hit, t, n = O.primitive_hit(1, mn, mx, mn, 0.0, [0, 0, 10], [0, 0, -1], 1.0, INF)
    assert hit and t == 7.0 and np.allclose(n, [0, 0, 1])
    hit, t, n = O.primitive_hit(1, mn, mx, mn, 0.0, [5, 0.5, 0.2], [-1, 0, 0], 1.0, INF)
    assert hit and t == 4.0 and np.allclose(n, [1, 0, 0])
    # the normal is chosen on distances normalised by the FULL extent (cuboid.rs:26): x wins only if strictly largest
    hit, t, n = O.primitive_hit(1, mn, mx, mn, 0.0, [0.2, -9, 0.3], [0, 1, 0], 1.0, INF)
    assert hit and t == 7.0 and np.allclose(n, [0, -1, 0])
    # a ray that starts inside reports t_min (see the bbox test)
    hit, t, _ = O.primitive_hit(1, mn, mx, mn, 0.0, [0, 0, 0], [0, 0, -1], 1e-6, INF)
    assert hit and t == 1e-6


def test_camera_basis():
    u, v, w = O.camera_basis([0, 0, -1])
    assert np.allclose(u, [1, 0, 0]) and np.allclose(v, [0, 1, 0]) and np.allclose(w, [0, 0, 1])
    # |up . w| > 0.999: up becomes w with its smallest-magnitude component set to 1 (coordinate_system.rs:16-28)
    u, v, w = O.camera_basis([0, -1, 0])
    assert np.allclose(w, [0, 1, 0])
    # w = (0,1,0): x (0) <= y and <= z -> up = (1,1,0); u = up x w = (0,0,1); v = w x u = (1,0,0)
    assert np.allclose(u, [0, 0, 1]) and np.allclose(v, [1, 0, 0])
    rng = np.random.default_rng(11)
    for _ in range(20):
        view = rng.normal(size=3)
        u, v, w = O.camera_basis(view)
        assert np.allclose(w, -view / np.linalg.norm(view))
        m = np.stack([u, v, w])
        assert np.allclose(m @ m.T, np.eye(3), atol=1e-12)


def test_quantiser_matches_png_export_formula(rt):
    """raytracer-cli/src/output/png_export.rs:4-10: clamp, *255, round half away from zero, y flip."""
    fb = rt.Framebuffer(3, 2)
    fb.pixels[0, 0] = [0.198, 0.198, 0.198]   # DEFAULT_BACKGROUND_COLOR (prelude.rs:41)
    fb.pixels[0, 1] = [-0.5, 0.5, 1.5]
    fb.pixels[1, 2] = [0.2, 0.2, 1.0]
    fb.pixels[1, 0] = [0.5 / 255.0, 1.5 / 255.0, 2.5 / 255.0]
    img = fb.to_rgb8()
    # renders/*.png background corners are (50, 50, 50) = round(0.198 * 255); spheres_1K is (51, 51, 255)
    assert tuple(img[1, 0]) == (50, 50, 50)
    assert tuple(img[1, 1]) == (0, 128, 255)
    assert tuple(img[0, 2]) == (51, 51, 255)
    assert np.array_equal(img, parity.quantize(fb.pixels)[::-1])


def test_background_pixels_of_reference_renders_match_default_background():
    """The published renders are the only numeric output of the reference that exists here:
    their untouched background corners must equal the quantised DEFAULT_BACKGROUND_COLOR."""
    ref_dir = "/root/reference/renders"
    if not os.path.isdir(ref_dir):
        pytest.skip("reference renders not present on this machine")
    from PIL import Image

    px = np.asarray(Image.open(os.path.join(ref_dir, "simple_sphere_scene.png")).convert("RGB"))
    assert tuple(px[0, 0]) == (50, 50, 50) and tuple(px[0, -1]) == (50, 50, 50)
    px = np.asarray(Image.open(os.path.join(ref_dir, "spheres_1K.png")).convert("RGB"))
    assert tuple(px[0, 0]) == (51, 51, 255)


def test_oracle_scene_background_and_determinism(rt, scene_path):
    sc = rt.parse_scene_file(scene_path("simple_sphere_scene"), image_width=64, image_height=48, recursion_depth=6)
    fb1, ids1, c1 = O.render(sc, 2, threads=1)
    fb2, ids2, c2 = O.render(sc, 2, threads=4)
    assert np.array_equal(fb1, fb2) and c1 == c2
    for f in rt.ID_FIELDS:
        assert np.array_equal(getattr(ids1, f), getattr(ids2, f))
    # the top rows of that scene see only background
    q = parity.quantize(fb1)
    assert tuple(q[-1, 0]) == (50, 50, 50)
    # ray accounting: camera rays = W*H*n*n; shadow rays = lit hits x point lights
    assert c1["rays"][0] == 64 * 48 * 4
    assert c1["rays"][2] == c1["shaded_hits"] * sc.n_point_lights


def test_oracle_f32_build_differs_only_slightly(rt, scene_path):
    sc = rt.parse_scene_file(scene_path("spheres_and_triangles"), image_width=96, image_height=96, recursion_depth=6)
    fb64, ids64, _ = O.render(sc, 2, bits=64)
    fb32, ids32, _ = O.render(sc, 2, bits=32)
    m = parity.id_metrics(ids32, ids64)
    assert m["primary"] > 0.999
    assert parity.image_metrics(fb32, fb64)["psnr_db"] > 40


@pytest.mark.parametrize("name", ["simple_sphere_scene", "spheres_and_triangles", "cornell_room", "spheres_1K",
                                  "bunny_small", "mixed_features", "mixed_ortho"])
def test_oracle_reproduces_committed_fixtures(rt, scene_path, name):
    """tests/golden/render_*.npz (written by tests/golden/make_golden.py) pin the oracle against drift."""
    g = np.load(os.path.join(GOLDEN, f"render_{name}.npz"))
    sc = rt.parse_scene_file(scene_path(name), image_width=64, image_height=48, recursion_depth=6)
    fb, ids, cnt = O.render(sc, 2)
    assert cnt["rays"] == g["rays"].tolist() and cnt["shaded_hits"] == int(g["shaded_hits"])
    for f in rt.ID_FIELDS:
        assert np.array_equal(getattr(ids, f), g[f]), f
    assert np.array_equal(fb, g["fb"])


# ---- stochastic antialiasing (antialias.rs:45-57): the generator and the offsets' formulas

def test_philox_known_answers():
    """Philox4x32-10 known-answer vectors of the generator's authors (Random123 kat_vectors:
    zero input, all ones, digits of pi)."""
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for counter, key, want in kat:
        assert O.philox(counter, key).tolist() == want


def test_sample_offsets_follow_the_reference_formulas():
    n = 4
    rs = []
    for sample in range(2000):
        p, q = (sample // n) % n, sample % n
        x = O.philox([sample, 0, 0, 0], [77, 0])
        r1 = float(((int(x[0]) << 32 | int(x[1])) >> 11) / 2.0 ** 53)
        r2 = float(((int(x[2]) << 32 | int(x[3])) >> 11) / 2.0 ** 53)
        assert 0.0 <= r1 < 1.0 and 0.0 <= r2 < 1.0
        rs.append((r1, r2))
        assert O.sample_offsets(0, 77, sample, n, p, q) == ((p + 0.5) / n, (q + 0.5) / n)  # antialias.rs:38-43
        assert O.sample_offsets(1, 77, sample, n, p, q) == ((p + r1) / n, (q + r2) / n)    # antialias.rs:45-50
        assert O.sample_offsets(2, 77, sample, n, p, q) == (r1, r2)                        # antialias.rs:52-57
    rs = np.array(rs)
    # uniform on the unit square: mean 1/2, variance 1/12, the two coordinates uncorrelated
    assert np.abs(rs.mean(axis=0) - 0.5).max() < 0.02
    assert np.abs(rs.var(axis=0) - 1.0 / 12.0).max() < 0.006
    assert abs(np.corrcoef(rs[:, 0], rs[:, 1])[0, 1]) < 0.06
    # another seed, another stream
    assert O.sample_offsets(2, 78, 5, n, 1, 1) != O.sample_offsets(2, 77, 5, n, 1, 1)


def test_stochastic_methods_converge_to_the_same_image(rt, scene_path):
    """Jittered and random sampling estimate the same pixel integral as the regular grid: at 64
    samples per pixel the three oracle images agree closely, and two seeds differ."""
    sc = rt.parse_scene_file(scene_path("simple_sphere_scene"), image_width=64, image_height=48, recursion_depth=3)
    base, _, _ = O.render(sc, 8, want_ids=False)
    jit, _, cj = O.render(sc, 8, want_ids=False, aa=1, seed=1)
    rnd, _, _ = O.render(sc, 8, want_ids=False, aa=2, seed=1)
    jit2, _, _ = O.render(sc, 8, want_ids=False, aa=1, seed=2)
    assert cj["rays"][0] == 64 * 48 * 64
    assert not np.array_equal(jit, jit2) and not np.array_equal(jit, base)
    assert np.abs(jit - base).mean() < 0.004 and np.abs(rnd - base).mean() < 0.012
    assert np.abs(jit - base).mean() < np.abs(rnd - base).mean()  # stratification lowers the variance
