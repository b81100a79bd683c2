"""rt-cli (cli/rt_cli.cpp): the reference CLI's flags and checks (raytracer-cli/src/main.rs:19-105)
and the PNG writer (output/mod.rs:7-17, png_export.rs:1-14)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "cli", "rt-cli")


@pytest.fixture(scope="module")
def cli():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cli")], stdout=subprocess.DEVNULL)
    return CLI


def run(cli, *args):
    p = subprocess.run([cli, *args], capture_output=True, text=True)
    return p.returncode, p.stderr


def test_argument_checks(cli, scene_path):
    rc, err = run(cli)
    assert rc != 0 and "--scene-path" in err
    rc, err = run(cli, "-i", scene_path("simple_sphere_scene"), "-r", "5")
    assert rc != 0 and "rays_per_pixel must be a perfect square" in err  # main.rs:78-83
    rc, err = run(cli, "-i", os.path.join(ROOT, "scenes", "does_not_exist.json"))
    assert rc != 0 and "scene" in err
    rc, err = run(cli, "-i", scene_path("simple_sphere_scene"), "--antialias-method", "blurry")
    assert rc != 0 and "invalid value" in err
    rc, err = run(cli, "-i", scene_path("simple_sphere_scene"), "--backend", "cpu")
    assert rc != 0 and "only 'hip'" in err
    rc, err = run(cli, "--help")
    assert rc == 0 and "--rays-per-pixel" in err


def test_png_writer_round_trip(rt, tmp_path):
    from PIL import Image

    rng = np.random.default_rng(3)
    for w, h in ((1, 1), (7, 5), (300, 250)):  # the last one needs more than one stored deflate block
        fb = rt.Framebuffer(w, h)
        fb.pixels[:] = rng.uniform(-0.3, 1.3, size=(h, w, 3)).astype(np.float32)
        path = str(tmp_path / f"img_{w}x{h}.png")
        rt.save_png(path, fb)
        img = np.asarray(Image.open(path).convert("RGB"))
        assert img.shape == (h, w, 3)
        assert np.array_equal(img, fb.to_rgb8())
    with pytest.raises(rt.RtError) as e:
        rt.save_png(str(tmp_path / "img.jpg"), fb)
    assert e.value.code == rt.RT_ERR_UNSUPPORTED and "'jpg' is not supported" in e.value.message
    rt.save_png(str(tmp_path / "no_extension"), fb)  # output/mod.rs:8: silently nothing
    assert not os.path.exists(str(tmp_path / "no_extension"))


@pytest.mark.gpu
def test_cli_renders_what_the_library_renders(cli, rt, backend, scene_path, tmp_path):
    from PIL import Image

    out = str(tmp_path / "frame.png")
    stats = str(tmp_path / "stats.json")
    ids_path = str(tmp_path / "ids.bin")
    rc, err = run(cli, "-i", scene_path("mixed_features"), "-o", out, "-x", "120", "-y", "80", "-r", "9", "-d", "4",
                  "--stats-json", stats, "--dump-ids", ids_path)
    assert rc == 0, err
    scene = rt.parse_scene_file(scene_path("mixed_features"), image_width=120, image_height=80, recursion_depth=4)
    dev = backend.upload(scene)
    fb, ids, st = dev.render(3, want_ids=True)
    assert np.array_equal(np.asarray(Image.open(out).convert("RGB")), fb.to_rgb8())
    raw = open(ids_path, "rb").read()
    assert raw[:4] == b"RTID"
    version, w, h, spp, npl = np.frombuffer(raw, np.uint32, 5, 4)
    assert (version, w, h, spp, npl) == (1, 120, 80, 9, scene.n_point_lights)
    body = np.frombuffer(raw, np.int32, offset=24)
    samples = 120 * 80 * 9
    assert body.size == samples * (5 + 2 * npl)
    assert np.array_equal(body[:samples], ids.primary_shape) and np.array_equal(body[samples:2 * samples], ids.primary_tri)
    assert np.array_equal(body[2 * samples:3 * samples], ids.final_shape) and np.array_equal(body[4 * samples:5 * samples], ids.bounces)
    assert np.array_equal(body[5 * samples:5 * samples + samples * npl] >= 0, np.asarray(ids.shadow_shape).reshape(-1) >= 0)
    s = json.load(open(stats))
    assert s["primary_rays"] == st.primary_rays == 120 * 80 * 9
    assert s["shadow_rays"] == st.shadow_rays and s["secondary_rays"] == st.secondary_rays
    dev.close()
    # --gpus N: the frame split over N backends in one process (here all on the one device) is the same frame
    multi = str(tmp_path / "multi.png")
    for gpus in ("3", "6"):
        rc, err = run(cli, "-i", scene_path("mixed_features"), "-o", multi, "-x", "120", "-y", "80", "-r", "9", "-d", "4",
                      "--gpus", gpus, "--same-device", "--stats-json", stats)
        assert rc == 0, err
        assert np.array_equal(np.asarray(Image.open(multi).convert("RGB")), np.asarray(Image.open(out).convert("RGB")))
        s = json.load(open(stats))
        assert s["primary_rays"] == 120 * 80 * 9 and s["shadow_rays"] == st.shadow_rays
    # defaults: 360x360, 4 rays per pixel, depth 3 (prelude.rs:54-57)
    rc, err = run(cli, "-i", scene_path("simple_sphere_scene"), "-o", out)
    assert rc == 0, err
    assert Image.open(out).size == (360, 360)
    # jittered sampling: reproducible with --seed, and (like the reference's thread_rng) different without
    shots = []
    for extra in (["--seed", "5"], ["--seed", "5"], []):
        rc, err = run(cli, "-i", scene_path("simple_sphere_scene"), "-o", out, "-x", "96", "-y", "64",
                      "--antialias-method", "jittered", *extra)
        assert rc == 0, err
        shots.append(np.asarray(Image.open(out).convert("RGB")).copy())
    assert np.array_equal(shots[0], shots[1])
    assert not np.array_equal(shots[0], shots[2])
