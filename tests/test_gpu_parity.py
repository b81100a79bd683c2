"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle."""
import os

import numpy as np
import pytest

import oracle_lib
import parity

pytestmark = pytest.mark.gpu

SCENES = ["simple_sphere_scene", "spheres_and_triangles", "cornell_room", "spheres_1K", "bunny_small",
          "mixed_features", "mixed_ortho"]


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("mode", [0, 1])
def test_scene_matches_oracle(rt, backend, scene_path, name, mode):
    scene = rt.parse_scene_file(scene_path(name), image_width=160, image_height=120, recursion_depth=6)
    dev = backend.upload(scene)
    fb, ids, stats = dev.render(3, mode=mode, want_ids=True)
    ref_fb, ref_ids, counters = oracle_lib.render(scene, 3, want_ids=True)
    im, idm = parity.assert_parity(fb.pixels, ids, ref_fb, ref_ids, occlusion_only=(mode == rt.RT_MODE_FAST),
                                   label=f"{name} mode {mode}")
    got = [stats.primary_rays, stats.secondary_rays, stats.shadow_rays]
    # ray counts follow from the ids, so they may differ by the same handful of rays
    for g, r in zip(got, counters["rays"]):
        assert abs(g - r) <= max(2, 1e-4 * r), (got, counters["rays"])
    dev.close()
