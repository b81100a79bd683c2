#!/usr/bin/env python3
"""Writes tests/golden/render_<scene>.npz: the CPU oracle's framebuffer, per-sample ids and ray
counts for every test scene at 64x48, 4 rays per pixel, depth 6, and
render_<scene>_<method>.npz for the two stochastic antialiasing methods (seed 2024) on two scenes,
and render_ggx_scene.npz for the GGX ray trees.

    python3 tests/golden/make_golden.py

The fixtures come from this repository's own oracle (oracle/rt_oracle.cpp), not from the
reference program, which cannot be built here.  They let a machine without the oracle's
build check the GPU path, and they catch unintended changes of the oracle itself.
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = ["simple_sphere_scene", "spheres_and_triangles", "cornell_room", "spheres_1K", "bunny_small", "mixed_features",
          "mixed_ortho"]
WIDTH, HEIGHT, SQRT_RPP, DEPTH = 64, 48, 2, 6
STOCHASTIC = [("mixed_features", 1, "jittered"), ("mixed_features", 2, "random"), ("bunny_small", 1, "jittered")]
STOCHASTIC_SEED = 2024


def main():
    import oracle_lib as O

    rt = O.rt
    importlib.import_module("scenes.gen_scenes").ensure_models()
    for name in SCENES:
        sc = rt.parse_scene_file(os.path.join(ROOT, "scenes", name + ".json"), image_width=WIDTH, image_height=HEIGHT,
                                 recursion_depth=DEPTH)
        fb, ids, cnt = O.render(sc, SQRT_RPP, threads=1)
        arrays = {f: getattr(ids, f) for f in rt.ID_FIELDS}
        np.savez_compressed(os.path.join(HERE, f"render_{name}.npz"), fb=fb, rays=np.array(cnt["rays"], np.int64),
                            shaded_hits=np.int64(cnt["shaded_hits"]), **arrays)
        print(name, cnt["rays"])
    for name, method, label in STOCHASTIC:
        sc = rt.parse_scene_file(os.path.join(ROOT, "scenes", name + ".json"), image_width=WIDTH, image_height=HEIGHT,
                                 recursion_depth=DEPTH)
        fb, ids, cnt = O.render(sc, SQRT_RPP, threads=1, aa=method, seed=STOCHASTIC_SEED)
        arrays = {f: getattr(ids, f) for f in rt.ID_FIELDS}
        np.savez_compressed(os.path.join(HERE, f"render_{name}_{label}.npz"), fb=fb, rays=np.array(cnt["rays"], np.int64),
                            shaded_hits=np.int64(cnt["shaded_hits"]), method=np.int64(method),
                            seed=np.int64(STOCHASTIC_SEED), **arrays)
        print(name, label, cnt["rays"])
    # GGXMirror ray trees (tests/edge_scenes.py::ggx_scene), seed 2024, depth 3
    import edge_scenes
    sc = rt.parse_scene(edge_scenes.ggx_scene(), "", image_width=WIDTH, image_height=HEIGHT, recursion_depth=3)
    fb, ids, cnt = O.render(sc, SQRT_RPP, threads=1, seed=STOCHASTIC_SEED)
    np.savez_compressed(os.path.join(HERE, "render_ggx_scene.npz"), fb=fb, rays=np.array(cnt["rays"], np.int64),
                        seed=np.int64(STOCHASTIC_SEED), primary_shape=ids.primary_shape, primary_tri=ids.primary_tri)
    print("ggx_scene", cnt["rays"])


if __name__ == "__main__":
    main()
