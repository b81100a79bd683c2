#!/usr/bin/env python3
"""Renders a few scenes with the library RT_HIP_LIB names and writes framebuffers and ray ids to an .npz, so that two
builds can be compared bit for bit (development aid for changes that must not move a single bit).

    RT_HIP_LIB=variants/a.so python3 tools/ab_bits.py gpurun_out/a.npz ; RT_HIP_LIB=... python3 tools/ab_bits.py gpurun_out/b.npz
    python3 tools/ab_bits.py --compare gpurun_out/a.npz gpurun_out/b.npz
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = a[k].tobytes() == b[k].tobytes()
        bad += not same
        print(k, "identical" if same else "DIFFERENT (%d elements)" % int((a[k] != b[k]).sum()))
    sys.exit(1 if bad else 0)

rt = importlib.import_module("rust-raytracer_amd")
out = {}
be = rt.Backend(0)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tempfile

import edge_scenes

jobs = []
for name, size, depth, rpp in [("bunny", 600, 6, 3), ("cornell_room", 600, 6, 3), ("spheres_and_triangles", 400, 6, 2), ("mixed_features", 320, 4, 2)]:
    jobs.append((name, rt.parse_scene_file(os.path.join(ROOT, "scenes", name + ".json"), image_width=size, image_height=size, recursion_depth=depth), rpp))
tmp = tempfile.mkdtemp()
for name, (text, data_dir, w, h, n) in edge_scenes.scenes(tmp).items():
    jobs.append((name, rt.parse_scene(text, data_dir, image_width=4 * w, image_height=4 * h, recursion_depth=4), n))
jobs.append(("ggx", rt.parse_scene(edge_scenes.ggx_scene(), "", image_width=160, image_height=120, recursion_depth=3), 2))
for name, scene, rpp in jobs:
    dev = be.upload(scene)
    for mode, tag in [(rt.RT_MODE_FAST, "fast"), (rt.RT_MODE_REFERENCE_ORDER, "ref")]:
        fb, ids, stats = dev.render(rpp, mode=mode, want_ids=True)
        out[f"{name}_{tag}_fb"] = np.asarray(fb.pixels)
        out[f"{name}_{tag}_final_shape"] = np.asarray(ids.final_shape)
        out[f"{name}_{tag}_bounces"] = np.asarray(ids.bounces)
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "arrays")
