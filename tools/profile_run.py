#!/usr/bin/env python3
"""Renders a few frames of one workload through the C ABI, with no torch and no
oracle, so that `rocprofv3 ... -- python3 tools/profile_run.py` sees only the
render kernels.

    python3 tools/profile_run.py [workload] [frames] [mode] [render flags (RT_RENDER_*)]
"""
from __future__ import annotations

import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "bunny"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    mode = sys.argv[3] if len(sys.argv) > 3 else "fast"
    flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    bench = importlib.import_module("bench")
    rt = importlib.import_module("rust-raytracer_amd")
    importlib.import_module("scenes.gen_scenes").ensure_models()
    scene_name, W, H, n, depth, tile, _ = bench.WORKLOADS[workload]
    scene = rt.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".json"), image_width=W, image_height=H,
                                recursion_depth=depth)
    be = rt.Backend(0)
    dev = be.upload(scene)
    m = rt.RT_MODE_FAST if mode == "fast" else rt.RT_MODE_REFERENCE_ORDER
    best, best_wall = None, None
    for f in range(frames):
        t0 = time.perf_counter()
        fb, _, st = dev.render(n, mode=m, tile=(tile, tile), flags=flags)
        dt = time.perf_counter() - t0
        if best is None or st.render_ms < best.render_ms:
            best = st
        best_wall = dt if best_wall is None else min(best_wall, dt)
    k = {rt.KERNEL_CLASSES[i]: round(best.kernel_ms[i], 4) for i in range(len(rt.KERNEL_CLASSES))}
    print(f"{workload} {mode}: best device {best.render_ms:.3f} ms, rays {best.total_rays()}, "
          f"{best.total_rays() / best.render_ms / 1e6:.3f} Grays/s, kernels {k}, checksum {float(np.sum(fb.pixels)):.6f}")
    # rt_render without statistics: kernels enqueued back to back, the frame copied over the bus, one wait
    pinned = rt.HostBuffer((H, W, 3))
    for label, out in (("pageable", None), ("page-locked", pinned.array)):
        walls = []
        for f in range(frames):
            t0 = time.perf_counter()
            dev.render(n, mode=m, tile=(tile, tile), flags=flags, want_stats=False, out=out)
            walls.append(time.perf_counter() - t0)
        w = min(walls)
        print(f"{workload} {mode}: best wall time of rt_render into {label} host memory {w * 1e3:.3f} ms, "
              f"{best.total_rays() / w / 1e9:.3f} Grays/s")
    pinned.close()
    dev.close()
    be.close()


if __name__ == "__main__":
    main()
