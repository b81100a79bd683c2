#!/usr/bin/env python3
"""Host-side count of what a wave packet touches (tests/hostsim, no GPU): for every group of 64
consecutive rays, the number of distinct nodes / triangles the group visits against its longest lane.

    python3 tools/wave_union.py [workload] [width] [height]
"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "bunny"
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    bench = importlib.import_module("bench")
    rt = importlib.import_module("rust-raytracer_amd")
    import hostsim_lib

    importlib.import_module("scenes.gen_scenes").ensure_models()
    scene_name, _, _, n, depth, tile, _ = bench.WORKLOADS[workload]
    scene = rt.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".json"), image_width=W, image_height=H,
                                recursion_depth=depth)
    L = hostsim_lib.lib()
    L.hostsim_wave_union.argtypes = [C.c_void_p, C.POINTER(rt.RtRenderArgs), C.c_int, C.c_void_p]
    a = rt.RtRenderArgs()
    a.sqrt_rays_per_pixel = n
    a.tile_w = a.tile_h = tile
    for light in range(-1, scene.n_point_lights):
        out = np.zeros(9, np.float64)
        rc = L.hostsim_wave_union(C.cast(scene.ptr, C.c_void_p), C.byref(a), light, out.ctypes.data)
        assert rc == 0
        w = max(out[0], 1)
        print(f"{'camera rays' if light < 0 else f'shadow rays, light {light}'}: {int(out[0])} waves; nodes per wave: union {out[1] / w:.1f}, "
              f"longest lane {out[2] / w:.1f}, mean lane {out[3] / w:.1f}; triangles per wave: union {out[4] / w:.1f}, "
              f"longest lane {out[5] / w:.1f}, mean lane {out[6] / w:.1f}; of the union's nodes, in levels 0-5 of their tree "
              f"{out[7] / w:.1f}, in levels 0-7 {out[8] / w:.1f}")


if __name__ == "__main__":
    main()
