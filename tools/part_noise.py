#!/usr/bin/env python3
"""Run-to-run spread of the per-part frame time: times the parts of an N-part split in the order
given, so that the same part can be measured several times.

    python3 tools/part_noise.py [parts] [frames] [order, e.g. 0,1,0,1,4,4] [frames in flight]
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    order = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(parts)) * 2
    flights = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    bench = importlib.import_module("bench")
    rt = importlib.import_module("rust-raytracer_amd")
    importlib.import_module("scenes.gen_scenes").ensure_models()
    scene_name, W, H, n, depth, tile, _ = bench.WORKLOADS["bunny"]
    tile = int(os.environ.get("RT_TILE", tile))
    scene = rt.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".json"), image_width=W, image_height=H,
                                recursion_depth=depth)
    bes = [rt.Backend(0) for _ in range(flights)]
    first = bes[0].upload(scene)  # one copy of the scene for all frames in flight
    devs = [first] + [first.on(b) for b in bes[1:]]
    sts = [torch.cuda.Stream() for _ in range(flights)]
    bufs = [torch.zeros(W * H * 3, dtype=torch.float32, device="cuda") for _ in range(flights)]
    out = []
    for p in order:
        a = devs[0].render_args(n, tile=(tile, tile), part=(p, parts), part0_weight=int(os.environ.get("RT_PART0_WEIGHT", "0")))
        for k in range(3 * flights):
            devs[k % flights].render_device(a, bufs[k % flights].data_ptr(), stream=sts[k % flights].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(frames):
            devs[k % flights].render_device(a, bufs[k % flights].data_ptr(), stream=sts[k % flights].cuda_stream)
        torch.cuda.synchronize()
        out.append((p, (time.perf_counter() - t0) / frames * 1e3))
    print(" ".join(f"{p}:{ms:.3f}" for p, ms in out))
    for d in devs:
        d.close()
    for b in bes:
        b.close()


if __name__ == "__main__":
    main()
