#!/usr/bin/env python3
"""How long one GPU's share of a frame takes when the frame is split over N parts, with the frames
enqueued back to back (no host synchronisation), against 1/N of the whole frame.

    python3 tools/part_timing.py [workload] [frames] [frames in flight]
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    workload = sys.argv[1] if len(sys.argv) > 1 else "bunny"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    bench = importlib.import_module("bench")
    rt = importlib.import_module("rust-raytracer_amd")
    importlib.import_module("scenes.gen_scenes").ensure_models()
    scene_name, W, H, n, depth, tile, _ = bench.WORKLOADS[workload]
    tile = int(os.environ.get("RT_TILE", tile))
    every = os.environ.get("RT_ALL_PARTS", "0") == "1"  # time every part instead of first / middle / last
    # RT_SPAN=K: each render covers K consecutive parts (RtRenderArgs.part_span), as a GPU does that renders what it
    # owes K frames of a stream with one set of launches; the figure is per part.  0: as many as make a quarter frame.
    span_env = int(os.environ.get("RT_SPAN", "1"))
    scene = rt.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".json"), image_width=W, image_height=H,
                                recursion_depth=depth)
    flights = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # frames in flight: one backend + stream each
    bes = [rt.Backend(0) for _ in range(flights)]
    first = bes[0].upload(scene)  # one copy of the scene for all frames in flight
    devs = [first] + [first.on(b) for b in bes[1:]]
    sts = [torch.cuda.Stream() for _ in range(flights)]
    whole = None
    for parts in (1, 2, 4, 8):
        worst = host = 0.0
        each = []
        span = span_env if span_env > 0 else max(1, parts // 4)
        span = min(span, parts)
        for p in (range(parts) if every else sorted({0, parts // 2, parts - 1})):
            a = devs[0].render_args(n, tile=(tile, tile), part=(p, parts), part_span=span if parts > 1 else 0)
            _, owned, stride = devs[0].partition_info(a)
            bufs = [torch.zeros(max(span * stride * tile * tile * 3, W * H * 3), dtype=torch.float32, device="cuda") for _ in range(flights)]
            for k in range(3 * flights):
                devs[k % flights].render_device(a, bufs[k % flights].data_ptr(), stream=sts[k % flights].cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(frames):
                devs[k % flights].render_device(a, bufs[k % flights].data_ptr(), stream=sts[k % flights].cuda_stream)
            enq = (time.perf_counter() - t0) / frames * 1e3  # host time to enqueue one frame
            torch.cuda.synchronize()
            each.append((time.perf_counter() - t0) / frames * 1e3 / span)
            worst = max(worst, each[-1])
            host = max(host, enq)
        if parts == 1:
            whole = worst
        print(f"{workload}, {flights} frame(s) in flight: {parts} parts" + (f", {span} per set of launches" if span > 1 else "") + f": slowest part {worst:.3f} ms per frame; "
              f"ideal {whole / parts:.3f} ms; efficiency without the gather {whole / parts / worst:.2f}; "
              f"host enqueue {host:.3f} ms per frame" + (f"; parts {' '.join(f'{e:.3f}' for e in each)}" if every else ""))
    for d in devs:
        d.close()
    for b in bes:
        b.close()


if __name__ == "__main__":
    main()
