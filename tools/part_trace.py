#!/usr/bin/env python3
"""Two uses.

  python3 tools/part_trace.py run <parts> <part> <frames> <flights>
      renders one part of the bunny frame repeatedly, `flights` frames in flight (run it under
      rocprofv3 --kernel-trace);
  python3 tools/part_trace.py read <kernel_trace.csv>
      reads the trace: per kernel name the mean duration, how much of the wall time at least one kernel
      was running, how many ran at once on average, and the wall time per frame.
"""
import csv
import importlib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(parts, part, frames, flights):
    import torch

    bench = importlib.import_module("bench")
    rt = importlib.import_module("rust-raytracer_amd")
    importlib.import_module("scenes.gen_scenes").ensure_models()
    scene_name, W, H, n, depth, tile, _ = bench.WORKLOADS["bunny"]
    scene = rt.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".json"), image_width=W, image_height=H,
                                recursion_depth=depth)
    bes = [rt.Backend(0) for _ in range(flights)]
    first = bes[0].upload(scene)  # one copy of the scene for all frames in flight
    devs = [first] + [first.on(b) for b in bes[1:]]
    sts = [torch.cuda.Stream() for _ in range(flights)]
    bufs = [torch.zeros(W * H * 3, dtype=torch.float32, device="cuda") for _ in range(flights)]
    a = devs[0].render_args(n, tile=(tile, tile), part=(part, parts))
    for k in range(frames):
        devs[k % flights].render_device(a, bufs[k % flights].data_ptr(), stream=sts[k % flights].cuda_stream)
    torch.cuda.synchronize()
    for d in devs:
        d.close()
    for b in bes:
        b.close()


def read(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            m = re.search(r"(k_\w+)", r["Kernel_Name"])
            if not m:
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1)))
    rows.sort()
    rows = rows[len(rows) // 3:]  # the settled part of the run
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    by = {}
    for s, e, n in rows:
        by.setdefault(n, []).append(e - s)
    first = rows[0][2]
    frames = sum(1 for r in rows if r[2] == first)
    print(f"{frames} frames in {(t1 - t0) / 1e3:.1f} us: {(t1 - t0) / 1e3 / frames:.1f} us per frame")
    for n, d in by.items():
        d.sort()
        print(f"  {n:26s} x{len(d):5d} mean {sum(d) / len(d) / 1e3:7.1f} us  median {d[len(d) // 2] / 1e3:7.1f}  p90 {d[len(d) * 9 // 10] / 1e3:7.1f}")
    # coverage and concurrency
    ev = []
    for s, e, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    live = 0
    last = ev[0][0]
    covered = 0
    area = 0
    for t, d in ev:
        if live > 0:
            covered += t - last
            area += (t - last) * live
        live += d
        last = t
    print(f"  some kernel running {covered / (t1 - t0):.3f} of the time; kernels running at once, mean {area / max(covered, 1):.2f}")
    # the gap between a kernel's end and the start of the next kernel of the same frame (stream order)
    kinds = list(by)
    print("  kernel time per frame, summed:", round(sum(sum(d) for d in by.values()) / frames / 1e3, 1), "us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(*(int(x) for x in sys.argv[2:6]))
    else:
        read(sys.argv[2])
