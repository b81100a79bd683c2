#!/usr/bin/env python3
"""When the waves of the two packet kernels of a frame start and end (a development build of the library,
-DRT_WAVE_TIMELINE, logs them): how long the waves run, and how many are resident over the kernel's time.

    make -C rust-raytracer_amd/csrc timeline        (here; the .so travels to the GPU box)
    python3 tools/wave_timeline.py [workload] [parts] [part]
"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RT_HIP_LIB", os.path.join(ROOT, "rust-raytracer_amd", "csrc", "librt_hip_timeline.so"))


def describe(name, t0, t1, slots):
    import numpy as np

    live = t1 > 0
    t0, t1 = t0[live].astype(np.int64), t1[live].astype(np.int64)
    if t0.size == 0:
        print(name, "no waves")
        return
    base = t0.min()
    t0, t1 = (t0 - base) * 10, (t1 - base) * 10  # ns
    dur = t1 - t0
    end = t1.max()
    q = np.percentile(dur, [50, 90, 99, 99.9, 100]) / 1e3
    print(f"{name}: {t0.size} waves with work, kernel span {end / 1e3:.1f} us; wave duration us: median {q[0]:.1f} p90 {q[1]:.1f} "
          f"p99 {q[2]:.1f} p99.9 {q[3]:.1f} max {q[4]:.1f}; sum {dur.sum() / 1e3:.0f} us = {dur.sum() / end / slots:.3f} of {slots} slots")
    # resident waves over time, in 20 steps
    steps = 20
    edges = np.linspace(0, end, steps + 1)
    occ = []
    for a, b in zip(edges[:-1], edges[1:]):
        overlap = np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None).sum() / (b - a)
        occ.append(overlap)
    print("   resident waves per twentieth of the span:", " ".join(f"{o:.0f}" for o in occ))
    # the longest waves: when they start (a wave that starts late and runs long is the kernel's tail)
    top = np.argsort(dur)[-8:][::-1]
    print("   longest waves (start us, duration us, index):", " ".join(f"({t0[i] / 1e3:.0f}, {dur[i] / 1e3:.0f}, {i})" for i in top))
    # when the last wave STARTS, and what runs after that
    print(f"   last start at {t0.max() / 1e3:.1f} us; waves still running 10/20/40 us before the end: "
          f"{(t1 > end - 10000).sum()} {(t1 > end - 20000).sum()} {(t1 > end - 40000).sum()}")


def simulate(name, dur, order, slots):
    """List scheduling of the measured wave durations on `slots` wave slots in the given order: the span a kernel
    would have with its waves started in that order (durations as measured, i.e. under the contention of this run)."""
    import heapq

    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for i in order:
        t = heapq.heappop(free) + dur[i]
        end = max(end, t)
        heapq.heappush(free, t)
    print(f"   simulated span, {name}: {end / 1e3:.1f} us")


def orders(cam, sh, slots):
    """What the shadow kernel's span would be with its packets started in another order: as they were, longest first
    (unknowable), and the chunks of 16 packets in the order of what the camera-ray waves of the same chunk cost."""
    import numpy as np

    live = sh[:, 1] > 0
    idx = np.nonzero(live)[0]
    t0 = (sh[idx, 0] - sh[idx, 0].min()) * 10.0
    dur = (sh[idx, 1] - sh[idx, 0]) * 10.0
    simulate("as started", dur, np.argsort(t0, kind="stable"), slots)
    simulate("longest first (an oracle)", dur, np.argsort(-dur, kind="stable"), slots)
    cam_dur = np.where(cam[:, 1] > 0, (cam[:, 1] - cam[:, 0]) * 10.0, 0.0)
    n_chunks = -(-cam.shape[0] // 16)
    cost = np.zeros(n_chunks)
    np.add.at(cost, np.arange(cam.shape[0]) // 16, cam_dur)
    chunk_of = idx // 16
    simulate("chunks by their camera waves' time, dearest first", dur, np.lexsort((t0, -cost[chunk_of])), slots)
    heavy = cost[chunk_of] > 3.0 * cost.mean()
    simulate("chunks above 3x the mean camera time first, the rest as started", dur, np.lexsort((t0, ~heavy)), slots)
    own = np.zeros(n_chunks)
    np.add.at(own, chunk_of, dur)
    simulate("chunks by their own shadow waves' time (an oracle per chunk)", dur, np.lexsort((t0, -own[chunk_of])), slots)


def main():
    import numpy as np
    import torch

    workload = sys.argv[1] if len(sys.argv) > 1 else "bunny"
    parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    part = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    bench = importlib.import_module("bench")
    rt = importlib.import_module("rust-raytracer_amd")
    importlib.import_module("scenes.gen_scenes").ensure_models()
    scene_name, W, H, n, depth, tile, _ = bench.WORKLOADS[workload]
    scene = rt.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".json"), image_width=W, image_height=H,
                                recursion_depth=depth)
    be = rt.Backend(0)
    dev = be.upload(scene)
    a = dev.render_args(n, tile=(tile, tile), part=(part, parts))
    _, owned, stride = dev.partition_info(a)
    packets = -(-owned * tile * tile * n * n // 64)
    log = torch.zeros(34 * packets + 64, dtype=torch.int64, device="cuda")  # camera, shadow, 6 bounces x (walk, shade), what follows the walks
    fb = torch.zeros(W * H * 3 + tile * tile * 3 * 700, dtype=torch.float32, device="cuda")
    L = rt.lib()
    L.rt_debug_wave_log.argtypes = [C.c_void_p]
    for _ in range(5):
        dev.render_device(a, fb.data_ptr())
        be.sync()
    log.zero_()
    torch.cuda.synchronize()
    assert L.rt_debug_wave_log(C.c_void_p(log.data_ptr())) == 0
    dev.render_device(a, fb.data_ptr())
    be.sync()
    torch.cuda.synchronize()
    assert L.rt_debug_wave_log(None) == 0
    v = log.cpu().numpy()
    cam = v[:2 * packets].reshape(-1, 2)
    sh = v[2 * packets:4 * packets].reshape(-1, 2)
    print(f"{workload}, part {part} of {parts}: {packets} packets")
    describe("camera rays + shade", cam[:, 0], cam[:, 1], 1024 * 7)
    describe("shadow rays + light sums", sh[:, 0], sh[:, 1], 1024 * 8)
    orders(cam, sh, 1024 * 8)
    both0 = np.concatenate([cam[:, 0], sh[:, 0]])
    both1 = np.concatenate([cam[:, 1], sh[:, 1]])
    describe("both kernels", both0, both1, 1024 * 8)
    for name, whole, after in (("camera kernel", cam, v[28 * packets:30 * packets].reshape(-1, 2)),
                               ("shadow kernel", sh, v[30 * packets:32 * packets].reshape(-1, 2))):
        live = (whole[:, 1] > 0) & (after[:, 1] > 0)
        if live.any():
            front = (after[live, 0] - whole[live, 0]) * 10 / 1e3
            back = (after[live, 1] - after[live, 0]) * 10 / 1e3
            print(f"{name}: from the wave's start to the end of its walk(s) us: median {np.median(front):.1f} mean {front.mean():.1f} p90 {np.percentile(front, 90):.1f}; "
                  f"what follows (the stage behind the hits / the light sums): median {np.median(back):.1f} mean {back.mean():.1f} p90 {np.percentile(back, 90):.1f}")
    head = v[32 * packets:34 * packets].reshape(-1, 2)
    live = (head[:, 1] > 0) & (sh[:, 1] > 0)
    if live.any():
        way_in = (head[live, 1] - head[live, 0]) * 10 / 1e3
        print(f"shadow kernel: the way in (chunk count, list entry, lit record into LDS) us: median {np.median(way_in):.1f} mean {way_in.mean():.1f} p90 {np.percentile(way_in, 90):.1f}")
    bo = v[4 * packets:28 * packets].reshape(packets, 6, 2, 2)  # [packet][bounce][walk / shade][start, end]
    if (bo[:, :, 0, 1] > 0).any():
        for b in range(6):
            live = bo[:, b, 1, 1] > 0
            if not live.any():
                continue
            walk = (bo[live, b, 0, 1] - bo[live, b, 0, 0]) * 10 / 1e3
            shade = (bo[live, b, 1, 1] - bo[live, b, 1, 0]) * 10 / 1e3
            print(f"mirror bounce {b + 1}: {int(live.sum())} packets; walk us: median {np.median(walk):.1f} p90 {np.percentile(walk, 90):.1f} "
                  f"max {walk.max():.1f}; stage behind it: median {np.median(shade):.1f} p90 {np.percentile(shade, 90):.1f} max {shade.max():.1f}")
        live = bo[:, :, 1, 1] > 0
        t0 = np.where(live, bo[:, :, 0, 0], np.iinfo(np.int64).max).min(axis=1)
        t1 = bo[:, :, 1, 1].max(axis=1)
        has = live.any(axis=1)
        describe("mirror bounces, per packet (first walk to last stage)", t0[has], t1[has], 1024 * 7)
    dev.close()
    be.close()


if __name__ == "__main__":
    main()
