#!/usr/bin/env python3
"""Benchmark of the render hot path: rays/s (camera + mirror + shadow rays).

    python bench.py --gpus 1 --steps 200 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step is one render of the whole frame with the scene already resident in HBM.
With N > 1 the frame is cut into interleaved tiles, rank r renders the tiles of
part r into a compact device buffer, an RCCL gather brings the buffers to rank 0
(one gather for the frames in flight together) and a kernel there restores the
image layout of every frame; total work is fixed, so scaling is "strong".  Rank 0
prints one JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, datasheet

WORKLOADS = {
    # name: (scene, width, height, sqrt_rpp, depth, tile, description)
    "bunny": ("bunny", 1000, 1000, 3, 6, 40,
              "bunny 1000x1000 rpp9 depth6: 3 instances of an 81920-triangle bunny-class stand-in mesh "
              "(procedural; the Stanford bunny file is not available) + ground box, 2 point lights"),
    "bunny4k": ("bunny", 3840, 2160, 8, 6, 32, "bunny 3840x2160 rpp64 depth6 (same scene)"),
    "spheres_1K": ("spheres_1K", 1000, 1000, 3, 6, 40, "spheres_1K 1000x1000 rpp9 depth6: 10x10x10 spheres, 4 point lights"),
    "cornell_room": ("cornell_room", 1000, 1000, 3, 6, 40, "cornell_room 1000x1000 rpp9 depth6: 6 boxes + 6 mirror spheres"),
    "spheres_and_triangles": ("spheres_and_triangles", 1000, 1000, 3, 6, 40, "spheres_and_triangles 1000x1000 rpp9 depth6"),
    "simple_sphere_scene": ("simple_sphere_scene", 1000, 1000, 3, 6, 40, "simple_sphere_scene 1000x1000 rpp9 depth6"),
}


def algorithmic_bytes(workload):
    path = os.path.join(ROOT, "bench", "algorithmic_bytes.json")
    key = "bunny" if workload == "bunny4k" else workload
    try:
        with open(path) as f:
            return json.load(f)["workloads"][key]["per_class"]
    except Exception:
        return None


def measured_counters(workload, kernel_class):
    """Counter figures of the dominant kernel from the committed PMC profile of this same command
    (profiles/pmc_traffic.json): HBM GB/s actually moved, L2 hit rate, VALU lane utilisation; or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(workload)
        if measured_traffic(workload, kernel_class) is None:
            return None
        return {k: entry.get(k) for k in ("hbm_gbs", "l2_hit_rate", "valu_lane_utilisation")}
    except Exception:
        return None


def measured_traffic(workload, kernel_class):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile of this same
    command (profiles/pmc_traffic.json, written by tools/summarize_pmc.py), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    want = {"trace_shadow": ("k_trace_packet<1,", "k_trace_fast<1>", "k_trace_shadow"),
            "trace_closest": ("k_trace_packet<0,", "k_trace_fast<0>", "k_trace_closest")}.get(kernel_class, ())
    try:
        with open(path) as f:
            entry = json.load(f).get(workload)
        return entry["bytes"] if entry and entry.get("kernel", "").startswith(want) else None
    except Exception:
        return None


def measured_hbm_gbs(torch):
    """What the box's HBM delivers to a plain streaming kernel, as a second denominator beside the
    datasheet figure: the better of a device-to-device copy (read + write) and a triad a = b + s * c
    (two reads + one write) over 1 GiB arrays, best of five runs each, in GB/s."""
    n = 1 << 28  # floats
    a, b, c = (torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(3))
    b.fill_(1.0)
    c.fill_(2.0)
    best = 0.0
    for op, moved in ((lambda: a.copy_(b), 2), (lambda: torch.add(b, c, alpha=0.5, out=a), 3)):
        op()
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            op()
            e1.record()
            e1.synchronize()
            best = max(best, moved * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b, c
    return best


def cpu_baseline(rt, scene_name, depth, sample_px, n, threads):
    import oracle_lib

    sc = rt.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".json"), image_width=sample_px,
                             image_height=sample_px, recursion_depth=depth)
    t0 = time.perf_counter()
    _, _, counters = oracle_lib.render(sc, n, want_ids=False, threads=threads)
    dt = time.perf_counter() - t0
    return sum(counters["rays"]) / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="bunny", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fast", choices=["fast", "reference"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # A frame cut into more than four parts leaves each GPU launches so small that a kernel lasts as
    # long as its slowest wave, and a frame's chain of kernels (0.5-0.7 ms for a 1/8 frame alone) sets
    # the pace: with HIP's default of four hardware queues at most four chains overlap.  Eight frames
    # in flight on queues of their own took the slowest 1/8 part from 0.184 to 0.177 ms per frame.  The
    # gather and un-tile streams need queues of their own as well: a stream that shares a hardware queue
    # with a render stream holds up that stream's kernels whenever it waits for an event (rank 0's step
    # with 8 queues for 11 streams: 0.25 ms per frame, with 32 queues: 0.19).  For a whole, half or
    # quarter frame the runtime's default is as good or slightly better.  The variable is read when the
    # HIP runtime starts, so it is set before torch is imported.
    many_parts = int(os.environ.get("WORLD_SIZE", "1")) > 4
    hw_queues = os.environ.get("RT_BENCH_HW_QUEUES", "32" if many_parts else "")
    if hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = hw_queues

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # RT_BENCH_MULTI_PATH=1 runs the N > 1 code (process group, grouped gather, un-tile) with a single
    # rank, as a check of that path on a one-GPU machine; the figures are then not the N = 1 benchmark.
    multi = world > 1 or os.environ.get("RT_BENCH_MULTI_PATH", "0") == "1"
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU implementation")
    if os.environ.get("RT_BENCH_BACKEND", "nccl") == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if multi:
        if os.environ.get("RT_BENCH_BACKEND", "nccl") == "gloo":  # rehearsal of the N > 1 path on a single GPU
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    rt = importlib.import_module("rust-raytracer_amd")
    gen = importlib.import_module("scenes.gen_scenes")
    if rank == 0:
        gen.ensure_models()
    if multi:
        dist.barrier()

    scene_name, W, H, n, depth, tile, desc = WORKLOADS[args.workload]
    scene = rt.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".json"), image_width=W, image_height=H,
                                recursion_depth=depth)
    # FLIGHTS frames are in flight at a time, each with a backend (ray queues), a stream and
    # output buffers of its own: the tail of one frame's kernels (a few long-running waves) then
    # overlaps the next frame's work instead of idling the GPU.  Measured on one MI355X for this
    # frame cut into 8 parts: 0.54 ms per part-frame with one frame in flight, 0.28 ms with four
    # (the whole frame: 1.82 -> 1.76 ms).
    # frames in flight per GPU (whole frame: 3 to 8 measure within 2 % of each other, 2 is 9 % slower)
    FLIGHTS = int(os.environ.get("RT_BENCH_FLIGHTS", "8" if many_parts else "4"))
    backends = [rt.Backend(local_rank) for _ in range(FLIGHTS)]
    t_up = time.perf_counter()
    devs = [be.upload(scene) for be in backends]
    upload_s = (time.perf_counter() - t_up) / FLIGHTS
    dev = devs[0]
    mode = rt.RT_MODE_FAST if args.mode == "fast" else rt.RT_MODE_REFERENCE_ORDER
    # Rank 0 also receives the other ranks' tiles and restores the image layout of every frame: with
    # 1/8 frames that is a tenth of its time even when the gather is a local copy
    # (tools/gather_host_cost.py: 0.184 ms per frame against 0.169 for the render alone), so from five
    # GPUs on it takes 6/8 of the share the others get: its whole step then measures 0.152 ms, which
    # leaves room for what a real seven-peer receive adds, and the other parts 0.164-0.180 ms
    # (equal shares: 0.165-0.179).
    part0_weight = int(os.environ.get("RT_BENCH_PART0_WEIGHT", "6" if many_parts else "0"))
    rargs = dev.render_args(n, mode=mode, tile=(tile, tile), part=(rank, world), part0_weight=part0_weight)
    tiles_total, tiles_owned, stride = dev.partition_info(rargs)
    tile_floats = tile * tile * 3

    # Streams of our own: the backend treats a null stream handle as "use the backend's private
    # stream", which would not be ordered with torch's copies and collectives.
    streams = [torch.cuda.Stream() for _ in range(FLIGHTS)]
    torch.cuda.set_stream(streams[0])
    staged = multi and dist.get_backend() != "nccl"  # rehearsal on one GPU: gather through host memory (gloo)
    counter = [0]

    if not multi:
        fbs = [torch.zeros(W * H * 3, dtype=torch.float32, device="cuda") for _ in range(FLIGHTS)]
        fb = fbs[0]

        def step(want_stats):
            b = counter[0] % FLIGHTS
            counter[0] += 1
            return devs[b].render_device(rargs, fbs[b].data_ptr(), stream=streams[b].cuda_stream, want_stats=want_stats)

        def drain():
            pass
    else:
        # GROUP consecutive frames are gathered together: every rank renders its part of each into one
        # buffer, ONE gather brings all of them to rank 0, and rank 0 restores the image layout frame
        # by frame on a side stream.  Two groups alternate, so the gather and un-tiling of one overlap
        # the renders of the other.  Measured on one MI355X with a one-rank RCCL group
        # (tools/gather_host_cost.py: the gather degenerates to a local copy, so this is a lower bound),
        # per 1/8 frame: render alone 0.169 ms; with a gather and an un-tile per frame 0.245 ms; per
        # group of 8 frames 0.191 ms, of 16 frames 0.183 ms (of which the un-tile kernels are 0.010).
        GROUP = 2 * FLIGHTS
        part_floats = stride * tile_floats
        part = [torch.zeros(GROUP * part_floats, dtype=torch.float32, device="cuda") for _ in range(2)]
        root = rank == 0
        gathered = [torch.zeros(world * GROUP * part_floats, dtype=torch.float32, device="cuda") if root else None for _ in range(2)]
        fb = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda") if root else None
        joiner = torch.cuda.Stream()     # waits for the frames of a group, then issues their gather
        side = torch.cuda.Stream()       # rank 0: waits for a gather and restores the image layout
        sent = [None, None]              # event: the gather that read part[g] / wrote gathered[g] is done
        untiled = [None, None]           # event: gathered[g] has been consumed

        def flush(g, count):
            """Gathers the first `count` frames of group g and, on rank 0, un-tiles them."""
            for b in range(min(count, FLIGHTS)):
                joiner.wait_stream(streams[b])
            mine = part[g][: count * part_floats]
            with torch.cuda.stream(joiner):
                if staged:  # rehearsal on one GPU: through host memory
                    host = mine.cpu()
                    bufs = [torch.empty_like(host) for _ in range(world)] if root else None
                    dist.gather(host, bufs, dst=0)
                    if root:
                        gathered[g][: world * count * part_floats].copy_(torch.cat(bufs))
                else:
                    if untiled[g] is not None:
                        joiner.wait_event(untiled[g])
                    into = [gathered[g][r * count * part_floats:(r + 1) * count * part_floats] for r in range(world)] if root else None
                    dist.gather(mine, into, dst=0)  # returns once enqueued; this stream then waits for RCCL's
                sent[g] = joiner.record_event()
            if root:
                with torch.cuda.stream(side):
                    side.wait_event(sent[g])
                    # gathered[g] holds [rank][frame][block]: frame f starts at f blocks, the ranks lie count blocks apart
                    for f in range(count):
                        dev.untile_device(rargs, gathered[g].data_ptr() + f * part_floats * 4, fb.data_ptr(),
                                          stream=side.cuda_stream, part_pitch_floats=count * part_floats)
                    untiled[g] = side.record_event()

        def step(want_stats):
            k = counter[0]
            counter[0] += 1
            g, f, b = (k // GROUP) % 2, k % GROUP, k % FLIGHTS
            with torch.cuda.stream(streams[b]):
                if sent[g] is not None:
                    streams[b].wait_event(sent[g])  # the gather of two groups ago has released part[g]
                st = devs[b].render_device(rargs, part[g].data_ptr() + f * part_floats * 4, stream=streams[b].cuda_stream,
                                           want_stats=want_stats)
            if f == GROUP - 1:
                flush(g, GROUP)
            return st

        def drain():
            k = counter[0]
            if k % GROUP:
                flush((k // GROUP) % 2, k % GROUP)
                counter[0] = (k // GROUP + 1) * GROUP  # the next frame starts a group
            torch.cuda.current_stream().wait_stream(joiner)
            torch.cuda.current_stream().wait_stream(side)

    def fence():
        drain()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up frames are rendered with the backend's full statistics (an event pair around every
    # kernel, counters read back, the host waits for each frame): they give the ray counts, which
    # are the same for every frame, and the split of a frame over the kernel classes.  The timed
    # frames are enqueued back to back without waiting; only the dominant kernel class carries an
    # event pair there (its durations are collected after the loop for the roofline), because
    # every pair drains the pipeline for a few microseconds.
    # Every flight twice at least: a backend's first frame touches its buffers for the first time, and
    # the kernel split is read from the last warm-up frame.
    warm_stats = [step(True) for _ in range(max(args.warmup, 2 * FLIGHTS))]
    fence()
    k_warm = [warm_stats[-1].kernel_ms[k] for k in range(rt.RT_K_COUNT)]
    dom = max(range(len(rt.KERNEL_CLASSES)), key=lambda k: k_warm[k])
    rargs.timed_classes = 1 << dom
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    fence()
    elapsed = time.perf_counter() - t0
    timed = rt.RtStats()
    for be in backends:
        part_t = be.collect_timings()
        for k in range(rt.RT_K_COUNT):
            timed.kernel_ms[k] += part_t.kernel_ms[k]
            timed.kernel_launches[k] += part_t.kernel_launches[k]
    rargs.timed_classes = 0
    stats = warm_stats[-1:]  # the last warm-up frame: caches and clocks have settled
    rays_local = float(warm_stats[-1].total_rays() * args.steps)

    # Outside the timed region: the gathered frame must be bit-identical with the same frame rendered
    # whole on rank 0 (static tile map, ordered per-pixel sums).
    frame_check = None
    if multi and world == 1:
        frame_check = "not applicable: a single part is rendered in image layout, not tile by tile"
    elif multi and rank == 0:
        whole = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
        dev.render_device(dev.render_args(n, mode=mode, tile=(tile, tile)), whole.data_ptr(), stream=streams[0].cuda_stream)
        torch.cuda.synchronize()
        if not torch.equal(whole, fb):
            raise SystemExit(f"gathered frame differs from the single-GPU frame in {int((whole != fb).sum())} values")
        frame_check = "gathered frame bit-identical with rank 0's single-GPU render"

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    rays = torch.tensor([rays_local], dtype=torch.float64, device="cuda")
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays = float(rays.item())

    if rank == 0:
        value = total_rays / elapsed
        s0 = stats[0]
        n_stat = len(stats)  # frames the kernel timings below were taken over
        # dominant kernel = the class with the most device time on this rank (warm-up frames);
        # its launch duration is the one measured in the timed frames
        k_ms = [sum(s.kernel_ms[k] for s in stats) for k in range(rt.RT_K_COUNT)]
        k_n = [sum(s.kernel_launches[k] for s in stats) for k in range(rt.RT_K_COUNT)]
        dom_name = rt.KERNEL_CLASSES[dom]
        avg_ms = timed.kernel_ms[dom] / max(timed.kernel_launches[dom], 1)
        ab = algorithmic_bytes(args.workload)
        roofline = None
        if ab is not None and dom_name in ("trace_closest", "trace_shadow"):
            if dom_name == "trace_shadow":
                bytes_per_step = s0.shadow_rays * ab["shadow"]["bytes_per_ray"]
                launches_per_step = max(k_n[dom] // n_stat, 1)
            else:
                bytes_per_step = (s0.primary_rays * ab["primary"]["bytes_per_ray"]
                                  + s0.secondary_rays * ab["secondary"]["bytes_per_ray"])
                launches_per_step = max(k_n[dom] // n_stat, 1)
            # launches of one step share the step's rays; price the mean launch
            bytes_per_launch = bytes_per_step / launches_per_step
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.workload, dom_name) if not multi else None,
                        "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms,
                        "launches_timed": int(timed.kernel_launches[dom])}
            # In the timed region the kernels of FLIGHTS frames share the GPU, so one launch lasts
            # longer than it would alone.  The same kernel alone on the GPU (last warm-up frame):
            alone_ms = k_ms[dom] / max(k_n[dom], 1)
            alone = bytes_per_launch / (alone_ms * 1e-3) / 1e9
            roofline["alone_on_the_gpu"] = {"avg_launch_ms": alone_ms, "achieved": alone, "frac": alone / HBM_PEAK_GBS}
            if not multi:
                roofline["counters"] = measured_counters(args.workload, dom_name)  # rocprofv3 --pmc passes, profiles/
            if not multi:
                torch.cuda.set_stream(torch.cuda.default_stream())
                hbm = measured_hbm_gbs(torch)
                roofline["peak_measured"] = hbm  # copy / triad on this box, GB/s
                roofline["frac_of_measured"] = achieved / hbm
        out = {
            "metric": "rays/sec (all ray types), Stanford bunny 1000² spp9 d6, 1/2/4/8 GPUs",
            "value": value,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc, "traversal": args.mode,
                       "precision": "fp32 tree search, f64 hit refinement and shading, f32 colour (as the reference's f64 geometry / f32 colour)",
                       "tile": tile, "tiles": tiles_total,
                       "rays_per_step": total_rays / args.steps, "parallelism": f"tiles/{world}",
                       "part0_weight": part0_weight, "frames_in_flight": FLIGHTS, "hw_queues": int(hw_queues) if hw_queues else "runtime default",
                       "scene_upload_ms": upload_s * 1e3,
                       "kernel_split_from": "warm-up frames (fully instrumented); roofline duration from the timed frames"},
            "kernel_ms_per_step": {rt.KERNEL_CLASSES[k]: k_ms[k] / n_stat for k in range(len(rt.KERNEL_CLASSES))},
            "device_ms_per_step": sum(s.render_ms for s in stats) / n_stat,
            "roofline": roofline,
        }
        if frame_check:
            out["config"]["frame_check"] = frame_check
        if not multi and not args.no_cpu_baseline:
            # one frame of the benchmarked workload itself on one core (the reference's loop is serial,
            # render.rs:36-50): about 9 s for the bunny frame; the 4K workload is sampled at 1000x1000
            sample_px = 1000
            v1, dt1 = cpu_baseline(rt, scene_name, depth, sample_px, 3, 1)
            out["cpu_baseline"] = {"value": v1, "unit": "rays/s", "cores": 1, "kind": "port",
                                   "sample": f"one frame of the same scene at {sample_px}x{sample_px} rpp9 depth{depth}, "
                                             f"C++ restatement of the reference (not the Rust build), {dt1:.1f} s"}
            cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
            vN, dtN = cpu_baseline(rt, scene_name, depth, 1000, 3, cores)
            out["cpu_baseline_all_cores"] = {"value": vN, "unit": "rays/s", "cores": cores, "kind": "port",
                                             "sample": f"same scene at 1000x1000 rpp9, OpenMP over pixel columns, {dtN:.1f} s"}
        print(json.dumps(out), flush=True)

    for d in devs:
        d.close()
    for be in backends:
        be.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
