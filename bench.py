#!/usr/bin/env python3
"""Benchmark of the render hot path: rays/s (camera + mirror + shadow rays).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step is one render of the whole frame with the scene already resident in HBM.
With N > 1 the frame is cut into interleaved tiles, rank r renders the tiles of
part r into a compact device buffer, one RCCL gather brings the buffers to rank 0
and a kernel there restores the image layout; total work is fixed, so scaling is
"strong".  Rank 0 prints one JSON line.
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


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from a committed PMC profile, or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(workload)
        return entry["bytes"] if entry else None
    except Exception:
        return None


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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="bunny", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fast", choices=["fast", "reference"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    rt = importlib.import_module("rust-raytracer_amd")
    gen = importlib.import_module("scenes.gen_scenes")
    if rank == 0:
        gen.ensure_models()
    if world > 1:
        dist.barrier()

    scene_name, W, H, n, depth, tile, desc = WORKLOADS[args.workload]
    scene = rt.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".json"), image_width=W, image_height=H,
                                recursion_depth=depth)
    backend = rt.Backend(local_rank)
    t_up = time.perf_counter()
    dev = backend.upload(scene)
    upload_s = time.perf_counter() - t_up
    mode = rt.RT_MODE_FAST if args.mode == "fast" else rt.RT_MODE_REFERENCE_ORDER
    rargs = dev.render_args(n, mode=mode, tile=(tile, tile), part=(rank, world))
    tiles_total, tiles_owned, stride = dev.partition_info(rargs)
    tile_floats = tile * tile * 3

    stream = torch.cuda.current_stream().cuda_stream
    if world > 1:
        part_buf = torch.zeros(stride * tile_floats, dtype=torch.float32, device="cuda")
        gathered = torch.zeros(world * stride * tile_floats, dtype=torch.float32, device="cuda") if rank == 0 else None
        fb = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda") if rank == 0 else None
    else:
        part_buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
        gathered, fb = None, part_buf

    def step(want_stats):
        st = dev.render_device(rargs, part_buf.data_ptr(), stream=stream, want_stats=want_stats)
        if world > 1:
            dist.gather(part_buf, list(gathered.chunk(world)) if rank == 0 else None, dst=0)
            if rank == 0:
                dev.untile_device(rargs, gathered.data_ptr(), fb.data_ptr(), stream=stream)
        return st

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    stats = [step(True) for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    rays = torch.tensor([float(sum(s.total_rays() for s in stats))], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays = float(rays.item())

    if rank == 0:
        value = total_rays / elapsed
        s0 = stats[0]
        # dominant kernel = the class with the most device time on this rank
        k_ms = [sum(s.kernel_ms[k] for s in stats) for k in range(rt.RT_K_COUNT)]
        k_n = [sum(s.kernel_launches[k] for s in stats) for k in range(rt.RT_K_COUNT)]
        dom = max(range(len(rt.KERNEL_CLASSES)), key=lambda k: k_ms[k])
        dom_name = rt.KERNEL_CLASSES[dom]
        avg_ms = k_ms[dom] / max(k_n[dom], 1)
        ab = algorithmic_bytes(args.workload)
        roofline = None
        if ab is not None and dom_name in ("trace_closest", "trace_shadow"):
            if dom_name == "trace_shadow":
                bytes_per_step = s0.shadow_rays * ab["shadow"]["bytes_per_ray"]
                launches_per_step = max(k_n[dom] // args.steps, 1)
            else:
                bytes_per_step = (s0.primary_rays * ab["primary"]["bytes_per_ray"]
                                  + s0.secondary_rays * ab["secondary"]["bytes_per_ray"])
                launches_per_step = max(k_n[dom] // args.steps, 1)
            # launches of one step share the step's rays; price the mean launch
            bytes_per_launch = bytes_per_step / launches_per_step
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.workload),
                        "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms}
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
            "config": {"workload": desc, "traversal": args.mode, "tile": tile, "tiles": tiles_total,
                       "rays_per_step": total_rays / args.steps, "parallelism": f"tiles/{world}",
                       "scene_upload_ms": upload_s * 1e3},
            "kernel_ms_per_step": {rt.KERNEL_CLASSES[k]: k_ms[k] / args.steps for k in range(len(rt.KERNEL_CLASSES))},
            "device_ms_per_step": sum(s.render_ms for s in stats) / args.steps,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            sample_px = 700 if args.workload.startswith("bunny") else 500
            v1, dt1 = cpu_baseline(rt, scene_name, depth, sample_px, 3, 1)
            out["cpu_baseline"] = {"value": v1, "unit": "rays/s", "cores": 1, "kind": "port",
                                   "sample": f"same scene at {sample_px}x{sample_px} rpp9 depth{depth}, "
                                             f"C++ restatement of the reference (not the Rust build), {dt1:.1f} s"}
            cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
            vN, dtN = cpu_baseline(rt, scene_name, depth, 1000, 3, cores)
            out["cpu_baseline_all_cores"] = {"value": vN, "unit": "rays/s", "cores": cores, "kind": "port",
                                             "sample": f"same scene at 1000x1000 rpp9, OpenMP over pixel columns, {dtN:.1f} s"}
        print(json.dumps(out), flush=True)

    dev.close()
    backend.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
