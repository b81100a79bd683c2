#!/usr/bin/env python3
"""Benchmark of the render hot path: rays/s (camera + mirror + shadow rays) of frames DELIVERED TO
HOST MEMORY, which is what the reference's render() returns (render.rs:7-23).

    python bench.py --gpus 1 --steps 200 --warmup 3
    python bench.py --gpus N ...          starts N rank processes itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step is one render of the whole frame with the scene already resident in HBM, ending with the
frame in page-locked host memory.  One GPU: rt_render_async on FLIGHTS backends in turn, so the copy
of frame k over the bus overlaps the kernels of frame k+1.  N > 1: the frame is cut into interleaved
tiles, rank r renders the tiles of part r into a compact device buffer, the library's RCCL exchange
(rt_comm_gather_frames: grouped ncclSend / ncclRecv) brings the parts of a group of frames to the
frames' roots -- frame k belongs to rank k mod N -- where a kernel restores the image layout and the
frame is copied to host memory.  Total work is fixed, so scaling is "strong".  Rank 0 prints one JSON
line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, datasheet
CLOCK_GHZ = 2.4        # MI355X peak engine clock
SIMDS = 1024           # 256 compute units x 4
CUS = 256

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

KERNEL_OF_CLASS = {"trace_shadow": ("k_shadow_shade_packet", "k_trace_packet<1,", "k_trace_fast<1>", "k_trace_shadow"),
                   "trace_closest": ("k_camera_shade_packet", "k_bounces_fast", "k_trace_packet<0,", "k_trace_fast<0>", "k_trace_closest"),
                   "shade": ("k_shade",)}


def algorithmic_bytes(workload):
    path = os.path.join(ROOT, "bench", "algorithmic_bytes.json")
    key = "bunny" if workload == "bunny4k" else workload
    try:
        with open(path) as f:
            return json.load(f)["workloads"][key]["per_class"]
    except Exception:
        return None


def pmc_profile(workload, kernel_class):
    """The committed counter profile of this workload (profiles/pmc_traffic.json, written by
    tools/summarize_pmc.py from separate rocprofv3 --pmc passes over `bench.py --flights 1`): figures
    of the kernel with the most device time, and the HBM bytes of a whole frame.  None when the
    profile is missing or was taken for another dominant kernel (kernel_class None: any kernel will do,
    for the whole-frame figures)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(workload)
        if entry and (kernel_class is None or entry.get("kernel", "").startswith(KERNEL_OF_CLASS.get(kernel_class, ()))):
            return entry
    except Exception:
        pass
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


def launch_ranks(n_gpus):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (torch.distributed.run,
    one per GPU) as children of this process, which has not touched the GPU, pass their output
    through (rank 0 prints the JSON line) and leave with their exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="bunny", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fast", choices=["fast", "reference"])
    ap.add_argument("--flights", type=int, default=0, help="frames in flight per GPU (0: 4, or 8 from five GPUs on)")
    ap.add_argument("--render-flags", type=int, default=0, help="RtRenderArgs.flags (RT_RENDER_*), for comparisons")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

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
    # RT_BENCH_MULTI_PATH=1 runs the N > 1 code (process group, RCCL communicator of the library, grouped
    # gather, un-tile) with a single rank, as a check of that path on a one-GPU machine; the figures
    # are then not the N = 1 benchmark.
    multi = world > 1 or os.environ.get("RT_BENCH_MULTI_PATH", "0") == "1"
    if multi and "RANK" not in os.environ:  # the one-rank check, started without a launcher
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU implementation")
    # RT_BENCH_BACKEND=gloo: rehearsal of the N > 1 bookkeeping with several ranks on ONE GPU; the tiles
    # then travel through host memory, because RCCL does not put two ranks on one device
    staged = os.environ.get("RT_BENCH_BACKEND", "nccl") == "gloo"
    if staged:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if multi:
        if staged:
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
    # output buffers of its own: the tail of one frame's kernels (a few long-running waves) and its
    # copy to the host then overlap the next frame's work instead of idling the GPU.
    FLIGHTS = args.flights or int(os.environ.get("RT_BENCH_FLIGHTS", "8" if many_parts else "4"))
    backends = [rt.Backend(local_rank) for _ in range(FLIGHTS)]
    t_up = time.perf_counter()
    devs = [be.upload(scene) for be in backends]
    upload_s = (time.perf_counter() - t_up) / FLIGHTS
    dev = devs[0]
    mode = rt.RT_MODE_FAST if args.mode == "fast" else rt.RT_MODE_REFERENCE_ORDER
    # Equal parts: the root of the gather rotates over the ranks from frame to frame (see flush below), so no rank has more to
    # do than another.  (RtRenderArgs.part0_weight is there for a caller that keeps one fixed root.)
    part0_weight = int(os.environ.get("RT_BENCH_PART0_WEIGHT", "0"))
    rargs = dev.render_args(n, mode=mode, tile=(tile, tile), part=(rank, world), part0_weight=part0_weight,
                            flags=args.render_flags)
    # The parts of a frame do not cost the same (the bunny frame cut in eight: 0.148-0.162 ms, the dearest
    # part holds most of a silhouette), and a fixed assignment makes every frame wait for the same GPU.
    # So the parts ROTATE over the ranks from frame to frame: rank r renders part (r + k) mod N of frame k,
    # and over N frames every rank has rendered every part once.  A backend caches the tile list of its
    # part, so each flight keeps one part: flight b renders the frames k = b (mod FLIGHTS), which all have
    # the same part on this rank when N divides FLIGHTS; otherwise the parts stay where they are.
    rotate_parts = world > 1 and FLIGHTS % world == 0 and part0_weight == 0
    flight_args = [dev.render_args(n, mode=mode, tile=(tile, tile), part=((rank + b) % world if rotate_parts else rank, world),
                                   part0_weight=part0_weight, flags=args.render_flags) for b in range(FLIGHTS)]
    tiles_total, tiles_owned, stride = dev.partition_info(rargs)
    tile_floats = tile * tile * 3
    fb_floats = W * H * 3

    # Streams of our own: the backend treats a null stream handle as "use the backend's private
    # stream", which would not be ordered with torch's copies.
    streams = [torch.cuda.Stream() for _ in range(FLIGHTS)]
    torch.cuda.set_stream(streams[0])
    counter = [0]
    comm = None

    if not multi:
        host = [rt.HostBuffer((H, W, 3)) for _ in range(FLIGHTS)]
        fbs = [torch.zeros(fb_floats, dtype=torch.float32, device="cuda") for _ in range(FLIGHTS)]
        delivered = host[0].array

        def step(want_stats):
            """One frame into page-locked host memory (the benchmarked path): rt_render_async on the next
            backend, after waiting for that backend's previous frame; with statistics, a blocking rt_render."""
            b = counter[0] % FLIGHTS
            counter[0] += 1
            if want_stats:
                return devs[b].render(n, mode=mode, tile=(tile, tile), flags=args.render_flags, out=host[b].array)[2]
            backends[b].sync()
            devs[b].render_async(rargs, host[b])
            return None

        def step_device():
            """The same frame left in device memory (no copy): rt_render_device on streams of our own."""
            b = counter[0] % FLIGHTS
            counter[0] += 1
            devs[b].render_device(rargs, fbs[b].data_ptr(), stream=streams[b].cuda_stream)

        def drain():
            for be in backends:
                be.sync()
    else:
        # GROUP consecutive frames are exchanged together, and THE ROOT ROTATES FROM FRAME TO FRAME: frame k
        # of the run belongs to rank k mod N.  Every rank renders its part of each frame of the group
        # into one buffer (ordered by root, so that what goes to one root is one message), ONE call of
        # rt_comm_gather_frames posts a send per (rank, root) pair and the receives, and every rank
        # restores the image layout of the frames it is root of on a side stream and sends each to
        # page-locked host memory.  Two groups alternate, so the exchange, un-tiling and copies of one
        # overlap the renders of the other.
        # Why groups: measured on one MI355X with a one-rank communicator (the exchange degenerates to a
        # local copy, so this is a lower bound), per 1/8 frame: render alone 0.169 ms; with a gather and an
        # un-tile per frame 0.245 ms (the stream hand-overs around a collective stall the pipeline); per
        # group of 8 frames 0.191 ms.
        # Why rotating roots: a frame is 12 MB; one GPU's PCIe link carries about 4000 such frames a second
        # while eight GPUs render 6000.  With the frames dealt round robin, receiving, un-tiling and the
        # copy home are shared by all GPUs and links, the xGMI links carry tiles both ways, every rank does
        # the same work, the parts can be equal, and when a run ends each GPU has one frame of the last
        # group left to send home instead of one GPU all of them.
        GROUP = 8
        SLOTS = -(-GROUP // world)       # frames of a group per root, at most
        part_floats = stride * tile_floats
        part = [torch.zeros(world * SLOTS * part_floats, dtype=torch.float32, device="cuda") for _ in range(2)]
        gathered = [torch.zeros(world * SLOTS * part_floats, dtype=torch.float32, device="cuda") for _ in range(2)]
        fb = [torch.zeros(fb_floats, dtype=torch.float32, device="cuda") for _ in range(2)]
        host_fb = [torch.zeros(fb_floats, dtype=torch.float32).pin_memory() for _ in range(2)]
        joiner = torch.cuda.Stream()     # waits for the frames of a group, then issues their exchange
        side = torch.cuda.Stream()       # as a root: waits for the exchange, restores the image layout, copies to the host
        sent = [None, None]              # event: the exchange that read part[g] / wrote gathered[g] is done
        untiled = [None, None]           # event: gathered[g] has been consumed
        last_delivered = [None]          # the host buffer that holds the newest frame this rank was root of
        frames_sent = [0]                # frames handed to an exchange so far: the next group's first root follows from it
        frames_rooted = [0]
        exchange_by = "gloo" if staged else "library"
        if not staged:
            # The library's own RCCL communicator: rank 0 makes the id, the launcher's process group carries
            # it.  Should that communicator not come up on some node (a second copy of librccl in the process,
            # a bootstrap problem), every rank falls back to the SAME exchange posted through torch.distributed
            # -- one send per (rank, root) pair, one receive each, in one batch -- and the line says so: a
            # scaling figure with a flag is worth more than none.  RT_BENCH_EXCHANGE=torch forces the fallback.
            failed = 0
            try:
                if os.environ.get("RT_BENCH_EXCHANGE", "library") == "torch":
                    raise rt.RtError(-4, "RT_BENCH_EXCHANGE=torch")
                ids = [rt.Comm.make_id() if rank == 0 else None]
            except rt.RtError as e:
                ids, failed = [None], 1
                print(f"[bench rank {rank}] rt_comm_id failed: {e}", file=sys.stderr)
            dist.broadcast_object_list(ids, src=0)
            if ids[0] is not None and not failed:
                try:
                    comm = rt.Comm(backends[0], ids[0], rank, world)
                except rt.RtError as e:
                    failed = 1
                    print(f"[bench rank {rank}] rt_comm_create failed: {e}", file=sys.stderr)
            else:
                failed = 1
            agreed = torch.tensor([failed], dtype=torch.int32, device="cuda")
            dist.all_reduce(agreed, op=dist.ReduceOp.MAX)
            if int(agreed.item()):
                if comm is not None:
                    comm.close()
                comm = None
                exchange_by = "torch"

        def flush(g, count):
            """Exchanges the first `count` frames of group g; this rank un-tiles and copies out those it is root of."""
            first_root = frames_sent[0] % world
            frames_sent[0] += count
            for b in range(min(count, FLIGHTS)):
                joiner.wait_stream(streams[b])
            with torch.cuda.stream(joiner):
                if untiled[g] is not None:
                    joiner.wait_event(untiled[g])
                if staged:  # rehearsal on one GPU: the same blocks through host memory, one gather per root
                    for p in range(min(count, world)):
                        root_rank = (first_root + p) % world
                        k = -(-(count - p) // world)
                        cpu = part[g][p * SLOTS * part_floats: (p * SLOTS + k) * part_floats].cpu()
                        bufs = [torch.empty_like(cpu) for _ in range(world)] if rank == root_rank else None
                        dist.gather(cpu, bufs, dst=root_rank)
                        if rank == root_rank:
                            for r in range(world):
                                gathered[g][r * SLOTS * part_floats: (r * SLOTS + k) * part_floats].copy_(bufs[r])
                elif comm is not None:
                    comm.gather_frames(part[g].data_ptr(), part_floats, count, first_root, SLOTS, gathered[g].data_ptr(),
                                       stream=joiner.cuda_stream)
                else:  # the fallback: rt_comm_gather_frames' sends and receives through torch.distributed
                    ops = []
                    for p in range(min(count, world)):
                        k = -(-(count - p) // world)
                        ops.append(dist.P2POp(dist.isend, part[g][p * SLOTS * part_floats: (p * SLOTS + k) * part_floats],
                                              (first_root + p) % world))
                    my_p = (rank - first_root) % world
                    if my_p < count:
                        k = -(-(count - my_p) // world)
                        for r in range(world):
                            ops.append(dist.P2POp(dist.irecv, gathered[g][r * SLOTS * part_floats: (r * SLOTS + k) * part_floats], r))
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()  # (orders the current stream behind the transfer; the host does not block)
                sent[g] = joiner.record_event()
            mine = [f for f in range(count) if rt.Comm.frame_blocks(world, SLOTS, f, first_root)[0] == rank]
            if mine:
                with torch.cuda.stream(side):
                    side.wait_event(sent[g])
                    for f in mine:
                        slot = rt.Comm.frame_blocks(world, SLOTS, f, first_root)[2]
                        o = frames_rooted[0] & 1
                        frames_rooted[0] += 1
                        # rank r's part of the frame lies at block r * SLOTS + slot, and it is part (r + f) mod N
                        # when the parts rotate (frame f of a group ran on flight f mod FLIGHTS)
                        ua = dev.render_args(n, mode=mode, tile=(tile, tile), part=(rank, world), part0_weight=part0_weight,
                                             part_rotation=(f % FLIGHTS) % world if rotate_parts else 0)
                        dev.untile_device(ua, gathered[g].data_ptr() + slot * part_floats * 4, fb[o].data_ptr(),
                                          stream=side.cuda_stream, part_pitch_floats=SLOTS * part_floats)
                        host_fb[o].copy_(fb[o], non_blocking=True)
                        last_delivered[0] = host_fb[o]
                    untiled[g] = side.record_event()

        def step(want_stats):
            k = counter[0]
            counter[0] += 1
            g, f, b = (k // GROUP) % 2, k % GROUP, k % FLIGHTS
            block = rt.Comm.frame_blocks(world, SLOTS, f)[1]
            with torch.cuda.stream(streams[b]):
                if sent[g] is not None:
                    streams[b].wait_event(sent[g])  # the exchange of two groups ago has released part[g]
                st = devs[b].render_device(flight_args[b], part[g].data_ptr() + block * part_floats * 4,
                                           stream=streams[b].cuda_stream, want_stats=want_stats)
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
            for be in backends:
                be.sync()

    def fence():
        drain()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        fence()
        return time.perf_counter() - t0

    # Warm-up frames are rendered with the backend's full statistics (an event pair around every
    # kernel, counters read back, the host waits for each frame, so every kernel runs alone on the
    # GPU): they give the ray counts, which are the same for every frame, and the split of a frame
    # over the kernel classes.  Every flight twice at least: a backend's first frame touches its
    # buffers for the first time.
    warm_stats = [step(True) for _ in range(max(args.warmup, 2 * FLIGHTS))]
    fence()
    s0 = warm_stats[-1]  # the last warm-up frame: caches and clocks have settled
    k_warm = [s0.kernel_ms[k] for k in range(rt.RT_K_COUNT)]
    dom = max(range(len(rt.KERNEL_CLASSES)), key=lambda k: k_warm[k])
    dom_name = rt.KERNEL_CLASSES[dom]
    rays_per_frame_local = float(s0.total_rays())

    # untimed frames through the path that is timed next: its buffers are touched for the first time,
    # the runtime grows its pools of signals and command buffers, the copy engines and clocks come up.
    # Measured: the enqueue cost per frame falls from 1.8 ms to 0.03 ms over the first 24 frames.
    for _ in range(8 * FLIGHTS if not multi else 4 * GROUP):
        step(False)
    # ---- the timed region: EXACTLY args.steps frames, each delivered to host memory
    elapsed = timed(lambda: step(False), args.steps)

    extra = {}
    solo_ms, solo_launches = None, 0
    if not multi:
        # the same frames left in device memory, FLIGHTS in flight (no copy to the host) ...
        for _ in range(FLIGHTS):
            step_device()
        el_dev = timed(step_device, args.steps)
        extra["value_device_resident"] = rays_per_frame_local * args.steps / el_dev
        extra["ms_per_step_device_resident"] = el_dev / args.steps * 1e3
        # ... and one frame at a time through the blocking entry point, rt_render into page-locked memory
        single = max(20, args.steps // 4)
        t0 = time.perf_counter()
        for _ in range(single):
            devs[0].render(n, mode=mode, tile=(tile, tile), flags=args.render_flags, want_stats=False, out=host[0].array)
        el_single = time.perf_counter() - t0
        extra["value_single_frame"] = rays_per_frame_local * single / el_single
        extra["ms_per_step_single_frame"] = el_single / single * 1e3
        # Duration of the dominant kernel for the roofline: frames one at a time (the kernel has the GPU
        # to itself, as in the profiles under profiles/), an event pair around that kernel class only.
        rargs.timed_classes = 1 << dom
        for _ in range(single):
            devs[0].render_device(rargs, fbs[0].data_ptr(), stream=streams[0].cuda_stream)
            backends[0].sync()
        t = backends[0].collect_timings()
        rargs.timed_classes = 0
        solo_ms = t.kernel_ms[dom] / max(t.kernel_launches[dom], 1)
        solo_launches = int(t.kernel_launches[dom])

    # Outside the timed region: the gathered frame must be bit-identical with the same frame rendered
    # whole (static tile map, ordered per-pixel sums).  Every rank that was a root checks the newest
    # frame it delivered against its own single-GPU render of the frame.
    frame_check = None
    if multi and world == 1:
        frame_check = "not applicable: a single part is rendered in image layout, not tile by tile"
    elif multi:
        bad = 0
        if last_delivered[0] is not None:
            whole = torch.zeros(fb_floats, dtype=torch.float32, device="cuda")
            dev.render_device(dev.render_args(n, mode=mode, tile=(tile, tile), flags=args.render_flags), whole.data_ptr(),
                              stream=streams[0].cuda_stream)
            backends[0].sync()
            torch.cuda.synchronize()
            bad = int((whole.cpu() != last_delivered[0]).sum())
        checked = torch.tensor([bad, 1 if last_delivered[0] is not None else 0], dtype=torch.int64)
        if not staged:
            checked = checked.cuda()
        dist.all_reduce(checked, op=dist.ReduceOp.SUM)
        if int(checked[0]) != 0:
            raise SystemExit(f"gathered frames differ from the single-GPU frame in {int(checked[0])} values")
        frame_check = (f"newest gathered frame in host memory bit-identical with a single-GPU render of the frame, "
                       f"on each of the {int(checked[1])} ranks that served as root")

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    rays = torch.tensor([rays_per_frame_local * args.steps], dtype=torch.float64, device="cuda")
    if multi:
        if staged:
            t, rays = t.cpu(), rays.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays = float(rays.item())

    if rank == 0:
        value = total_rays / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = [s0.kernel_ms[k] for k in range(rt.RT_K_COUNT)]
        k_n = [s0.kernel_launches[k] for k in range(rt.RT_K_COUNT)]
        ab = algorithmic_bytes(args.workload)
        roofline = roofline_issue = None
        hbm_true = None
        if solo_ms is None:  # N > 1: the instrumented warm-up frame's figure (its kernels also ran alone)
            solo_ms = k_ms[dom] / max(k_n[dom], 1)
        roof_note = None
        if dom_name not in ("trace_closest", "trace_shadow"):
            # The stage with the most device time is not a traversal (a mirror scene: the f64 shade stage, one
            # launch per bounce, bound by the latency of its dependent loads; it has no byte yardstick).  The
            # HBM object is then given for the traversal class with the most time, and says so.
            roof_note = f"the class with the most device time is '{dom_name}'; figures are for the dominant traversal class"
            dom = max((k for k in range(len(rt.KERNEL_CLASSES)) if rt.KERNEL_CLASSES[k] in ("trace_closest", "trace_shadow")),
                      key=lambda k: k_ms[k])
            dom_name = rt.KERNEL_CLASSES[dom]
            solo_ms = k_ms[dom] / max(k_n[dom], 1)
            solo_launches = 0
        prof = pmc_profile(args.workload, dom_name) if not multi else None
        prof_frame = pmc_profile(args.workload, None) if not multi else None
        if ab is not None and dom_name in ("trace_closest", "trace_shadow"):
            if dom_name == "trace_shadow":
                bytes_per_step = s0.shadow_rays * ab["shadow"]["bytes_per_ray"]
            else:
                bytes_per_step = (s0.primary_rays * ab["primary"]["bytes_per_ray"]
                                  + s0.secondary_rays * ab["secondary"]["bytes_per_ray"])
            launches_per_step = max(k_n[dom], 1)
            bytes_per_launch = bytes_per_step / launches_per_step  # launches of one step share the step's rays
            achieved = bytes_per_launch / (solo_ms * 1e-3) / 1e9
            roofline = {
                "bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": prof["bytes"] if prof else None,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": solo_ms,
                "launches_timed": solo_launches or int(k_n[dom]),
                "duration_from": "HIP events around this kernel class on its launch stream, frames rendered one at a "
                                 "time after the timed region (the kernel alone on the GPU, as in profiles/*_solo)",
                "frac_above_1_is_expected": "algorithmic bytes are the REFERENCE's traversal (oracle node and primitive "
                                            "counts x record sizes); this kernel visits about half the nodes in trees "
                                            "of its own, fetches each once per 64 rays and is served from L2 / MALL. "
                                            "HBM does not bound it: see roofline_issue and hbm_true_frac",
            }
            if prof:
                roofline["counters"] = {k: prof.get(k) for k in ("hbm_gbs", "l2_hit_rate", "valu_lane_utilisation")}
                roofline["traffic_frac"] = prof["bytes"] / (solo_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            if roof_note:
                roofline["note"] = roof_note
        if prof and prof.get("valu_wave_insts"):
            # What does bound the kernel: instruction issue.  A wave64 vector instruction occupies its SIMD
            # for 4 cycles; the scalar unit of a compute unit issues one instruction per cycle.
            valu_ms = prof["valu_wave_insts"] * 4.0 / SIMDS / (CLOCK_GHZ * 1e9) * 1e3
            scalar_ms = prof["scalar_insts"] / CUS / (CLOCK_GHZ * 1e9) * 1e3
            bound_ms = max(valu_ms, scalar_ms)
            roofline_issue = {"kernel": prof["kernel"], "valu_wave_insts": prof["valu_wave_insts"], "scalar_insts": prof["scalar_insts"],
                              "clock_ghz": CLOCK_GHZ, "valu_bound_ms": valu_ms, "scalar_bound_ms": scalar_ms, "bound_ms": bound_ms,
                              "kernel_ms": solo_ms, "frac": bound_ms / solo_ms,
                              "from": "instruction counts per launch: profiles/pmc_traffic.json (rocprofv3 --pmc); kernel_ms: live, as roofline.avg_launch_ms",
                              # the duration of the same kernel in the kernel trace taken with the counter passes: when it
                              # is far from the live one, the committed counts belong to another version of the kernel
                              "kernel_ms_in_profile": prof["avg_us"] * 1e-3 if prof.get("avg_us") else None,
                              "profile_matches_this_build": (abs(prof["avg_us"] * 1e-3 - solo_ms) <= 0.1 * solo_ms) if prof.get("avg_us") else None}
        if prof_frame and prof_frame.get("frame_hbm_bytes"):
            hbm_true = prof_frame["frame_hbm_bytes"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
        if roofline is not None and not multi:
            torch.cuda.set_stream(torch.cuda.default_stream())
            hbm = measured_hbm_gbs(torch)
            roofline["peak_measured"] = hbm  # copy / triad on this box, GB/s
            roofline["frac_of_measured"] = roofline["achieved"] / hbm
        out = {
            "metric": "rays/sec (all ray types), Stanford bunny 1000² spp9 d6, 1/2/4/8 GPUs",
            "value": value,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc, "traversal": args.mode,
                       "value_is": "frames delivered to page-locked host memory (what render() returns, render.rs:7-23), "
                                   f"{FLIGHTS} frames in flight",
                       "precision": ("fp32 tree search, f64 hit refinement (point, normal, mirror direction, self-hit tests), "
                                     + ("light sums in f32 on the f32-rounded vectors (fast mode), " if args.mode == "fast" else "f64 light sums, ")
                                     + "f32 colour (the reference: f64 geometry, f32 colour)"),
                       "tile": tile, "tiles": tiles_total,
                       "rays_per_step": total_rays / args.steps, "parallelism": f"tiles/{world}",
                       "part0_weight": part0_weight, "frames_in_flight": FLIGHTS, "hw_queues": int(hw_queues) if hw_queues else "runtime default",
                       "rccl_ranks": dist.get_world_size() if multi else 1,
                       "gather": ("none (one GPU)" if not multi else
                                  "host memory over gloo (one-GPU rehearsal)" if staged else
                                  "FALLBACK: the library's communicator did not come up; the same sends and receives "
                                  "through torch.distributed (batch_isend_irecv)" if exchange_by == "torch" else
                                  f"rt_comm_gather_frames: groups of {GROUP} frames, one grouped ncclSend / ncclRecv exchange per group "
                                  "on the library's RCCL communicator, frame k gathered to rank k mod N"),
                       "parts": ("one (whole frames)" if world == 1 else
                                 "rotating: rank r renders part (r + k) mod N of frame k" if rotate_parts else
                                 "fixed: rank r renders part r"),
                       "render_flags": args.render_flags,
                       "scene_upload_ms": upload_s * 1e3,
                       "kernel_split_from": "last warm-up frame (fully instrumented, every kernel alone on the GPU)"},
            "kernel_ms_per_step": {rt.KERNEL_CLASSES[k]: k_ms[k] for k in range(len(rt.KERNEL_CLASSES))},
            "device_ms_per_step": s0.render_ms,
            "roofline": roofline,
            "roofline_issue": roofline_issue,
            "hbm_true_frac": hbm_true,
        }
        out.update(extra)
        if frame_check:
            out["config"]["frame_check"] = frame_check
        if not multi:
            # the delivered frame is a frame: compare a few pixels of it with the device copy's statistics
            out["config"]["delivered_frame_checksum"] = float(delivered.astype("float64").sum())
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

    if comm is not None:
        comm.close()
    if not multi:
        for h in host:
            h.close()
    for d in devs:
        d.close()
    for be in backends:
        be.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
