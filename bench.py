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
tiles, rank r renders the tiles of part (r + k) mod N of frame k into a compact device buffer, the
library's RCCL exchange (rt_comm_gather_frames: grouped ncclSend / ncclRecv) brings the parts of a
group of frames to the frames' roots -- frame k belongs to rank k mod N -- where a kernel restores the
image layout and the frame is copied to host memory.  Total work is fixed, so scaling is "strong".
Rank 0 prints one JSON line.
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
L2_PEAK_GBS = 34500.0  # the eight XCDs' L2s together, measured streaming rate
CLOCK_GHZ = 2.4        # MI355X peak engine clock
SIMDS = 1024           # 256 compute units x 4
CUS = 256

WORKLOADS = {
    # name: (scene, width, height, sqrt_rpp, depth, tile, description)
    "bunny": ("bunny", 1000, 1000, 3, 6, 40, "bunny 1000x1000 rpp9 depth6: {mesh} + ground box, 2 point lights"),
    "bunny4k": ("bunny", 3840, 2160, 8, 6, 32, "bunny 3840x2160 rpp64 depth6: {mesh} + ground box, 2 point lights"),
    "spheres_1K": ("spheres_1K", 1000, 1000, 3, 6, 40, "spheres_1K 1000x1000 rpp9 depth6: 10x10x10 spheres, 4 point lights"),
    "cornell_room": ("cornell_room", 1000, 1000, 3, 6, 40, "cornell_room 1000x1000 rpp9 depth6: 6 boxes + 6 mirror spheres"),
    "spheres_and_triangles": ("spheres_and_triangles", 1000, 1000, 3, 6, 40, "spheres_and_triangles 1000x1000 rpp9 depth6"),
    "simple_sphere_scene": ("simple_sphere_scene", 1000, 1000, 3, 6, 40, "simple_sphere_scene 1000x1000 rpp9 depth6"),
}

KERNEL_OF_CLASS = {"trace_shadow": ("k_shadow_shade_packet", "k_trace_packet<1,", "k_trace_fast<1>", "k_trace_shadow"),
                   "trace_closest": ("k_camera_shade_packet", "k_bounces_fast", "k_trace_packet<0,", "k_trace_fast<0>", "k_trace_closest"),
                   "shade": ("k_shade",)}

GROUP = 8  # frames per exchange of the N > 1 stream


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


def cpu_baseline(rt, scene_file, depth, sample_px, n, threads):
    """The CPU restatement of the reference on one frame of the scene; (rays/s, seconds, counters)."""
    import oracle_lib

    sc = rt.parse_scene_file(scene_file, image_width=sample_px, image_height=sample_px, recursion_depth=depth)
    t0 = time.perf_counter()
    _, _, counters = oracle_lib.render(sc, n, want_ids=False, threads=threads)
    dt = time.perf_counter() - t0
    return sum(counters["rays"]) / dt, dt, counters


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


class Frames:
    """A workload resident on this rank's GPU: FLIGHTS backends (each the ray queues of one frame in flight),
    the scene uploaded through each, and a stream per flight."""

    def __init__(self, ctx, workload, flights):
        rt, torch = ctx.rt, ctx.torch
        self.ctx = ctx
        self.workload = workload
        scene_name, self.W, self.H, self.n, self.depth, self.tile, desc = WORKLOADS[workload]
        if scene_name == "bunny":
            scene_name, mesh = ctx.bunny
            desc = desc.format(mesh=mesh)
        self.scene_name, self.desc = scene_name, desc
        self.scene_file = os.path.join(ROOT, "scenes", scene_name + ".json")
        scene = rt.parse_scene_file(self.scene_file, image_width=self.W, image_height=self.H, recursion_depth=self.depth)
        self.flights = flights
        self.backends = [rt.Backend(ctx.local_rank) for _ in range(flights)]
        t_up = time.perf_counter()
        if os.environ.get("RT_BENCH_SHARE_SCENE", "1") == "1":  # one copy of the scene for all frames in flight
            first = self.backends[0].upload(scene)
            self.devs = [first] + [first.on(be) for be in self.backends[1:]]
        else:
            self.devs = [be.upload(scene) for be in self.backends]
        self.upload_s = (time.perf_counter() - t_up) / len({int(d.handle.value) for d in self.devs})
        self.dev = self.devs[0]
        # Streams of our own: the backend treats a null stream handle as "use the backend's private stream", which
        # would not be ordered with torch's copies.
        self.streams = [torch.cuda.Stream() for _ in range(flights)]
        self.fb_floats = self.W * self.H * 3
        self.counter = 0

    def args(self, **kw):
        c = self.ctx
        return self.dev.render_args(self.n, mode=c.mode, tile=(self.tile, self.tile), flags=c.render_flags, **kw)

    def fence(self):
        self.drain()
        if self.ctx.multi:
            self.ctx.dist.barrier()
        self.ctx.torch.cuda.synchronize()

    def timed(self, fn, steps):
        """Wall-clock of `steps` calls of fn: all ranks start together behind a barrier with idle GPUs, and each stops
        its clock when ITS last frame is where it belongs (the caller takes the maximum over the ranks, which is
        the moment the last frame of the job was delivered; a closing barrier inside the clock would add its own
        latency to a region of a few milliseconds)."""
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.drain()
        self.ctx.torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if self.ctx.multi:
            self.ctx.dist.barrier()
        return elapsed

    def close(self):
        for d in self.devs:
            d.close()
        for be in self.backends:
            be.close()


class OneGpu(Frames):
    def __init__(self, ctx, workload, flights):
        super().__init__(ctx, workload, flights)
        rt, torch = ctx.rt, ctx.torch
        self.host = [rt.HostBuffer((self.H, self.W, 3)) for _ in range(flights)]
        self.fbs = [torch.zeros(self.fb_floats, dtype=torch.float32, device="cuda") for _ in range(flights)]
        self.rargs = self.args()
        self.delivered = self.host[0].array

    def step(self, want_stats=False):
        """One frame into page-locked host memory (the benchmarked path): rt_render_async on the next backend,
        after waiting for that backend's previous frame; with statistics, a blocking rt_render."""
        b = self.counter % self.flights
        self.counter += 1
        if want_stats:
            c = self.ctx
            return self.devs[b].render(self.n, mode=c.mode, tile=(self.tile, self.tile), flags=c.render_flags, out=self.host[b].array)[2]
        self.backends[b].sync()
        self.devs[b].render_async(self.rargs, self.host[b])
        return None

    def step_device(self):
        """The same frame left in device memory (no copy): rt_render_device on streams of our own."""
        b = self.counter % self.flights
        self.counter += 1
        self.devs[b].render_device(self.rargs, self.fbs[b].data_ptr(), stream=self.streams[b].cuda_stream)

    def drain(self):
        for be in self.backends:
            be.sync()

    def rays_per_frame(self, stats):
        return float(stats.total_rays())

    def close(self):
        for h in self.host:
            h.close()
        super().close()


class ManyGpus(Frames):
    """The stream of frames over N ranks.  GROUP consecutive frames are exchanged together, and THE ROOT ROTATES FROM
    FRAME TO FRAME: frame k of the run belongs to rank k mod N.  Every rank renders its part of each frame of the
    group into one buffer (ordered by root, so that what goes to one root is one message), ONE call of
    rt_comm_gather_frames posts a send per (rank, root) pair and the receives, and every rank restores the image
    layout of the frames it is root of on a side stream and sends each to page-locked host memory.  Two groups
    alternate, so the exchange, un-tiling and copies of one overlap the renders of the other.

    Why groups: measured on one MI355X with a one-rank communicator (the exchange degenerates to a local copy, so
    this is a lower bound), per 1/8 frame: render alone 0.169 ms; with a gather and an un-tile per frame 0.245 ms (the
    stream hand-overs around a collective stall the pipeline); per group of 8 frames 0.191 ms.
    Why rotating roots: a frame is 12 MB; one GPU's PCIe link carries about 4000 such frames a second while eight
    GPUs render 6000.  With the frames dealt round robin, receiving, un-tiling and the copy home are shared by all
    GPUs and links, the xGMI links carry tiles both ways, every rank does the same work, the parts can be equal,
    and when a run ends each GPU has one frame of the last group left to send home instead of one GPU all of them.
    Why rotating parts: the parts of a frame do not cost the same (the dearest holds most of a silhouette), and a
    fixed assignment makes every frame wait for the same GPU: rank r renders part (r + k) mod N of frame k.
    Why several frames per set of launches beyond four ranks (SPAN): a kernel lasts as long as its slowest wave
    (up to 0.14 ms in the shadow kernel) whatever the size of its grid, so the three launches of a 1/8 frame keep
    the GPU busy for 0.21 ms to do 0.083 ms of work, and eight such frames in flight overlap to 0.097 ms per
    frame.  What a rank owes SPAN consecutive frames -- the parts (r + k) ... (r + k + SPAN - 1) mod N, neighbouring
    tiles -- it renders with one set of launches (RtRenderArgs.part_span), each part into its own frame's block:
    two at a time 0.087 ms per part.  A step is still one whole frame, delivered as before."""

    def __init__(self, ctx, workload, flights):
        super().__init__(ctx, workload, flights)
        rt, torch, world, rank = ctx.rt, ctx.torch, ctx.world, ctx.rank
        self.slots = -(-GROUP // world)  # frames of a group per root, at most
        # A backend caches the tile list of the parts it renders, so each flight keeps its parts: flight b renders the
        # frames k with (k div SPAN) = b (mod FLIGHTS), which start at the same part on this rank when N divides
        # SPAN * FLIGHTS; otherwise the parts stay where they are, one per set of launches.
        # The blocks of consecutive frames lie `slots` blocks apart in the send buffer ([root position][slot]); a span
        # never straddles the end of a round of roots because it divides the number of ranks.
        span = int(os.environ.get("RT_BENCH_SPAN", "0")) or (max(2, world // 4) if world >= 4 else 1)
        if (GROUP % span != 0 or world % span != 0 or (flights * span) % world != 0  # (a span's parts differ, so they rotate)
                or ctx.part0_weight != 0):
            span = 1
        self.span = span
        self.rotate = world > 1 and (flights * span) % world == 0 and ctx.part0_weight == 0
        self.tiles_total, self.tiles_owned, self.stride = self.dev.partition_info(self.args(part=(rank, world), part0_weight=ctx.part0_weight))
        self.part_floats = self.stride * self.tile * self.tile * 3
        self.flight_args = [self.args(part=((rank + span * b) % world if self.rotate else rank, world), part0_weight=ctx.part0_weight,
                                      part_span=span if span > 1 else 0,
                                      part_span_pitch_floats=self.slots * self.part_floats if span > 1 else 0) for b in range(flights)]
        pf, S = self.part_floats, self.slots
        self.part = [torch.zeros(world * S * pf, dtype=torch.float32, device="cuda") for _ in range(2)]
        self.gathered = [torch.zeros(world * S * pf, dtype=torch.float32, device="cuda") for _ in range(2)]
        self.fb = [torch.zeros(self.fb_floats, dtype=torch.float32, device="cuda") for _ in range(2)]
        self.host_fb = [torch.zeros(self.fb_floats, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.joiner = torch.cuda.Stream()  # waits for the frames of a group, then issues their exchange
        self.side = torch.cuda.Stream()    # as a root: waits for the exchange, restores the image layout, copies to the host
        self.sent = [None, None]           # event: the exchange that read part[g] / wrote gathered[g] is done
        self.untiled = [None, None]        # event: gathered[g] has been consumed
        self.last_delivered = None         # the host buffer that holds the newest frame this rank was root of
        self.frames_sent = 0               # frames handed to an exchange so far: the next group's first root follows from it
        self.frames_rooted = 0
        self.phase_events = None           # set by measure_phases

    # -- the three things a group goes through
    def render(self, k, want_stats=False):
        """Frame k of the run (and, with SPAN > 1, its successors of the same span) on its flight's stream."""
        if k % self.span:
            return None  # rendered together with the span's first frame
        torch = self.ctx.torch
        g, f, b = (k // GROUP) % 2, k % GROUP, (k // self.span) % self.flights
        block = self.ctx.rt.Comm.frame_blocks(self.ctx.world, self.slots, f)[1]
        with torch.cuda.stream(self.streams[b]):
            if self.sent[g] is not None:
                self.streams[b].wait_event(self.sent[g])  # the exchange of two groups ago has released part[g]
            return self.devs[b].render_device(self.flight_args[b], self.part[g].data_ptr() + block * self.part_floats * 4,
                                              stream=self.streams[b].cuda_stream, want_stats=want_stats)

    def exchange(self, g, count, first_root):
        """On the joiner stream, behind the group's renders: the parts of `count` frames to their roots."""
        c, torch, dist = self.ctx, self.ctx.torch, self.ctx.dist
        world, rank, S, pf = c.world, c.rank, self.slots, self.part_floats
        part, gathered = self.part[g], self.gathered[g]
        for b in range(self.flights):
            self.joiner.wait_stream(self.streams[b])
        with torch.cuda.stream(self.joiner):
            if self.untiled[g] is not None:
                self.joiner.wait_event(self.untiled[g])
            if self.phase_events is not None:
                self.phase_events["exchange"].append(self.joiner.record_event(torch.cuda.Event(enable_timing=True)))
            if c.staged:  # rehearsal on one GPU: the same blocks through host memory, one gather per root
                for p in range(min(count, world)):
                    root_rank = (first_root + p) % world
                    k = -(-(count - p) // world)
                    cpu = part[p * S * pf: (p * S + k) * pf].cpu()
                    bufs = [torch.empty_like(cpu) for _ in range(world)] if rank == root_rank else None
                    dist.gather(cpu, bufs, dst=root_rank)
                    if rank == root_rank:
                        for r in range(world):
                            gathered[r * S * pf: (r * S + k) * pf].copy_(bufs[r])
            elif c.comm is not None:
                c.comm.gather_frames(part.data_ptr(), pf, count, first_root, S, gathered.data_ptr(), stream=self.joiner.cuda_stream)
            else:  # the fallback: rt_comm_gather_frames' sends and receives through torch.distributed
                ops = []
                for p in range(min(count, world)):
                    k = -(-(count - p) // world)
                    ops.append(dist.P2POp(dist.isend, part[p * S * pf: (p * S + k) * pf], (first_root + p) % world))
                my_p = (rank - first_root) % world
                if my_p < count:
                    k = -(-(count - my_p) // world)
                    for r in range(world):
                        ops.append(dist.P2POp(dist.irecv, gathered[r * S * pf: (r * S + k) * pf], r))
                for req in dist.batch_isend_irecv(ops):
                    req.wait()  # (orders the current stream behind the transfer; the host does not block)
            if self.phase_events is not None:
                self.phase_events["exchange"].append(self.joiner.record_event(torch.cuda.Event(enable_timing=True)))
            self.sent[g] = self.joiner.record_event()

    def deliver(self, g, count, first_root, k0):
        """On the side stream, behind the exchange: the frames this rank is root of, un-tiled and copied home.
        k0: the number of the group's first frame in the run (the parts rotate with it)."""
        c, torch, rt = self.ctx, self.ctx.torch, self.ctx.rt
        mine = [f for f in range(count) if rt.Comm.frame_blocks(c.world, self.slots, f, first_root)[0] == c.rank]
        if not mine:
            return
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.sent[g])
            for f in mine:
                slot = rt.Comm.frame_blocks(c.world, self.slots, f, first_root)[2]
                o = self.frames_rooted & 1
                self.frames_rooted += 1
                if self.phase_events is not None:
                    self.phase_events["deliver"].append(self.side.record_event(torch.cuda.Event(enable_timing=True)))
                # rank r's part of the frame lies at block r * SLOTS + slot, and it is part (r + k) mod N of frame k of
                # the run when the parts rotate
                ua = self.args(part=(c.rank, c.world), part0_weight=c.part0_weight, part_rotation=(k0 + f) % c.world if self.rotate else 0)
                self.dev.untile_device(ua, self.gathered[g].data_ptr() + slot * self.part_floats * 4, self.fb[o].data_ptr(),
                                       stream=self.side.cuda_stream, part_pitch_floats=self.slots * self.part_floats)
                self.host_fb[o].copy_(self.fb[o], non_blocking=True)
                if self.phase_events is not None:
                    self.phase_events["deliver"].append(self.side.record_event(torch.cuda.Event(enable_timing=True)))
                self.last_delivered = self.host_fb[o]
            self.untiled[g] = self.side.record_event()

    def flush(self, g, count, k0):
        first_root = self.frames_sent % self.ctx.world
        self.frames_sent += count
        self.exchange(g, count, first_root)
        self.deliver(g, count, first_root, k0)

    def step(self, want_stats=False):
        k = self.counter
        self.counter += 1
        st = self.render(k, want_stats)
        if k % GROUP == GROUP - 1:
            self.flush((k // GROUP) % 2, GROUP, k - (GROUP - 1))
        return st

    def drain(self):
        k = self.counter
        if k % GROUP:
            self.flush((k // GROUP) % 2, k % GROUP, k - k % GROUP)
            self.counter = (k // GROUP + 1) * GROUP  # the next frame starts a group
        cur = self.ctx.torch.cuda.current_stream()
        cur.wait_stream(self.joiner)
        cur.wait_stream(self.side)
        for be in self.backends:
            be.sync()

    def rays_per_frame(self, stats):
        return float(stats.total_rays()) / self.span

    def measure_phases(self):
        """Outside the timed region: what this rank's share of ONE group costs, phase by phase, each phase alone on
        the GPU -- render_ms: the group's renders from first enqueue to the last kernel's end, per frame (wall
        clock around a device wait); exchange_ms: the grouped send / receive, per frame (events on its stream;
        through host memory in the gloo rehearsal); untile_copy_ms: restoring the layout of one frame and its copy
        to page-locked memory (events on the side stream; 0 on a rank that was root of none)."""
        torch = self.ctx.torch
        self.fence()
        t0 = time.perf_counter()
        for k in range(GROUP):
            self.render(self.counter + k)
        torch.cuda.synchronize()
        render_ms = (time.perf_counter() - t0) / GROUP * 1e3
        self.counter += GROUP
        self.phase_events = {"exchange": [], "deliver": []}
        t0 = time.perf_counter()
        self.flush(((self.counter - 1) // GROUP) % 2, GROUP, self.counter - GROUP)
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) * 1e3
        ev = self.phase_events
        self.phase_events = None
        exchange_ms = ev["exchange"][0].elapsed_time(ev["exchange"][1]) / GROUP
        if self.ctx.staged:
            exchange_ms = wall_ms / GROUP  # host-side gathers: the stream sees none of it
        pairs = list(zip(ev["deliver"][0::2], ev["deliver"][1::2]))
        untile_copy_ms = sum(a.elapsed_time(b) for a, b in pairs) / len(pairs) if pairs else 0.0
        return {"render_ms": render_ms, "exchange_ms": exchange_ms, "untile_copy_ms": untile_copy_ms}


class Context:
    pass


def make_comm(ctx, backend):
    """The library's own RCCL communicator: rank 0 makes the id, the launcher's process group carries it.  Should
    that communicator not come up on some node (a second copy of librccl in the process, a bootstrap problem),
    every rank falls back to the SAME exchange posted through torch.distributed -- one send per (rank, root) pair,
    one receive each, in one batch -- and the line says so: a scaling figure with a flag is worth more than none.
    RT_BENCH_EXCHANGE=torch forces the fallback."""
    rt, torch, dist = ctx.rt, ctx.torch, ctx.dist
    ctx.comm, ctx.exchange, ctx.rccl_version, ctx.exchange_note = None, "gloo" if ctx.staged else "library", None, None
    if ctx.staged:
        return
    failed = 0
    try:
        if os.environ.get("RT_BENCH_EXCHANGE", "library") == "torch":
            raise rt.RtError(-4, "RT_BENCH_EXCHANGE=torch")
        ctx.rccl_version = rt.Comm.rccl_version()
        ids = [rt.Comm.make_id() if ctx.rank == 0 else None]
    except rt.RtError as e:
        ids, failed = [None], 1
        ctx.exchange_note = f"rt_comm_id: {e}"
        print(f"[bench rank {ctx.rank}] rt_comm_id failed: {e}", file=sys.stderr)
    dist.broadcast_object_list(ids, src=0)
    if ids[0] is not None and not failed:
        # ncclCommInitRank on a thread of its own with a limit: a bootstrap that never returns (it has only ever
        # run with one device) must end in the flagged fallback, not in a run without a line
        import threading
        box = {}

        def create():
            try:
                box["comm"] = rt.Comm(backend, ids[0], ctx.rank, ctx.world)
            except rt.RtError as e:
                box["error"] = e

        limit = float(os.environ.get("RT_BENCH_COMM_TIMEOUT", "90"))
        worker = threading.Thread(target=create, daemon=True)
        worker.start()
        worker.join(limit)
        if "comm" in box:
            ctx.comm = box["comm"]
        else:
            failed = 1
            why = box.get("error", f"no answer within {limit:.0f} s")
            ctx.exchange_note = f"rt_comm_create: {why}"
            print(f"[bench rank {ctx.rank}] rt_comm_create failed: {why}", file=sys.stderr)
    else:
        failed = 1
    agreed = torch.tensor([failed], dtype=torch.int32, device="cuda")
    dist.all_reduce(agreed, op=dist.ReduceOp.MAX)
    if int(agreed.item()):
        if ctx.comm is not None:
            ctx.comm.close()
        ctx.comm = None
        ctx.exchange = "torch-fallback"
        try:
            ctx.rccl_version = int("".join(f"{int(x):02d}" for x in torch.cuda.nccl.version()[:3])) if ctx.rccl_version is None else ctx.rccl_version
        except Exception:
            pass


def reduce_over_ranks(ctx, values, op):
    """MAX / MIN / SUM of a list of floats over the ranks (a plain list back)."""
    torch, dist = ctx.torch, ctx.dist
    t = torch.tensor(values, dtype=torch.float64, device="cpu" if ctx.staged or not ctx.multi else "cuda")
    if ctx.multi:
        dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
    return [float(x) for x in t.tolist()]


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
    ap.add_argument("--no-extras", action="store_true", help="skip what runs after the timed region besides the roofline's kernel timing "
                                                              "(the 3840x2160 frames, the reference-order frames)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    # A frame cut into more than four parts leaves each GPU launches so small that a kernel lasts as
    # long as its slowest wave, and a frame's chain of kernels (0.2 ms for a 1/8 frame alone) sets
    # the pace: with HIP's default of four hardware queues at most four chains overlap.  Eight frames
    # in flight on queues of their own took the slowest 1/8 part from 0.184 to 0.177 ms per frame.  The
    # gather and un-tile streams need queues of their own as well: a stream that shares a hardware queue
    # with a render stream holds up that stream's kernels whenever it waits for an event (rank 0's step
    # with 8 queues for 11 streams: 0.25 ms per frame, with 32 queues: 0.19).  For a whole, half or
    # quarter frame the runtime's default is as good or slightly better.  (More than eight frames in flight
    # lose again: 16 cost a whole frame 13 % and a 1/8 frame 30 %.)  The variable is read when the
    # HIP runtime starts, so it is set before torch is imported.
    many_parts = int(os.environ.get("WORLD_SIZE", "1")) > 4
    hw_queues = os.environ.get("RT_BENCH_HW_QUEUES", "32" if many_parts else "")
    if hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = hw_queues

    import torch
    import torch.distributed as dist

    ctx = Context()
    ctx.torch, ctx.dist = torch, dist
    ctx.rank = rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ctx.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    # RT_BENCH_MULTI_PATH=1 runs the N > 1 code (process group, RCCL communicator of the library, grouped
    # gather, un-tile) with a single rank, as a check of that path on a one-GPU machine; the figures
    # are then not the N = 1 benchmark.
    ctx.multi = multi = world > 1 or os.environ.get("RT_BENCH_MULTI_PATH", "0") == "1"
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
    ctx.staged = staged = os.environ.get("RT_BENCH_BACKEND", "nccl") == "gloo"
    if staged:
        local_rank = local_rank % torch.cuda.device_count()
    ctx.local_rank = local_rank
    torch.cuda.set_device(local_rank)
    if multi:
        if staged:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ctx.rt = rt = importlib.import_module("rust-raytracer_amd")
    gen = importlib.import_module("scenes.gen_scenes")
    if rank == 0:
        gen.ensure_models()
        bunny = [gen.bunny_workload_scene()]  # the real mesh if scenes/models/bunny.obj is there, else the stand-in
    else:
        bunny = [None]
    if multi:
        dist.broadcast_object_list(bunny, src=0)  # (also the barrier behind rank 0's writing of the model files)
    ctx.bunny = bunny[0]
    real_mesh = ctx.bunny[0] != "bunny"

    ctx.mode = rt.RT_MODE_FAST if args.mode == "fast" else rt.RT_MODE_REFERENCE_ORDER
    ctx.render_flags = args.render_flags
    # Equal parts: the root of the gather rotates over the ranks from frame to frame, so no rank has more to
    # do than another.  (RtRenderArgs.part0_weight is there for a caller that keeps one fixed root.)
    ctx.part0_weight = int(os.environ.get("RT_BENCH_PART0_WEIGHT", "0"))

    # FLIGHTS frames are in flight at a time, each with a backend (ray queues), a stream and
    # output buffers of its own: the tail of one frame's kernels (a few long-running waves) and its
    # copy to the host then overlap the next frame's work instead of idling the GPU.
    FLIGHTS = args.flights or int(os.environ.get("RT_BENCH_FLIGHTS", "8" if many_parts else "4"))
    run = (ManyGpus if multi else OneGpu)(ctx, args.workload, FLIGHTS)
    torch.cuda.set_stream(run.streams[0])
    if multi:
        make_comm(ctx, run.backends[0])
    else:
        ctx.comm, ctx.exchange, ctx.rccl_version, ctx.exchange_note = None, "none (one GPU)", None, None

    # Warm-up frames are rendered with the backend's full statistics (an event pair around every
    # kernel, counters read back, the host waits for each frame, so every kernel runs alone on the
    # GPU): they give the ray counts, which are the same for every frame, and the split of a frame
    # over the kernel classes.  Every flight twice at least: a backend's first frame touches its
    # buffers for the first time.
    span = getattr(run, "span", 1)
    warm_stats = [s for s in (run.step(True) for _ in range(max(args.warmup, 2 * FLIGHTS * span))) if s is not None]
    run.fence()
    s0 = warm_stats[-1]  # the last warm-up frame: caches and clocks have settled
    k_warm = [s0.kernel_ms[k] for k in range(rt.RT_K_COUNT)]
    dom = max(range(len(rt.KERNEL_CLASSES)), key=lambda k: k_warm[k])
    dom_name = rt.KERNEL_CLASSES[dom]
    rays_per_frame_local = run.rays_per_frame(s0)

    # untimed frames through the path that is timed next: its buffers are touched for the first time,
    # the runtime grows its pools of signals and command buffers, the copy engines and clocks come up.
    # Measured: the enqueue cost per frame falls from 1.8 ms to 0.03 ms over the first 24 frames.
    for _ in range(8 * FLIGHTS if not multi else 4 * GROUP):
        run.step()
    # ---- the timed region: EXACTLY args.steps frames, each delivered to host memory
    elapsed = run.timed(run.step, args.steps)

    extra = {}
    solo_ms, solo_launches = None, 0
    if not multi:
        # the same frames left in device memory, FLIGHTS in flight (no copy to the host) ...
        for _ in range(FLIGHTS):
            run.step_device()
        el_dev = run.timed(run.step_device, args.steps)
        extra["value_device_resident"] = rays_per_frame_local * args.steps / el_dev
        extra["ms_per_step_device_resident"] = el_dev / args.steps * 1e3
        # ... and one frame at a time through the blocking entry point, rt_render into page-locked memory
        single = max(20, args.steps // 4)
        t0 = time.perf_counter()
        for _ in range(single):
            run.dev.render(run.n, mode=ctx.mode, tile=(run.tile, run.tile), flags=args.render_flags, want_stats=False, out=run.host[0].array)
        el_single = time.perf_counter() - t0
        extra["value_single_frame"] = rays_per_frame_local * single / el_single
        extra["ms_per_step_single_frame"] = el_single / single * 1e3
        # Duration of the dominant kernel for the roofline: frames one at a time (the kernel has the GPU
        # to itself, as in the profiles under profiles/), an event pair around that kernel class only.
        run.rargs.timed_classes = 1 << dom
        for _ in range(single):
            run.dev.render_device(run.rargs, run.fbs[0].data_ptr(), stream=run.streams[0].cuda_stream)
            run.backends[0].sync()
        t = run.backends[0].collect_timings()
        run.rargs.timed_classes = 0
        solo_ms = t.kernel_ms[dom] / max(t.kernel_launches[dom], 1)
        solo_launches = int(t.kernel_launches[dom])
        if args.mode == "fast" and not args.no_extras:
            # The benchmarked mode stops a shadow ray at the first blocker it meets, so its shadow rays are checked on
            # "hidden or not"; blocker ids are compared in reference-order mode, timed here on the same frame.
            t0 = time.perf_counter()
            for _ in range(5):
                run.dev.render(run.n, mode=rt.RT_MODE_REFERENCE_ORDER, tile=(run.tile, run.tile), want_stats=False, out=run.host[0].array)
            extra["value_reference_order"] = rays_per_frame_local * 5 / (time.perf_counter() - t0)
            # ... and in the fast mode itself with RT_RENDER_CLOSEST_BLOCKERS (packets on the mode's own trees, every shadow
            # ray to its closest blocker): the same pixels, blocker ids comparable one to one
            t0 = time.perf_counter()
            for _ in range(5):
                run.dev.render(run.n, tile=(run.tile, run.tile), want_stats=False, out=run.host[0].array, flags=rt.RT_RENDER_CLOSEST_BLOCKERS)
            extra["value_closest_blockers"] = rays_per_frame_local * 5 / (time.perf_counter() - t0)
            extra["parity_mode_note"] = ("shadow rays are checked on the occlusion bit in this mode (fast: any-hit); blocker shape and triangle "
                                         "ids are checked (closest blocker, light/point.rs:35) in the fast mode with RT_RENDER_CLOSEST_BLOCKERS, "
                                         f"which renders this frame at {extra['value_closest_blockers'] / 1e9:.2f} Grays/s, and in reference-order "
                                         f"mode at {extra['value_reference_order'] / 1e9:.2f} Grays/s (5 blocking frames each, same run)")

    # Outside the timed region: the gathered frame must be bit-identical with the same frame rendered
    # whole (static tile map, ordered per-pixel sums).  Every rank that was a root checks the newest
    # frame it delivered against its own single-GPU render of the frame.
    frame_check = None
    phases = None
    if multi and world == 1:
        frame_check = "not applicable: a single part is rendered in image layout, not tile by tile"
        phases = run.measure_phases()
    elif multi:
        bad = 0
        if run.last_delivered is not None:
            whole = torch.zeros(run.fb_floats, dtype=torch.float32, device="cuda")
            run.dev.render_device(run.args(), whole.data_ptr(), stream=run.streams[0].cuda_stream)
            run.backends[0].sync()
            torch.cuda.synchronize()
            bad = int((whole.cpu() != run.last_delivered).sum())
        checked = reduce_over_ranks(ctx, [bad, 1 if run.last_delivered is not None else 0], "SUM")
        if int(checked[0]) != 0:
            raise SystemExit(f"gathered frames differ from the single-GPU frame in {int(checked[0])} values")
        frame_check = (f"newest gathered frame in host memory bit-identical with a single-GPU render of the frame, "
                       f"on each of the {int(checked[1])} ranks that served as root")
        phases = run.measure_phases()

    # ---- config 5 of the benchmark list, 3840 x 2160 at 64 rays per pixel: 5 frames through the same path, after
    # the timed region.  The configuration the multi-GPU sentence of the target names, and the one where the fixed
    # cost of a frame's launches and of the exchange vanishes beside the render.
    if args.workload == "bunny" and not args.no_extras:
        big = (ManyGpus if multi else OneGpu)(ctx, "bunny4k", 2)
        big_stats = [s for s in (big.step(True) for _ in range(2 * getattr(big, "span", 1))) if s is not None]
        big_rays = reduce_over_ranks(ctx, [big.rays_per_frame(big_stats[-1])], "SUM")[0]
        for _ in range(GROUP if multi else 4):
            big.step()
        # (as many frames as keep the run's tail -- the last frame's way home, 199 MB over one PCIe link, 3.6 ms -- small
        # beside the run: 5 on one GPU, where a frame takes ten times that, 3 per rank on several)
        n_big = 5 if not multi else 3 * world
        el_big = reduce_over_ranks(ctx, [big.timed(big.step, n_big)], "MAX")[0]
        extra["value_bunny4k"] = big_rays * n_big / el_big
        extra["ms_per_step_bunny4k"] = el_big / n_big * 1e3
        extra["bunny4k_is"] = f"{big.desc}; {n_big} frames delivered to host memory after the timed region, 2 frames in flight per GPU"
        big.close()

    elapsed = reduce_over_ranks(ctx, [elapsed], "MAX")[0]
    total_rays = reduce_over_ranks(ctx, [rays_per_frame_local * args.steps], "SUM")[0]
    if phases is not None:
        keys = sorted(phases)
        hi = reduce_over_ranks(ctx, [phases[k] for k in keys], "MAX")
        lo = reduce_over_ranks(ctx, [phases[k] for k in keys], "MIN")
        extra["per_rank_ms"] = {k: {"max": hi[i], "min": lo[i]} for i, k in enumerate(keys)}
        extra["per_rank_ms"]["what"] = ("one group of 8 frames after the timed region, each phase alone on the GPU, per frame: this rank's "
                                        "renders (wall clock to the last kernel's end), the grouped send / receive (events), un-tile + copy "
                                        "home of one frame it is root of (events); maximum and minimum over the ranks")
        extra["per_rank_ms"]["render_ms_alone_rank0"] = s0.render_ms / span  # rank 0's instrumented warm-up render: its kernels alone on the GPU

    if rank == 0:
        value = total_rays / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = [s0.kernel_ms[k] for k in range(rt.RT_K_COUNT)]
        k_n = [s0.kernel_launches[k] for k in range(rt.RT_K_COUNT)]
        ab = algorithmic_bytes(args.workload) if not real_mesh or not args.workload.startswith("bunny") else None
        roofline = roofline_issue = roofline_l2 = None
        hbm_true = None
        cpu = {}
        if not multi and not args.no_cpu_baseline:
            # one frame of the benchmarked workload itself on one core (the reference's loop is serial,
            # render.rs:36-50): about 9 s for the bunny frame; the 4K workload is sampled at 1000x1000
            sample_px = 1000
            v1, dt1, counters = cpu_baseline(rt, run.scene_file, run.depth, sample_px, 3, 1)
            cpu["cpu_baseline"] = {"value": v1, "unit": "rays/s", "cores": 1, "kind": "port",
                                   "sample": f"one frame of the same scene at {sample_px}x{sample_px} rpp9 depth{run.depth}, "
                                             f"C++ restatement of the reference (not the Rust build), {dt1:.1f} s"}
            cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
            vN, dtN, _ = cpu_baseline(rt, run.scene_file, run.depth, 1000, 3, cores)
            cpu["cpu_baseline_all_cores"] = {"value": vN, "unit": "rays/s", "cores": cores, "kind": "port",
                                             "sample": f"same scene at 1000x1000 rpp9, OpenMP over pixel columns, {dtN:.1f} s"}
            if ab is None and args.workload.startswith("bunny"):
                # a mesh file of this machine's: no committed byte counts, so they come from that same oracle frame
                ab = importlib.import_module("tools.algorithmic_bytes").per_ray_bytes(counters)
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
        prof = pmc_profile(args.workload, dom_name) if not multi and not real_mesh else None
        prof_frame = pmc_profile(args.workload, None) if not multi and not real_mesh else None
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
                                            "HBM does not bound it: see roofline_issue, roofline_l2 and hbm_true_frac",
            }
            if prof:
                roofline["counters"] = {k: prof.get(k) for k in ("hbm_gbs", "l2_hit_rate", "valu_lane_utilisation")}
                roofline["traffic_frac"] = prof["bytes"] / (solo_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            if roof_note:
                roofline["note"] = roof_note
        if prof and prof.get("valu_active_quad_cycles"):
            # What does bound the kernel: instruction issue, from the busy counters of the committed profile.  Each
            # figure's formula stands beside it; the counter values are those of profiles/pmc_traffic.json.
            valu_ms = prof["valu_active_quad_cycles"] * 4.0 / SIMDS / (CLOCK_GHZ * 1e9) * 1e3
            scalar_cycles = prof.get("salu_cycles", 0.0) + prof.get("smem_cycles", 0.0) + prof.get("branch_insts", 0.0)
            scalar_ms = scalar_cycles / CUS / (CLOCK_GHZ * 1e9) * 1e3
            bound_ms = max(valu_ms, scalar_ms)
            roofline_issue = {
                "kernel": prof["kernel"],
                "SQ_ACTIVE_INST_VALU": prof["valu_active_quad_cycles"], "SQ_INST_CYCLES_SALU": prof.get("salu_cycles"),
                "SQ_INST_CYCLES_SMEM": prof.get("smem_cycles"), "SQ_INSTS_BRANCH": prof.get("branch_insts"),
                "SQ_WAIT_ANY": prof.get("wait_any"), "SQ_WAVE_CYCLES": prof.get("wave_cycles"),
                "clock_ghz": CLOCK_GHZ,
                "valu_bound_ms": valu_ms, "valu_bound_is": "SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / clock",
                "scalar_bound_ms": scalar_ms, "scalar_bound_is": "(SQ_INST_CYCLES_SALU + SQ_INST_CYCLES_SMEM + SQ_INSTS_BRANCH) / 256 CUs / clock",
                "bound_ms": bound_ms, "kernel_ms": solo_ms, "frac": bound_ms / solo_ms, "frac_is": "max(valu_bound_ms, scalar_bound_ms) / kernel_ms",
                "wait_frac": (prof["wait_any"] / prof["wave_cycles"]) if prof.get("wait_any") and prof.get("wave_cycles") else None,
                "wait_frac_is": "SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of its resident cycles a wave spends waiting",
                "from": "counters per launch: profiles/pmc_traffic.json (rocprofv3 --pmc); kernel_ms: live, as roofline.avg_launch_ms",
                # the duration of the same kernel in the kernel trace taken with the counter passes: when it
                # is far from the live one, the committed counts belong to another version of the kernel
                "kernel_ms_in_profile": prof["avg_us"] * 1e-3 if prof.get("avg_us") else None,
                "profile_matches_this_build": (abs(prof["avg_us"] * 1e-3 - solo_ms) <= 0.1 * solo_ms) if prof.get("avg_us") else None}
        if prof and prof.get("tcc_req"):
            l2_gbs = prof["tcc_req"] * 128.0 / (solo_ms * 1e-3) / 1e9
            roofline_l2 = {"kernel": prof["kernel"], "TCC_REQ_sum": prof["tcc_req"], "achieved": l2_gbs, "peak": L2_PEAK_GBS, "unit": "GB/s",
                           "frac": l2_gbs / L2_PEAK_GBS, "frac_is": "TCC_REQ_sum x 128 B / kernel_ms / 34.5 TB/s (the L2s' streaming rate)",
                           "kernel_ms": solo_ms}
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
            "exchange": ctx.exchange,
            "rccl_version": ctx.rccl_version,
            "exchange_note": ctx.exchange_note or ("another rank's communicator failed" if ctx.exchange == "torch-fallback" else None),
            "config": {"workload": run.desc, "traversal": args.mode,
                       "mesh": "file found on this machine (scenes/models/bunny.obj)" if real_mesh and args.workload.startswith("bunny") else
                               "procedural stand-in" if args.workload.startswith("bunny") else "none",
                       "value_is": "frames delivered to page-locked host memory (what render() returns, render.rs:7-23), "
                                   f"{FLIGHTS} frames in flight",
                       "precision": ("fp32 tree search, f64 hit refinement (point, normal, mirror direction, self-hit tests), "
                                     + ("light sums in f32 on the f32-rounded vectors (fast mode), " if args.mode == "fast" else "f64 light sums, ")
                                     + "f32 colour (the reference: f64 geometry, f32 colour)"),
                       "tile": run.tile, "tiles": getattr(run, "tiles_total", None),
                       "rays_per_step": total_rays / args.steps, "parallelism": f"tiles/{world}",
                       "part0_weight": ctx.part0_weight, "frames_in_flight": FLIGHTS, "hw_queues": int(hw_queues) if hw_queues else "runtime default",
                       "rccl_ranks": dist.get_world_size() if multi else 1,
                       "gather": ("none (one GPU)" if not multi else
                                  "host memory over gloo (one-GPU rehearsal)" if staged else
                                  "FALLBACK: the library's communicator did not come up; the same sends and receives "
                                  "through torch.distributed (batch_isend_irecv)" if ctx.exchange == "torch-fallback" else
                                  f"rt_comm_gather_frames: groups of {GROUP} frames, one grouped ncclSend / ncclRecv exchange per group "
                                  "on the library's RCCL communicator, frame k gathered to rank k mod N"),
                       "parts": ("one (whole frames)" if world == 1 else
                                 "rotating: rank r renders part (r + k) mod N of frame k" if run.rotate else
                                 "fixed: rank r renders part r"),
                       "parts_per_set_of_launches": span,
                       "timed_region": "all ranks start behind a barrier with idle GPUs; each stops its clock when its last frame is in host "
                                       "memory; the maximum over the ranks is taken",
                       "render_flags": args.render_flags,
                       "scene_copies_per_gpu": len({int(d.handle.value) for d in run.devs}),  # the frames in flight share one uploaded scene
                       "scene_upload_ms": run.upload_s * 1e3,
                       "kernel_split_from": "last warm-up frame (fully instrumented, every kernel alone on the GPU)"},
            "kernel_ms_per_step": {rt.KERNEL_CLASSES[k]: k_ms[k] / span for k in range(len(rt.KERNEL_CLASSES))},
            "device_ms_per_step": s0.render_ms / span,
            "roofline": roofline,
            "roofline_issue": roofline_issue,
            "roofline_l2": roofline_l2,
            "hbm_true_frac": hbm_true,
        }
        out.update(extra)
        out.update(cpu)
        if frame_check:
            out["config"]["frame_check"] = frame_check
        if not multi:
            # the delivered frame is a frame: compare a few pixels of it with the device copy's statistics
            out["config"]["delivered_frame_checksum"] = float(run.delivered.astype("float64").sum())
        print(json.dumps(out), flush=True)

    if ctx.comm is not None:
        ctx.comm.close()
    run.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
