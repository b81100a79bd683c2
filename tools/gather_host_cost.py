#!/usr/bin/env python3
"""What bench.py's multi-GPU step costs rank 0 per frame beyond the render itself, measured on ONE GPU
with a one-rank RCCL group: the frame is cut into 8 parts, part 0 is rendered, and the gather
degenerates to a local copy, so the figures are the cost of the stream hand-overs and of the Python and
torch.distributed layers alone -- a lower bound of what a real 8-rank gather adds.

    python3 tools/gather_host_cost.py [frames] [group]

group = frames gathered together (1 = a gather per frame; bench.py gathers the frames in flight together).
RT_FLIGHTS = frames in flight (default 8), RT_PARTS = parts (default 8), RT_PART0_WEIGHT = RtRenderArgs.part0_weight
(default 0), GPU_MAX_HW_QUEUES as for bench.py.
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    flights, parts = int(os.environ.get("RT_FLIGHTS", "8")), int(os.environ.get("RT_PARTS", "8"))
    group = int(sys.argv[2]) if len(sys.argv) > 2 else flights
    assert flights % group == 0 or group % flights == 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    bench = importlib.import_module("bench")
    rt = importlib.import_module("rust-raytracer_amd")
    importlib.import_module("scenes.gen_scenes").ensure_models()
    scene_name, W, H, n, depth, tile, _ = bench.WORKLOADS["bunny"]
    scene = rt.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".json"), image_width=W, image_height=H,
                                recursion_depth=depth)
    bes = [rt.Backend(0) for _ in range(flights)]
    first = bes[0].upload(scene)  # one copy of the scene for all frames in flight
    devs = [first] + [first.on(b) for b in bes[1:]]
    a = devs[0].render_args(n, tile=(tile, tile), part=(0, parts), part0_weight=int(os.environ.get("RT_PART0_WEIGHT", "0")))
    _, owned, stride = devs[0].partition_info(a)
    pf = stride * tile * tile * 3
    streams = [torch.cuda.Stream() for _ in range(flights)]
    joiner, side = torch.cuda.Stream(), torch.cuda.Stream()
    part = [torch.zeros(group * pf, dtype=torch.float32, device="cuda") for _ in range(2)]
    big = [torch.zeros(parts * group * pf, dtype=torch.float32, device="cuda") for _ in range(2)]  # what rank 0 would receive
    gathered = [x[: group * pf] for x in big]
    fb = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    pending, untiled = [None, None], [None, None]
    what = os.environ.get("RT_STEP", "all")  # render | gather | all | copy | copy+untile
    t_all = 0.0
    for k in range(frames + 4 * group):
        if k == 4 * group:
            torch.cuda.synchronize()
            t_all = time.perf_counter()
        g, f = (k // group) % 2, k % group
        b = k % flights
        with torch.cuda.stream(streams[b]):
            if pending[g] is not None:
                streams[b].wait_event(pending[g])
            devs[b].render_device(a, part[g].data_ptr() + f * pf * 4, stream=streams[b].cuda_stream)
        if f != group - 1 or what == "render":
            continue
        for s in streams:
            joiner.wait_stream(s)
        with torch.cuda.stream(joiner):
            if untiled[g] is not None:
                joiner.wait_event(untiled[g])
            if what.startswith("copy"):  # a plain device copy in place of the collective
                gathered[g].copy_(part[g])
            else:
                dist.gather(part[g], [gathered[g]], dst=0)
            pending[g] = joiner.record_event()
        if what in ("all", "copy+untile"):
            with torch.cuda.stream(side):
                side.wait_event(pending[g])
                # one rank's [frame][block]; the blocks of the other seven ranks would lie `group` blocks apart
                for fr in range(group):
                    devs[0].untile_device(a, big[g].data_ptr() + fr * pf * 4, fb.data_ptr(), stream=side.cuda_stream,
                                          part_pitch_floats=group * pf)
                untiled[g] = side.record_event()
    host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    print(f"{flights} frames in flight, {group} per gather, step = {what}: host {host / frames * 1e3:.3f} ms per frame, "
          f"wall {wall / frames * 1e3:.3f} ms per frame")
    for d in devs:
        d.close()
    for b in bes:
        b.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
