#!/usr/bin/env python3
"""Why does 1/N of a frame cost more than 1/N of the frame's time?  Renders, with `flights` frames in flight:
the whole frame; each of N interleaved parts (mean and slowest); and N contiguous bands of the frame as
windows (same pixels and launches per render as a part, but neighbouring tiles).

    python3 tools/part_probe.py [frames] [flights] [parts ...]
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    flights = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    part_counts = [int(x) for x in sys.argv[3:]] or [8]
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
    bufs = [torch.zeros(W * H * 3 + 40 * 40 * 3 * 700, dtype=torch.float32, device="cuda") for _ in range(flights)]

    def timed(args_list, reps):
        """`reps` rounds over args_list (one render each, round robin over the flights); ms per render"""
        k = 0
        for _ in range(-(-4 * flights // len(args_list))):  # every flight has rendered this four times
            for a in args_list:
                devs[k % flights].render_device(a, bufs[k % flights].data_ptr(), stream=sts[k % flights].cuda_stream)
                k += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for a in args_list:
                devs[k % flights].render_device(a, bufs[k % flights].data_ptr(), stream=sts[k % flights].cuda_stream)
                k += 1
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * len(args_list)) * 1e3

    whole = timed([devs[0].render_args(n, tile=(tile, tile))], frames)
    print(f"whole frame, {flights} in flight: {whole:.4f} ms")
    for parts in part_counts:
        each = [timed([devs[0].render_args(n, tile=(tile, tile), part=(p, parts))], frames) for p in range(parts)]
        rot = timed([devs[0].render_args(n, tile=(tile, tile), part=(p, parts)) for p in range(parts)], max(frames // parts, 1))
        band = H // parts
        bands = [timed([devs[0].render_args(n, tile=(tile, tile), window=(0, b * band, W, band))], frames) for b in range(parts)]
        for span in (2, 4):
            if span >= parts:
                continue
            sp = timed([devs[0].render_args(n, tile=(tile, tile), part=(p, parts), part_span=span) for p in range(0, parts, span)],
                       max(frames // parts, 1)) / span
            print(f"{parts} parts, {span} at a time, rotating: {sp:.4f} per part (eff {whole / parts / sp:.3f})")
        print(f"{parts} parts: ideal {whole / parts:.4f}; interleaved mean {sum(each) / parts:.4f} slowest {max(each):.4f} "
              f"(eff {whole / parts / (sum(each) / parts):.3f} / {whole / parts / max(each):.3f}); rotating {rot:.4f} (eff {whole / parts / rot:.3f}); "
              f"contiguous bands mean {sum(bands) / parts:.4f} (eff {whole / parts / (sum(bands) / parts):.3f})")
    print(f"whole frame again: {timed([devs[0].render_args(n, tile=(tile, tile))], frames):.4f} ms")
    for d in devs:
        d.close()
    for b in bes:
        b.close()


if __name__ == "__main__":
    main()
