#!/usr/bin/env python3
"""Ray counts and kernel times (each kernel alone on the GPU) of every part of an N-part split.

    python3 tools/part_stats.py [parts]
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    bench = importlib.import_module("bench")
    rt = importlib.import_module("rust-raytracer_amd")
    importlib.import_module("scenes.gen_scenes").ensure_models()
    scene_name, W, H, n, depth, tile, _ = bench.WORKLOADS["bunny"]
    tile = int(os.environ.get("RT_TILE", tile))
    scene = rt.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".json"), image_width=W, image_height=H,
                                recursion_depth=depth)
    be = rt.Backend(0)
    dev = be.upload(scene)
    st = torch.cuda.Stream()
    buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    for p in range(parts):
        a = dev.render_args(n, tile=(tile, tile), part=(p, parts))
        _, owned, stride = dev.partition_info(a)
        for _ in range(5):
            s = dev.render_device(a, buf.data_ptr(), stream=st.cuda_stream, want_stats=True)
        ms = " ".join(f"{rt.KERNEL_CLASSES[k]}={s.kernel_ms[k]:.3f}" for k in range(len(rt.KERNEL_CLASSES)))
        print(f"part {p}: tiles {owned} primary {s.primary_rays} shadow {s.shadow_rays} secondary {s.secondary_rays} "
              f"device {s.render_ms:.3f} ms  {ms}")
    dev.close()
    be.close()


if __name__ == "__main__":
    main()
