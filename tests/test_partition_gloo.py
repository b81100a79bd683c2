"""Two ranks over gloo rehearse the multi-GPU frame protocol on the CPU: each rank renders
the tiles it owns (the backend's stage bodies on the host), packs them into the compact
tile-major part buffer that rt_render_device writes, one gather brings the part buffers to
rank 0, and rank 0 rebuilds the frame the way rt_untile_device does.  The result must be
bit-identical with the unpartitioned frame (static tile map, ordered per-pixel sums)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, TILE, N = 100, 70, 16, 2


def pack_part(fb, tiles, tiles_x, stride):
    """Compact part buffer: slot s holds tile tiles[s], tile_h rows of tile_w RGB pixels, bottom-up,
    zero padded at the image edge and up to `stride` tiles."""
    out = np.zeros((stride, TILE, TILE, 3), np.float32)
    for s, k in enumerate(tiles):
        ty, tx = divmod(int(k), tiles_x)
        blk = fb[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
        out[s, :blk.shape[0], :blk.shape[1]] = blk
    return out


def unpack_parts(gathered, tile_lists, tiles_x):
    fb = np.zeros((H, W, 3), np.float32)
    for p, tiles in enumerate(tile_lists):
        for s, k in enumerate(tiles):
            ty, tx = divmod(int(k), tiles_x)
            y0, x0 = ty * TILE, tx * TILE
            h, w = min(TILE, H - y0), min(TILE, W - x0)
            fb[y0:y0 + h, x0:x0 + w] = gathered[p][s, :h, :w]
    return fb


def worker(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hostsim_lib as HS

        rt = HS.rt
        sc = rt.parse_scene_file(os.path.join(ROOT, "scenes", "mixed_features.json"), image_width=W, image_height=H,
                                 recursion_depth=6)
        a = rt.RtRenderArgs()
        a.sqrt_rays_per_pixel, a.tile_w, a.tile_h, a.part_index, a.part_count = N, TILE, TILE, rank, world
        total, owned, stride = C_partition_info(rt, a)
        tiles_x = (W + TILE - 1) // TILE
        tiles = rt.partition_tiles(W, H, a)
        assert tiles.size == owned
        fb, _, stats = HS.render(sc, N, want_ids=False, tile=(TILE, TILE), part=(rank, world))
        part = torch.from_numpy(pack_part(fb, tiles, tiles_x, stride))
        rays = torch.tensor([float(stats.total_rays())], dtype=torch.float64)
        bufs = [torch.zeros_like(part) for _ in range(world)] if rank == 0 else None
        dist.gather(part, bufs, dst=0)
        dist.all_reduce(rays)
        if rank == 0:
            lists = []
            for p in range(world):
                a.part_index = p
                lists.append(rt.partition_tiles(W, H, a))
            frame = unpack_parts([b.numpy() for b in bufs], lists, tiles_x)
            full, _, full_stats = HS.render(sc, N, want_ids=False, tile=(TILE, TILE))
            np.savez(result_path, same=np.array_equal(frame, full), rays=rays.item(), full_rays=full_stats.total_rays(),
                     covered=sum(len(l) for l in lists), total=total)
    finally:
        dist.destroy_process_group()


def C_partition_info(rt, a):
    import ctypes as C

    t, o, m = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert rt.lib().rt_partition_info(W, H, C.byref(a), C.byref(t), C.byref(o), C.byref(m)) == 0
    return t.value, o.value, m.value


@pytest.mark.timeout(300)
def test_two_rank_gather_rebuilds_the_frame(tmp_path):
    import hostsim_lib as HS

    HS.lib()  # build the simulator once, before the ranks race to
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    result = str(tmp_path / "result.npz")
    mp.spawn(worker, args=(2, port, result), nprocs=2, join=True)
    r = np.load(result)
    assert bool(r["same"])
    assert r["covered"] == r["total"]
    assert r["rays"] == r["full_rays"]  # the parts traced exactly the rays of the whole frame
