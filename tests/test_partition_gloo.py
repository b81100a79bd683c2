"""Ranks over gloo rehearse the multi-GPU frame protocol on the CPU: each rank renders
the tiles it owns (the backend's stage bodies on the host), packs them into the compact
tile-major part buffer that rt_render_device writes, one gather brings the part buffers to
rank 0, and rank 0 rebuilds the frame the way rt_untile_device does.  The result must be
bit-identical with the unpartitioned frame (static tile map, ordered per-pixel sums).  Variants:
three ranks with a smaller share for rank 0 (RtRenderArgs.part0_weight), and several frames
gathered with one collective, which rank 0 receives as [rank][frame][block]."""
import importlib
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


def worker(rank, world, port, result_path, weight=0, frames=1):
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
        a.sqrt_rays_per_pixel, a.tile_w, a.tile_h, a.part_index, a.part_count, a.part0_weight = N, TILE, TILE, rank, world, weight
        total, owned, stride = C_partition_info(rt, a)
        tiles_x = (W + TILE - 1) // TILE
        tiles = rt.partition_tiles(W, H, a)
        assert tiles.size == owned
        # frame f: jittered sampling with seed f, so that the frames of a group differ (frame 0: the regular grid)
        mine, n_rays = [], 0.0
        for f in range(frames):
            fb, _, stats = HS.render(sc, N, want_ids=False, tile=(TILE, TILE), part=(rank, world), part0_weight=weight,
                                     aa=1 if f else 0, seed=f)
            mine.append(pack_part(fb, tiles, tiles_x, stride))
            n_rays += stats.total_rays()
        part = torch.from_numpy(np.stack(mine))  # [frame][block]: what one rank sends
        rays = torch.tensor([n_rays], dtype=torch.float64)
        bufs = [torch.zeros_like(part) for _ in range(world)] if rank == 0 else None
        dist.gather(part, bufs, dst=0)  # ONE collective for all frames
        dist.all_reduce(rays)
        if rank == 0:
            lists = []
            for p in range(world):
                a.part_index = p
                lists.append(rt.partition_tiles(W, H, a))
            same, full_rays = True, 0
            for f in range(frames):  # rank 0 holds [rank][frame][block]
                frame = unpack_parts([b.numpy()[f] for b in bufs], lists, tiles_x)
                full, _, full_stats = HS.render(sc, N, want_ids=False, tile=(TILE, TILE), aa=1 if f else 0, seed=f)
                same = same and np.array_equal(frame, full)
                full_rays += full_stats.total_rays()
            np.savez(result_path, same=same, rays=rays.item(), full_rays=full_rays,
                     covered=sum(len(l) for l in lists), total=total, sizes=np.array([len(l) for l in lists]))
    finally:
        dist.destroy_process_group()


def C_partition_info(rt, a):
    import ctypes as C

    t, o, m = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert rt.lib().rt_partition_info(W, H, C.byref(a), C.byref(t), C.byref(o), C.byref(m)) == 0
    return t.value, o.value, m.value


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,weight,frames", [(2, 0, 1), (3, 5, 2)])
def test_gather_rebuilds_the_frame(tmp_path, world, weight, frames):
    import hostsim_lib as HS

    HS.lib()  # build the simulator once, before the ranks race to
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    result = str(tmp_path / "result.npz")
    mp.spawn(worker, args=(world, port, result, weight, frames), nprocs=world, join=True)
    r = np.load(result)
    assert bool(r["same"])
    assert r["covered"] == r["total"]
    assert r["rays"] == r["full_rays"]  # the parts traced exactly the rays of the whole frames
    if weight:
        # rank 0 took the smaller share (35 tiles over 21 lanes: the shares are coarse on this small grid)
        assert r["sizes"][0] <= r["sizes"][1:].min() and r["sizes"][0] < r["sizes"][1:].mean()


def test_frame_exchange_layout_is_a_permutation():
    """The block layout of rt_comm_gather_frames (Comm.frame_blocks): for every world size, group length
    and first root, the frames of one root occupy adjacent blocks of the send buffer in frame order
    (one message per root), no two frames share a block, and a root's frames land in distinct slots."""
    rt = importlib.import_module("rust-raytracer_amd")
    for world in (1, 2, 3, 4, 8):
        for count in (1, 2, 4, 7, 8, 16):
            slots = -(-count // world)
            for first_root in range(world):
                send_blocks, per_root = set(), {}
                for f in range(count):
                    root, at, slot = rt.Comm.frame_blocks(world, slots, f, first_root)
                    assert root == (first_root + f) % world and 0 <= at < world * slots and at not in send_blocks
                    send_blocks.add(at)
                    per_root.setdefault(root, []).append((at, slot))
                for root, frames in per_root.items():
                    p = (root - first_root) % world
                    assert [a for a, _ in frames] == list(range(p * slots, p * slots + len(frames)))
                    assert [s for _, s in frames] == list(range(len(frames)))
                    assert len(frames) == (count - p + world - 1) // world  # what the library sends to this root
    with pytest.raises(ValueError):
        rt.Comm.frame_blocks(2, 1, 2)
