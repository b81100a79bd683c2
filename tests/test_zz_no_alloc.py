"""rt_render_device_static: the render entry point that allocates nothing (its work memory is a static
array inside the library's code object), against rt_render_device, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def render_both(rt, backend, scene, n, **kw):
    """The same frame through rt_render_device and rt_render_device_static; returns the two images and
    the device's free memory around the static call."""
    import torch

    dev = backend.upload(scene)
    W, H = dev.width, dev.height
    want = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    got = torch.full((W * H * 3,), -1.0, dtype=torch.float32, device="cuda")
    a = dev.render_args(n, **kw)
    torch.cuda.synchronize()
    dev.render_device(a, want.data_ptr())
    backend.sync()
    free_before = torch.cuda.mem_get_info()[0]
    dev.render_device_static(a, got.data_ptr())
    free_after_enqueue = torch.cuda.mem_get_info()[0]
    backend.sync()
    free_after = torch.cuda.mem_get_info()[0]
    out = want.cpu().numpy().reshape(H, W, 3), got.cpu().numpy().reshape(H, W, 3)
    dev.close()
    return out, (free_before, free_after_enqueue, free_after)


def test_static_memory_render_equals_the_allocating_one(rt, backend, scene_path):
    """Two calls in a row, the second with a smaller input; then a frame far larger than the static array
    (rendered in many small batches), a scene with mirrors, stochastic sample offsets and the
    reference-order mode; and what the entry point refuses."""
    big = rt.parse_scene_file(scene_path("bunny_small"), image_width=400, image_height=300, recursion_depth=6)
    small = rt.parse_scene_file(scene_path("mixed_features"), image_width=96, image_height=64, recursion_depth=6)
    (want, got), mem = render_both(rt, backend, big, 3)
    assert np.array_equal(got, want) and got.min() >= 0.0
    assert mem[0] == mem[1] == mem[2], mem  # not a byte of device memory taken
    (want, got), mem = render_both(rt, backend, small, 2)  # smaller: the array still holds the first frame's data
    assert np.array_equal(got, want)
    assert mem[0] == mem[1] == mem[2], mem
    # 1000 x 1000 at 9 rays per pixel = 9 M samples, about 35 batches through a 64 MiB array
    frame = rt.parse_scene_file(scene_path("cornell_room"), image_width=1000, image_height=1000, recursion_depth=6)
    (want, got), _ = render_both(rt, backend, frame, 3, tile=(40, 40))
    assert np.array_equal(got, want)
    (want, got), _ = render_both(rt, backend, small, 3, antialias_method=1, seed=77)
    assert np.array_equal(got, want)
    (want, got), _ = render_both(rt, backend, small, 2, mode=rt.RT_MODE_REFERENCE_ORDER)
    assert np.array_equal(got, want)
    # per-sample ids come out the same
    import torch

    dev = backend.upload(small)
    ids_np = rt.RayIds(96 * 64 * 4, dev.n_point_lights)
    _, ids_ref, _ = dev.render(2, want_ids=True)
    prim = torch.full((96 * 64 * 4,), -1, dtype=torch.int32, device="cuda")
    fb = torch.zeros(96 * 64 * 3, dtype=torch.float32, device="cuda")
    ids = rt.RtRayIds()
    ids.primary_shape = prim.data_ptr()
    torch.cuda.synchronize()
    dev.render_device_static(dev.render_args(2), fb.data_ptr(), ids=ids)
    backend.sync()
    assert np.array_equal(prim.cpu().numpy(), ids_ref.primary_shape)
    # refused: a part of a frame; a tile that cannot fit the static array
    with pytest.raises(rt.RtError) as e:
        dev.render_device_static(dev.render_args(2, part=(0, 2)), fb.data_ptr())
    assert e.value.code == rt.RT_ERR_INVALID
    with pytest.raises(rt.RtError) as e:
        dev.render_device_static(dev.render_args(16, tile=(64, 64)), fb.data_ptr())  # 1 Mi samples in one tile
    assert e.value.code == rt.RT_ERR_INVALID and "static array" in e.value.message
    ok, _, _ = dev.render(2)  # the backend is still usable
    assert ok.pixels.any()
    dev.close()


def test_static_memory_frames_on_two_streams_do_not_overlap(rt, backend, scene_path):
    """All backends of a GPU share its one static array.  Frames enqueued back to back through two backends on
    two streams, with no ordering of the caller's between them, must still come out right: the library makes
    each frame wait (on the device) for the frame that used the array before it.  Small tiles, so that the
    frames run in many batches each and would trample each other's queues if they overlapped."""
    import torch

    scenes = [rt.parse_scene_file(scene_path("cornell_room"), image_width=480, image_height=360, recursion_depth=6),
              rt.parse_scene_file(scene_path("bunny_small"), image_width=400, image_height=300, recursion_depth=6)]
    other = rt.Backend(0)
    bes = [backend, other]
    devs = [bes[k].upload(scenes[k]) for k in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    args = [devs[k].render_args(3, tile=(16, 16), batch_samples=16 * 16 * 9 * 40) for k in range(2)]
    want = [torch.zeros(devs[k].width * devs[k].height * 3, dtype=torch.float32, device="cuda") for k in range(2)]
    got = [[torch.full_like(want[k], -1.0) for _ in range(3)] for k in range(2)]
    torch.cuda.synchronize()
    for k in range(2):
        devs[k].render_device(args[k], want[k].data_ptr())
        bes[k].sync()
    for r in range(3):
        for k in range(2):  # alternating backends and streams, nothing waited for in between
            devs[k].render_device_static(args[k], got[k][r].data_ptr(), stream=streams[k].cuda_stream)
    for k in range(2):
        bes[k].sync()
    torch.cuda.synchronize()
    for k in range(2):
        for r in range(3):
            assert torch.equal(got[k][r], want[k]), (k, r)
    for d in devs:
        d.close()
    other.close()
