"""Host-side mirror of the reference's public API over the C ABI of librt_hip.so.

The reference crate exports (raytracer-lib/src/lib.rs:17-23)

    parse_scene(json, data_path, width, height, aspect, depth, disable_shadows, render_normals) -> Scene
    render(&Scene, sqrt_rays_per_pixel, AntialiasMethod, per_pixel_cb) -> Framebuffer

This module keeps those two names, argument meanings and error behaviour
(`parse_scene` raises where the reference returns Err or panics; `render` has no
error path of its own besides device failures) and forwards everything to the
library declared in include/rt_hip.h.  There is no CPU implementation here: if
the HIP library cannot be loaded, or no GPU is present, calls raise.

Python is only the harness language of this repository (tests, benchmark); the
boundary a Rust caller would bind is the C header.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIB points development builds (kernel variants being timed against each other) at another
# build of the same library; it never selects a different implementation.
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "csrc", "librt_hip.so")

RT_OK = 0
RT_ERR_INVALID = -1
RT_ERR_PARSE = -2
RT_ERR_UNSUPPORTED = -3
RT_ERR_DEVICE = -4
RT_ERR_IO = -5

RT_MODE_FAST = 0
RT_MODE_REFERENCE_ORDER = 1
RT_RENDER_NO_PACKETS = 1
RT_RENDER_ROW_MAJOR_PIXELS = 2
RT_RENDER_UNFUSED_SHADOWS = 4
RT_RENDER_UNFUSED_CAMERA = 8
RT_RENDER_TREE_DEPTH_FIRST = 16
RT_RENDER_CLOSEST_BLOCKERS = 32
RT_UPLOAD_REFERENCE_MESH_TREES = 1
RT_COMM_ID_BYTES = 128
RT_STATIC_SCRATCH_BYTES = 64 << 20
RT_K_COUNT = 8
KERNEL_CLASSES = ("raygen", "trace_closest", "shade", "trace_shadow", "accumulate", "resolve")


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rt error {code}: {message}")
        self.code = code
        self.message = message


class AntialiasMethod(enum.IntEnum):
    """antialias.rs:6-10"""

    Normal = 0
    Jittered = 1
    Random = 2


# ---- ctypes mirrors of include/rt_scene.h (only what the harness reads) -------------


class RtVec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class RtCamera(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("pixels_x", C.c_uint32),
        ("pixels_y", C.c_uint32),
        ("_pad", C.c_uint32),
        ("position", RtVec3),
        ("u", RtVec3),
        ("v", RtVec3),
        ("w", RtVec3),
        ("left", C.c_double),
        ("right", C.c_double),
        ("top", C.c_double),
        ("bottom", C.c_double),
        ("focal_length", C.c_double),
    ]


class RtShader(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("diffuse", C.c_float * 3),
        ("specular", C.c_float * 3),
        ("shininess", C.c_float),
        ("roughness", C.c_double),
        ("samples", C.c_uint32),
        ("_pad", C.c_uint32),
    ]


class RtLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("intensity", C.c_float * 3), ("position", RtVec3)]


class RtShape(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("shader", C.c_int32),
        ("mesh", C.c_int32),
        ("proto", C.c_int32),
        ("p0", RtVec3),
        ("p1", RtVec3),
        ("p2", RtVec3),
        ("radius", C.c_double),
        ("inv_transform", C.c_double * 16),
        ("normal_matrix", C.c_double * 16),
        ("bbox_min", RtVec3),
        ("bbox_max", RtVec3),
    ]


class RtMesh(C.Structure):
    _fields_ = [("n_triangles", C.c_uint64), ("vertices", C.POINTER(C.c_double))]


class RtScene(C.Structure):
    _fields_ = [
        ("camera", RtCamera),
        ("background", C.c_float * 3),
        ("recursion_depth", C.c_uint16),
        ("disable_shadows", C.c_uint8),
        ("_pad0", C.c_uint8),
        ("image_width", C.c_uint32),
        ("image_height", C.c_uint32),
        ("n_shapes", C.c_uint32),
        ("n_protos", C.c_uint32),
        ("n_meshes", C.c_uint32),
        ("n_shaders", C.c_uint32),
        ("n_lights", C.c_uint32),
        ("_pad1", C.c_uint32),
        ("shapes", C.POINTER(RtShape)),
        ("protos", C.POINTER(RtShape)),
        ("meshes", C.POINTER(RtMesh)),
        ("shaders", C.POINTER(RtShader)),
        ("lights", C.POINTER(RtLight)),
    ]


class RtParseArgs(C.Structure):
    _fields_ = [
        ("image_width", C.c_uint32),
        ("image_height", C.c_uint32),
        ("aspect_ratio", C.c_double),
        ("recursion_depth", C.c_int32),
        ("disable_shadows", C.c_uint8),
        ("render_normals", C.c_uint8),
        ("_pad", C.c_uint8 * 2),
    ]


# void (*progress)(uint64_t pixels_done, uint64_t pixels_total, void *user)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_uint64, C.c_uint64, C.c_void_p)


class RtRenderArgs(C.Structure):
    _fields_ = [
        ("sqrt_rays_per_pixel", C.c_uint16),
        ("timed_classes", C.c_uint16),
        ("antialias_method", C.c_int32),
        ("mode", C.c_int32),
        ("tile_w", C.c_uint32),
        ("tile_h", C.c_uint32),
        ("part_index", C.c_uint32),
        ("part_count", C.c_uint32),
        ("part0_weight", C.c_uint32),
        ("seed", C.c_uint64),
        ("progress", C.c_void_p),
        ("progress_user", C.c_void_p),
        ("batch_samples", C.c_uint32),
        ("xcd_chunk", C.c_uint32),
        ("flags", C.c_uint32),
        ("part_rotation", C.c_uint32),
        ("window_x0", C.c_uint32),
        ("window_y0", C.c_uint32),
        ("window_w", C.c_uint32),
        ("window_h", C.c_uint32),
        ("part_span", C.c_uint32),
        ("_reserved0", C.c_uint32),
        ("part_span_pitch_floats", C.c_uint64),
    ]


class RtUploadOptions(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("_reserved", C.c_uint32 * 3)]


class RtRayIds(C.Structure):
    _fields_ = [
        ("primary_shape", C.c_void_p),
        ("primary_tri", C.c_void_p),
        ("final_shape", C.c_void_p),
        ("final_tri", C.c_void_p),
        ("bounces", C.c_void_p),
        ("shadow_shape", C.c_void_p),
        ("shadow_tri", C.c_void_p),
    ]


ID_FIELDS = ("primary_shape", "primary_tri", "final_shape", "final_tri", "bounces", "shadow_shape", "shadow_tri")


class RtStats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64),
        ("secondary_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("shaded_hits", C.c_uint64),
        ("render_ms", C.c_double),
        ("kernel_ms", C.c_double * RT_K_COUNT),
        ("kernel_launches", C.c_uint32 * RT_K_COUNT),
    ]

    def total_rays(self) -> int:
        return int(self.primary_rays + self.secondary_rays + self.shadow_rays)


_lib = None


def _share_hip_runtime_with_torch() -> None:
    """A PyTorch wheel bundles a HIP runtime of its own (torch/lib/libamdhip64.so, same soname as the
    system's).  A process must hold ONE runtime: if librt_hip.so is loaded first it brings in the
    system's, and torch, imported afterwards, finds no GPU.  So when torch is installed but not yet
    imported, its copy is loaded first and librt_hip.so binds to it, exactly as when torch is imported
    first.  Programs without torch (rt-cli) use the system runtime."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    bundled = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        try:
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> C.CDLL:
    """Loads librt_hip.so (built in-tree by `make -C rust-raytracer_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(RT_ERR_DEVICE, f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}`")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rt_last_error.restype = C.c_char_p
    L.rt_parse_scene.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(RtParseArgs), C.POINTER(C.POINTER(RtScene))]
    L.rt_parse_scene_file.argtypes = [C.c_char_p, C.POINTER(RtParseArgs), C.POINTER(C.POINTER(RtScene))]
    L.rt_scene_destroy.argtypes = [C.POINTER(RtScene)]
    L.rt_scene_destroy.restype = None
    L.rt_backend_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rt_backend_destroy.argtypes = [vp]
    L.rt_backend_destroy.restype = None
    L.rt_scene_upload.argtypes = [vp, C.POINTER(RtScene), C.POINTER(vp)]
    L.rt_scene_free.argtypes = [vp]
    L.rt_scene_free.restype = None
    L.rt_render.argtypes = [vp, vp, C.POINTER(RtRenderArgs), vp, C.POINTER(RtRayIds), C.POINTER(RtStats)]
    L.rt_render_device.argtypes = [vp, vp, C.POINTER(RtRenderArgs), vp, C.POINTER(RtRayIds), vp, C.POINTER(RtStats)]
    L.rt_scene_upload_with.argtypes = [vp, C.POINTER(RtScene), C.POINTER(RtUploadOptions), C.POINTER(vp)]
    L.rt_render_device_static.argtypes = [vp, vp, C.POINTER(RtRenderArgs), vp, C.POINTER(RtRayIds), vp]
    L.rt_render_async.argtypes = [vp, vp, C.POINTER(RtRenderArgs), vp]
    L.rt_backend_sync.argtypes = [vp]
    L.rt_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.rt_host_free.argtypes = [vp]
    L.rt_host_free.restype = None
    L.rt_multi_create.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_uint32, C.POINTER(vp)]
    L.rt_multi_destroy.argtypes = [vp]
    L.rt_multi_destroy.restype = None
    L.rt_render_multi.argtypes = [vp, C.POINTER(RtRenderArgs), vp, C.POINTER(RtStats)]
    L.rt_comm_id.argtypes = [vp]
    L.rt_comm_rccl_version.argtypes = [C.POINTER(C.c_int)]
    L.rt_comm_create.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.rt_comm_destroy.argtypes = [vp]
    L.rt_comm_destroy.restype = None
    L.rt_comm_gather.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint32, vp]
    L.rt_comm_gather_frames.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.rt_collect_timings.argtypes = [vp, C.POINTER(RtStats)]
    L.rt_trace_rays.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, vp, C.c_int32, vp, vp, vp]
    L.rt_partition_info.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(RtRenderArgs), C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.rt_partition_tiles.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(RtRenderArgs), vp, C.c_uint32,
                                     C.POINTER(C.c_uint32)]
    L.rt_untile_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(RtRenderArgs), vp, vp, vp]
    L.rt_untile_device_pitched.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(RtRenderArgs), vp, C.c_size_t, vp, vp]
    L.rt_framebuffer_to_rgb8.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.rt_save_png.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.rt_scene_point_light_count.argtypes = [vp]
    L.rt_scene_point_light_count.restype = C.c_uint32
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error().decode("utf-8", "replace"))


def _share_rccl_with_torch() -> None:
    """The communicators below load librccl by its soname when the first one is made.  A PyTorch wheel
    bundles a copy (torch/lib/librccl.so) built against its bundled HIP runtime; when that runtime is
    the one this process holds (see _share_hip_runtime_with_torch), its RCCL must be the one as well.
    Importing torch BEFORE the first communicator is made takes care of that, and of the order of
    set-up and tear-down between the two: a process that made a communicator first and imported torch
    afterwards ended with "double free or corruption" when it exited (seen on MI355X, ROCm 7.2 /
    torch 2.10)."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is not None:
        importlib.import_module("torch")


class Multi:
    """rt_multi_create / rt_render_multi: one frame split over the given DeviceScenes (one backend each),
    the parts' tiles gathered on the first one's device over RCCL."""

    def __init__(self, dev_scenes):
        lib()
        _share_rccl_with_torch()
        n = len(dev_scenes)
        self.dev_scenes = list(dev_scenes)
        bes = (C.c_void_p * n)(*[d.backend.handle for d in dev_scenes])
        dss = (C.c_void_p * n)(*[d._h for d in dev_scenes])
        self._h = C.c_void_p()
        _check(lib().rt_multi_create(bes, dss, n, C.byref(self._h)))

    def render(self, args: "RtRenderArgs", out: Optional[np.ndarray] = None, want_stats: bool = True):
        """Returns (pixels [H, W, 3] float32, RtStats | None).  `out` may be a page-locked array (HostBuffer)."""
        W, H = self.dev_scenes[0].width, self.dev_scenes[0].height
        fb = np.zeros((H, W, 3), np.float32) if out is None else out
        stats = RtStats() if want_stats else None
        _check(lib().rt_render_multi(self._h, C.byref(args), fb.ctypes.data, C.byref(stats) if stats is not None else None))
        return fb, stats

    def close(self) -> None:
        if self._h:
            lib().rt_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """rt_comm_*: this process's end of the RCCL communicator of a one-process-per-GPU launch."""

    @staticmethod
    def rccl_version() -> int:
        """rt_comm_rccl_version: ncclGetVersion of the RCCL the library's communicators use."""
        lib()
        _share_rccl_with_torch()
        v = C.c_int()
        _check(lib().rt_comm_rccl_version(C.byref(v)))
        return int(v.value)

    @staticmethod
    def make_id() -> bytes:
        lib()
        _share_rccl_with_torch()
        buf = (C.c_uint8 * RT_COMM_ID_BYTES)()
        _check(lib().rt_comm_id(buf))
        return bytes(buf)

    def __init__(self, backend: "Backend", comm_id: bytes, rank: int, world: int):
        lib()
        _share_rccl_with_torch()
        assert len(comm_id) == RT_COMM_ID_BYTES
        buf = (C.c_uint8 * RT_COMM_ID_BYTES).from_buffer_copy(comm_id)
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        _check(lib().rt_comm_create(backend.handle, buf, rank, world, C.byref(self._h)))

    def gather(self, part_dev_ptr: int, floats: int, gathered_dev_ptr: int, root: int = 0, stream: int = 0) -> None:
        _check(lib().rt_comm_gather(self._h, part_dev_ptr, floats, gathered_dev_ptr or None, root, stream or None))

    def gather_frames(self, parts_dev_ptr: int, floats: int, count: int, first_root: int, slots: int,
                      gathered_dev_ptr: int, stream: int = 0) -> None:
        """rt_comm_gather_frames: `count` frames, frame f to rank (first_root + f) % world; see frame_blocks."""
        _check(lib().rt_comm_gather_frames(self._h, parts_dev_ptr, floats, count, first_root, slots, gathered_dev_ptr or None,
                                           stream or None))

    @staticmethod
    def frame_blocks(world: int, slots: int, f: int, first_root: int = 0):
        """Where frame f of a group lives (rt_comm_gather_frames' layout): (root rank, block index of this
        rank's part in the send buffer, block index of rank 0's part in the root's gather buffer); rank r's
        part lies r * slots blocks further on."""
        p, s = f % world, f // world
        if s >= slots:
            raise ValueError("frame does not fit: slots * world < frames of the group")
        return (first_root + p) % world, p * slots + s, s

    def close(self) -> None:
        if self._h:
            lib().rt_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostBuffer:
    """rt_host_alloc: page-locked host memory as a float32 numpy array of the given shape."""

    def __init__(self, shape):
        self.shape = tuple(int(x) for x in shape)
        count = int(np.prod(self.shape))
        self._p = C.c_void_p()
        _check(lib().rt_host_alloc(count * 4, C.byref(self._p)))
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_float)), shape=(count,)).reshape(self.shape)

    @property
    def ptr(self) -> int:
        return self._p.value

    def close(self) -> None:
        if self._p:
            self.array = None
            lib().rt_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def partition_tiles(width: int, height: int, args: "RtRenderArgs") -> np.ndarray:
    """Tile indices owned by args.part_index (rt_partition_tiles); host-only, needs no GPU."""
    n = C.c_uint32()
    _check(lib().rt_partition_tiles(width, height, C.byref(args), None, 0, C.byref(n)))
    out = np.empty(n.value, np.int32)
    _check(lib().rt_partition_tiles(width, height, C.byref(args), out.ctypes.data, n.value, C.byref(n)))
    return out


class Scene:
    """Owner of an RtScene built by the loader (the reference's `Scene`, scene/mod.rs:14-25)."""

    def __init__(self, ptr):
        self._ptr = ptr

    @property
    def c(self) -> RtScene:
        return self._ptr.contents

    @property
    def ptr(self):
        return self._ptr

    @property
    def image_width(self) -> int:
        return int(self.c.image_width)

    @property
    def image_height(self) -> int:
        return int(self.c.image_height)

    @property
    def n_point_lights(self) -> int:
        return sum(1 for i in range(self.c.n_lights) if self.c.lights[i].type == 0)

    def close(self) -> None:
        if self._ptr is not None:
            lib().rt_scene_destroy(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _parse_args(image_width, image_height, aspect_ratio, recursion_depth, disable_shadows, render_normals) -> RtParseArgs:
    a = RtParseArgs()
    a.image_width = int(image_width or 0)
    a.image_height = int(image_height or 0)
    a.aspect_ratio = float(aspect_ratio) if aspect_ratio else 0.0
    a.recursion_depth = -1 if recursion_depth is None else int(recursion_depth)
    a.disable_shadows = 1 if disable_shadows else 0
    a.render_normals = 1 if render_normals else 0
    return a


def parse_scene(
    scene_json: str,
    scene_data_path: str,
    image_width: Optional[int] = None,
    image_height: Optional[int] = None,
    aspect_ratio: Optional[float] = None,
    recursion_depth: Optional[int] = None,
    disable_shadows: bool = False,
    render_normals: bool = False,
) -> Scene:
    """parse_scene -- scene/mod.rs:378-387 (same argument order and defaults)."""
    a = _parse_args(image_width, image_height, aspect_ratio, recursion_depth, disable_shadows, render_normals)
    out = C.POINTER(RtScene)()
    _check(lib().rt_parse_scene(scene_json.encode(), scene_data_path.encode(), C.byref(a), C.byref(out)))
    return Scene(out)


def parse_scene_file(scene_path: str, **kw) -> Scene:
    """The CLI's read + parse sequence -- raytracer-cli/src/main.rs:51-67."""
    a = _parse_args(
        kw.get("image_width"), kw.get("image_height"), kw.get("aspect_ratio"), kw.get("recursion_depth"),
        kw.get("disable_shadows", False), kw.get("render_normals", False),
    )
    out = C.POINTER(RtScene)()
    _check(lib().rt_parse_scene_file(scene_path.encode(), C.byref(a), C.byref(out)))
    return Scene(out)


class Framebuffer:
    """framebuffer.rs:4-8: width, height, pixels[i + j*width] = [r, g, b], row 0 at the bottom."""

    def __init__(self, width: int, height: int, pixels: Optional[np.ndarray] = None):
        self.width = width
        self.height = height
        self.pixels = np.zeros((height, width, 3), np.float32) if pixels is None else pixels

    def get_pixel(self, i: int, j: int) -> np.ndarray:
        return self.pixels[j, i]

    def to_rgb8(self) -> np.ndarray:
        """save_to_png's conversion (raytracer-cli/src/output/png_export.rs:4-10): top-down rows of u8."""
        out = np.empty((self.height, self.width, 3), np.uint8)
        src = np.ascontiguousarray(self.pixels, np.float32)
        _check(lib().rt_framebuffer_to_rgb8(src.ctypes.data, self.width, self.height, out.ctypes.data))
        return out


class RayIds:
    """Per-sample id arrays (RtRayIds), host numpy."""

    def __init__(self, n_samples: int, n_point_lights: int):
        self.primary_shape = np.full(n_samples, -1, np.int32)
        self.primary_tri = np.full(n_samples, -1, np.int32)
        self.final_shape = np.full(n_samples, -1, np.int32)
        self.final_tri = np.full(n_samples, -1, np.int32)
        self.bounces = np.zeros(n_samples, np.int32)
        self.shadow_shape = np.full((n_samples, max(n_point_lights, 0)), -2, np.int32)
        self.shadow_tri = np.full((n_samples, max(n_point_lights, 0)), -2, np.int32)

    def as_struct(self) -> RtRayIds:
        s = RtRayIds()
        for f in ID_FIELDS:
            arr = getattr(self, f)
            setattr(s, f, arr.ctypes.data if arr.size else None)
        return s


class Backend:
    """One GPU: rt_backend_create / rt_backend_destroy."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().rt_backend_create(device, C.byref(self._h)))
        self.device = device

    @property
    def handle(self):
        return self._h

    def collect_timings(self) -> RtStats:
        """rt_collect_timings: device times of the frames enqueued with RtRenderArgs.timed_classes."""
        st = RtStats()
        _check(lib().rt_collect_timings(self._h, C.byref(st)))
        return st

    def upload(self, scene: Scene, flags: int = 0) -> "DeviceScene":
        """rt_scene_upload; with `flags` (RT_UPLOAD_*) rt_scene_upload_with."""
        h = C.c_void_p()
        if flags:
            opts = RtUploadOptions()
            opts.flags = int(flags)
            _check(lib().rt_scene_upload_with(self._h, scene.ptr, C.byref(opts), C.byref(h)))
        else:
            _check(lib().rt_scene_upload(self._h, scene.ptr, C.byref(h)))
        return DeviceScene(self, scene, h)

    def sync(self) -> None:
        """rt_backend_sync: waits for what was enqueued through this backend and raises if a frame failed on the device."""
        _check(lib().rt_backend_sync(self._h))

    def close(self) -> None:
        if self._h:
            lib().rt_backend_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceScene:
    def __init__(self, backend: Backend, scene: Scene, handle):
        self.backend = backend
        self.scene = scene
        self._h = handle
        self.width = scene.image_width
        self.height = scene.image_height
        self.n_point_lights = int(lib().rt_scene_point_light_count(handle))

    @property
    def handle(self):
        return self._h

    def close(self) -> None:
        if self._h:
            if getattr(self, "_lender", None) is None:
                lib().rt_scene_free(self._h)
            self._h = C.c_void_p()

    def on(self, backend: Backend) -> "DeviceScene":
        """The same uploaded scene, rendered through another backend of the same GPU: a scene on the device is only
        read, so frames in flight on several backends can share one copy of the trees (and the caches hold one).
        The view does not own the scene: close the original after the last frame through any view."""
        view = DeviceScene.__new__(DeviceScene)
        view.__dict__.update(self.__dict__)
        view.backend, view._lender = backend, self
        return view

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_args(self, sqrt_rays_per_pixel, antialias_method=AntialiasMethod.Normal, mode=RT_MODE_FAST,
                    tile=(0, 0), part=(0, 1), seed=0, part0_weight=0, batch_samples=0, xcd_chunk=0, flags=0,
                    part_rotation=0, window=None, part_span=0, part_span_pitch_floats=0) -> RtRenderArgs:
        a = RtRenderArgs()
        a.part_span, a.part_span_pitch_floats = int(part_span), int(part_span_pitch_floats)
        if window is not None:  # (x0, y0, w, h): only those pixels, render_pixel's counterpart (render.rs:53-84)
            a.window_x0, a.window_y0, a.window_w, a.window_h = (int(v) for v in window)
        a.batch_samples, a.xcd_chunk, a.flags = int(batch_samples), int(xcd_chunk), int(flags)
        a.part_rotation = int(part_rotation)
        a.sqrt_rays_per_pixel = int(sqrt_rays_per_pixel)
        a.antialias_method = int(antialias_method)
        a.mode = int(mode)
        a.tile_w, a.tile_h = int(tile[0]), int(tile[1])
        a.part_index, a.part_count = int(part[0]), int(part[1])
        a.part0_weight = int(part0_weight)
        a.seed = int(seed)
        return a

    def render(self, sqrt_rays_per_pixel: int, antialias_method=AntialiasMethod.Normal, mode=RT_MODE_FAST,
               want_ids: bool = False, tile=(0, 0), part=(0, 1), seed=0, progress=None, part0_weight=0,
               batch_samples=0, flags=0, want_stats: bool = True, out: Optional[np.ndarray] = None, window=None,
               part_span=0):
        """render(&scene, sqrt_rpp, aa, cb) -- render.rs:7-23.  Returns (Framebuffer, RayIds|None, RtStats|None).
        `seed` keys the sample offsets of the Jittered / Random methods (antialias.rs:45-57);
        `progress(pixels_done, pixels_total)` is called after every finished batch of tiles.
        `want_stats=False` passes no RtStats: the frame is then enqueued without an event pair around
        every kernel and waited for once.  `out`: render into this [H, W, 3] float32 array (e.g. a
        HostBuffer's) instead of a fresh one.  `window` = (x0, y0, w, h): render_pixel's counterpart
        (render.rs:53-84) -- only those pixels, and the framebuffer and ids returned cover the window."""
        a = self.render_args(sqrt_rays_per_pixel, antialias_method, mode, tile, part, seed, part0_weight,
                             batch_samples=batch_samples, flags=flags, window=window, part_span=part_span)
        out_w, out_h = (int(window[2]), int(window[3])) if window is not None and window[2] and window[3] else (self.width, self.height)
        keep = None
        if progress is not None:
            keep = PROGRESS_FN(lambda done, total, _user: progress(int(done), int(total)))
            a.progress = C.cast(keep, C.c_void_p)
        fb = Framebuffer(out_w, out_h, out)
        ids = RayIds(out_w * out_h * int(sqrt_rays_per_pixel) ** 2, self.n_point_lights) if want_ids else None
        ids_struct = ids.as_struct() if ids is not None else None
        stats = RtStats() if want_stats else None
        _check(
            lib().rt_render(
                self.backend.handle, self._h, C.byref(a), fb.pixels.ctypes.data,
                C.byref(ids_struct) if ids_struct is not None else None, C.byref(stats) if stats is not None else None,
            )
        )
        return fb, ids, stats

    def render_async(self, args: RtRenderArgs, host_buffer: "HostBuffer") -> None:
        """rt_render_async: enqueue the frame and its copy into page-locked memory; Backend.sync() waits."""
        _check(lib().rt_render_async(self.backend.handle, self._h, C.byref(args), host_buffer.ptr))

    def partition_info(self, args: RtRenderArgs):
        """(tiles_total, tiles_owned, tiles_owned_max) of a partitioned render."""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().rt_partition_info(self.width, self.height, C.byref(args), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def render_device(self, args: RtRenderArgs, fb_dev_ptr: int, stream: int = 0, ids: Optional[RtRayIds] = None,
                      want_stats: bool = False):
        """rt_render_device: `fb_dev_ptr` is a device address (e.g. a torch tensor's data_ptr())."""
        stats = RtStats() if want_stats else None
        _check(
            lib().rt_render_device(
                self.backend.handle, self._h, C.byref(args), fb_dev_ptr, C.byref(ids) if ids is not None else None,
                stream or None, C.byref(stats) if stats is not None else None,
            )
        )
        return stats

    def render_device_static(self, args: RtRenderArgs, fb_dev_ptr: int, stream: int = 0, ids: Optional[RtRayIds] = None) -> None:
        """rt_render_device_static: the same frame with all work memory in the library's static device array."""
        _check(lib().rt_render_device_static(self.backend.handle, self._h, C.byref(args), fb_dev_ptr,
                                             C.byref(ids) if ids is not None else None, stream or None))

    def untile_device(self, args: RtRenderArgs, gathered_dev_ptr: int, fb_dev_ptr: int, stream: int = 0,
                      part_pitch_floats: int = 0) -> None:
        """rt_untile_device; with `part_pitch_floats` the parts' blocks lie that far apart (rt_untile_device_pitched)."""
        if part_pitch_floats:
            _check(lib().rt_untile_device_pitched(self.backend.handle, self.width, self.height, C.byref(args),
                                                  gathered_dev_ptr, part_pitch_floats, fb_dev_ptr, stream or None))
        else:
            _check(lib().rt_untile_device(self.backend.handle, self.width, self.height, C.byref(args), gathered_dev_ptr,
                                          fb_dev_ptr, stream or None))

    def trace_rays(self, origins, directions, tmin, tmax, mode=RT_MODE_FAST):
        """BVH::closest_hit (geometry/bvh.rs:140) on caller-supplied rays."""
        o = np.ascontiguousarray(origins, np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float64).reshape(-1, 3)
        n = o.shape[0]
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, np.float64), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float64), (n,)))
        t = np.empty(n, np.float32)
        shape = np.empty(n, np.int32)
        tri = np.empty(n, np.int32)
        _check(
            lib().rt_trace_rays(
                self.backend.handle, self._h, n, o.ctypes.data, d.ctypes.data, tmin.ctypes.data, tmax.ctypes.data,
                int(mode), t.ctypes.data, shape.ctypes.data, tri.ctypes.data,
            )
        )
        return t, shape, tri


def render(scene: Scene, sqrt_rays_per_pixel: int, antialias_method=AntialiasMethod.Normal, per_pixel_cb=None,
           device: int = 0, mode: int = RT_MODE_FAST, seed: int = 0) -> Framebuffer:
    """Drop-in for raytracer_lib::render (render.rs:7-12).

    `per_pixel_cb` is called once per pixel of the frame, with no arguments, on the calling
    thread, as the reference does (render.rs:80-82, where it advances a progress bar) -- but in
    bursts: a GPU frame finishes in batches of tiles, and the calls for a batch's pixels are made
    when the batch is done.
    """
    state = {"done": 0}

    def progress(done, total):
        for _ in range(done - state["done"]):
            per_pixel_cb()
        state["done"] = done

    be = Backend(device)
    try:
        ds = be.upload(scene)
        try:
            fb, _, _ = ds.render(sqrt_rays_per_pixel, antialias_method, mode, seed=seed,
                                 progress=progress if per_pixel_cb is not None else None)
        finally:
            ds.close()
    finally:
        be.close()
    return fb


def save_png(path: str, fb: Framebuffer) -> None:
    """output::save -- raytracer-cli/src/output/mod.rs:7-17 (only .png is supported there too)."""
    src = np.ascontiguousarray(fb.pixels, np.float32)
    _check(lib().rt_save_png(path.encode(), src.ctypes.data, fb.width, fb.height))
