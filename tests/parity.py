"""Comparison metrics shared by the parity tests, smoke() and the benchmark.

The bars come from the task statement of this backend:
  * primary and shadow hit ids identical on >= 99.99 % of rays,
  * final 8-bit RGB within +-1 LSB per channel on >= 99.9 % of pixels, PSNR >= 50 dB.
"""
from __future__ import annotations

import numpy as np

ID_MATCH_MIN = 0.9999
LSB_MATCH_MIN = 0.999
PSNR_MIN_DB = 50.0


def quantize(fb: np.ndarray) -> np.ndarray:
    """png_export.rs:6-8: round(clamp(c, 0, 1) * 255) in f32, half away from zero."""
    c = np.clip(fb.astype(np.float32), np.float32(0.0), np.float32(1.0)) * np.float32(255.0)
    return np.floor(c + np.float32(0.5)).astype(np.uint8)


def image_metrics(fb_a: np.ndarray, fb_b: np.ndarray) -> dict:
    qa, qb = quantize(fb_a).astype(np.int32), quantize(fb_b).astype(np.int32)
    diff = np.abs(qa - qb)
    within = (diff.max(axis=-1) <= 1).mean()
    mse = float((diff.astype(np.float64) ** 2).mean())
    psnr = float("inf") if mse == 0.0 else 10.0 * np.log10(255.0 ** 2 / mse)
    return {"lsb1_fraction": float(within), "psnr_db": psnr, "max_abs_diff": int(diff.max()),
            "max_float_diff": float(np.abs(fb_a.astype(np.float64) - fb_b.astype(np.float64)).max())}


def id_metrics(a, b, occlusion_only: bool = False) -> dict:
    """a, b: RayIds.  With occlusion_only, shadow entries are compared as
    "blocked / visible / no ray" (a stop-at-first-blocker traversal reports some
    blocker, not the closest one)."""
    out = {}
    n = a.primary_shape.size
    prim = (a.primary_shape == b.primary_shape) & (a.primary_tri == b.primary_tri)
    out["primary"] = float(prim.mean())
    fin = (a.final_shape == b.final_shape) & (a.final_tri == b.final_tri) & (a.bounces == b.bounces)
    out["final"] = float(fin.mean())
    if a.shadow_shape.size:
        cast = (a.shadow_shape != -2) | (b.shadow_shape != -2)
        if occlusion_only:
            sa = np.where(a.shadow_shape >= 0, 1, a.shadow_shape)
            sb = np.where(b.shadow_shape >= 0, 1, b.shadow_shape)
            same = sa == sb
        else:
            same = (a.shadow_shape == b.shadow_shape) & (a.shadow_tri == b.shadow_tri)
        out["shadow"] = float(same[cast].mean()) if cast.any() else 1.0
        out["shadow_rays_compared"] = int(cast.sum())
    else:
        out["shadow"] = 1.0
        out["shadow_rays_compared"] = 0
    out["samples"] = int(n)
    return out


def assert_parity(fb_gpu, ids_gpu, fb_ref, ids_ref, occlusion_only=False, label=""):
    im = image_metrics(fb_gpu, fb_ref)
    idm = id_metrics(ids_gpu, ids_ref, occlusion_only) if ids_gpu is not None else {}
    msg = f"{label}: image {im} ids {idm}"
    assert im["lsb1_fraction"] >= LSB_MATCH_MIN, msg
    assert im["psnr_db"] >= PSNR_MIN_DB, msg
    if idm:
        assert idm["primary"] >= ID_MATCH_MIN, msg
        assert idm["final"] >= ID_MATCH_MIN, msg
        assert idm["shadow"] >= ID_MATCH_MIN, msg
    return im, idm
