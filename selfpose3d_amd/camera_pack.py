"""Host-side preparation of the per-(sample, view) camera records the HIP kernels consume.

The reference does this work inside its batch x view Python loop, with ~6 device->host
syncs and one OpenCV call per (sample, view)
(/root/reference/lib/models/project_layer.py:64-75, lib/utils/transforms.py:61-103,
lib/utils/cameras.py:13-24).  Here the whole batch is packed in one vectorised numpy pass
into a (B, V, 64) fp32 table (layout: include/sp3d.h `SP3D_CAM_*`) and uploaded with ONE
host->device copy; the table is cached while the caller keeps passing the same ``meta``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

CAM_STRIDE = 64
# field offsets inside a record (keep in sync with include/sp3d.h)
CAM_R, CAM_T, CAM_F, CAM_C, CAM_K, CAM_P, CAM_A, CAM_W0, CAM_H0, CAM_FLIP = 0, 9, 12, 14, 16, 19, 21, 27, 28, 29
# derived block (include/sp3d.h), filled by finish(): what the packed projection reads, in its order
CAM_RXY, CAM_TXY, CAM_RZ, CAM_TZ, CAM_K2, CAM_TAME, CAM_P2, CAM_F2, CAM_C2, CAM_WH, CAM_AXY, CAM_FLIP2 = 32, 38, 40, 43, 44, 47, 48, 50, 52, 54, 56, 62


def _np(x, dtype=None):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    return x if dtype is None else x.astype(dtype)


def get_affine_transform_batch(center, scale, rot, output_size) -> np.ndarray:
    """Batched twin of ``get_affine_transform(center, scale, rot, output_size)`` (inv=0, shift=0).

    Follows /root/reference/lib/utils/transforms.py:61-103 step by step, including the
    float32 storage of the three src/dst points (:86-96) before the float64 3-point solve
    that ``cv2.getAffineTransform`` performs (:99-101).
      center (n,2) float64 | scale (n,2) float32 | rot (n,) degrees -> (n,2,3) float64
    """
    center = np.asarray(center, np.float64).reshape(-1, 2)
    n = center.shape[0]
    scale = np.asarray(scale, np.float32).reshape(n, 2)
    rot = np.asarray(rot, np.float64).reshape(n)
    scale_tmp = scale * np.float32(200.0)                                    # :75 (float32)
    src_w = scale_tmp[:, 0].astype(np.float64)
    src_h = scale_tmp[:, 1].astype(np.float64)
    dst_w, dst_h = float(output_size[0]), float(output_size[1])
    rot_rad = np.pi * rot / 180.0                                            # :79
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    wide = src_w >= src_h                                                    # :80
    spx = np.where(wide, 0.0, src_h * -0.5)
    spy = np.where(wide, src_w * -0.5, 0.0)
    src_dir = np.stack([spx * cs - spy * sn, spx * sn + spy * cs], 1)        # get_dir :131-138
    dst_dir = np.where(wide[:, None], np.array([0.0, dst_w * -0.5]), np.array([dst_h * -0.5, 0.0])).astype(np.float32)

    src = np.zeros((n, 3, 2), np.float32)
    dst = np.zeros((n, 3, 2), np.float32)
    src[:, 0] = center                                                       # :89
    src[:, 1] = center + src_dir                                             # :90
    dst[:, 0] = [dst_w * 0.5, dst_h * 0.5]                                   # :91
    dst[:, 1] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir               # :92

    def third(a, b):                                                         # get_3rd_point :126-128 (float32)
        d = a - b
        return b + np.stack([-d[:, 1], d[:, 0]], 1).astype(np.float32)

    src[:, 2] = third(src[:, 0], src[:, 1])
    dst[:, 2] = third(dst[:, 0], dst[:, 1])

    s = src.astype(np.float64)
    d = dst.astype(np.float64)
    A = np.zeros((n, 6, 6))
    rhs = np.zeros((n, 6))
    for i in range(3):
        A[:, 2 * i, 0], A[:, 2 * i, 1], A[:, 2 * i, 2] = s[:, i, 0], s[:, i, 1], 1.0
        A[:, 2 * i + 1, 3], A[:, 2 * i + 1, 4], A[:, 2 * i + 1, 5] = s[:, i, 0], s[:, i, 1], 1.0
        rhs[:, 2 * i] = d[:, i, 0]
        rhs[:, 2 * i + 1] = d[:, i, 1]
    return np.linalg.solve(A, rhs[:, :, None])[:, :, 0].reshape(n, 2, 3)


def pack_cameras(meta: Sequence[dict], batch: int, img_size: Sequence[int],
                 flip_xcoords: Optional[torch.Tensor] = None) -> np.ndarray:
    """meta: list[V] of collated dicts (App. B of SURVEY.md) -> (B, V, 64) float32 table.

    Field semantics per reference line:
      R,T,f,c,k,p  fp32 casts of the camera dict          (cameras.py:13-24)
      A            fp32 cast of get_affine_transform(...) (project_layer.py:69-72)
      W0,H0        ``center * 2`` compared in fp32         (project_layer.py:68,78-79)
      flip         ``flip_xcoords[i]``                     (project_layer.py:82)
    """
    V = len(meta)
    B = int(batch)
    tab = np.zeros((B, V, CAM_STRIDE), np.float32)
    flips = None if flip_xcoords is None else _np(flip_xcoords).astype(bool).reshape(B)
    # one pass over all B x V records (the per-step host cost of a graphed step is this function: the per-view form, ~35
    # small numpy calls and one 6x6 solve per view, took 0.79 ms for 5 views x 4 samples; this one 0.3 ms)

    def field(get, width, dtype=np.float32):                                              # -> (B, V, width)
        vals = [get(m) for m in meta]
        if all(isinstance(v, torch.Tensor) and v.dtype == vals[0].dtype and v.device.type == "cpu" for v in vals):
            # a DataLoader's collated tensors: ONE stack for the V views instead of V conversions (same values: the cast to
            # `dtype` happens on the same source numbers either way)
            return torch.stack([v.detach().reshape(B, width) for v in vals], 1).numpy().astype(dtype, copy=False)
        return np.stack([_np(v, dtype).reshape(B, width) for v in vals], 1)
    center = field(lambda m: m["center"], 2, np.float64)
    rot = field(lambda m: m["rotation"], 1, np.float64)
    scales = []
    for m in meta:
        sc = _np(m["scale"]).reshape(B, -1)
        scales.append(np.repeat(sc, 2, 1) if sc.shape[1] == 1 else sc)          # scalar scale -> [s, s] (transforms.py:72-73)
    scale = np.stack(scales, 1).astype(np.float32)
    A = get_affine_transform_batch(center.reshape(B * V, 2), scale.reshape(B * V, 2), rot.reshape(B * V), img_size)
    tab[:, :, CAM_R:CAM_R + 9] = field(lambda m: m["camera"]["R"], 9)
    tab[:, :, CAM_T:CAM_T + 3] = field(lambda m: m["camera"]["T"], 3)
    tab[:, :, CAM_F:CAM_F + 1] = field(lambda m: m["camera"]["fx"], 1)
    tab[:, :, CAM_F + 1:CAM_F + 2] = field(lambda m: m["camera"]["fy"], 1)
    tab[:, :, CAM_C:CAM_C + 1] = field(lambda m: m["camera"]["cx"], 1)
    tab[:, :, CAM_C + 1:CAM_C + 2] = field(lambda m: m["camera"]["cy"], 1)
    tab[:, :, CAM_K:CAM_K + 3] = field(lambda m: m["camera"]["k"], 3)
    tab[:, :, CAM_P:CAM_P + 2] = field(lambda m: m["camera"]["p"], 2)
    tab[:, :, CAM_A:CAM_A + 6] = A.astype(np.float32).reshape(B, V, 6)
    tab[:, :, CAM_W0:CAM_H0 + 1] = (center * 2.0).astype(np.float32)
    if flips is not None:
        tab[:, :, CAM_FLIP] = flips.astype(np.float32)[:, None]
    return finish(tab)


def finish(tab: np.ndarray) -> np.ndarray:
    """fill the DERIVED fields of camera records (..., 48) in place from their fields 0..29 - the numpy twin of the C ABI's
    sp3d_camera_finish (include/sp3d.h): the packed-fp32 projection's operand pairs as aligned neighbours, and the result of
    its per-view affine sanity test (every |A| <= 1e30 as an integer comparison of the bit patterns: NaN / inf fail).
    Call it again after editing a record by hand."""
    t = tab.reshape(-1, CAM_STRIDE)
    for c in range(3):
        t[:, CAM_RXY + 2 * c] = t[:, CAM_R + c]
        t[:, CAM_RXY + 2 * c + 1] = t[:, CAM_R + 3 + c]
        t[:, CAM_AXY + 2 * c] = t[:, CAM_A + c]
        t[:, CAM_AXY + 2 * c + 1] = t[:, CAM_A + 3 + c]
    t[:, CAM_RZ:CAM_RZ + 3] = t[:, CAM_R + 6:CAM_R + 9]
    t[:, CAM_K2:CAM_K2 + 3] = t[:, CAM_K:CAM_K + 3]
    t[:, CAM_TXY:CAM_TXY + 2] = t[:, CAM_T:CAM_T + 2]
    t[:, CAM_TZ] = t[:, CAM_T + 2]
    t[:, CAM_P2:CAM_P2 + 2] = t[:, CAM_P:CAM_P + 2]
    t[:, CAM_F2:CAM_F2 + 2] = t[:, CAM_F:CAM_F + 2]
    t[:, CAM_C2:CAM_C2 + 2] = t[:, CAM_C:CAM_C + 2]
    t[:, CAM_WH] = t[:, CAM_W0]
    t[:, CAM_WH + 1] = t[:, CAM_H0]
    t[:, CAM_FLIP2] = t[:, CAM_FLIP]
    mag = np.ascontiguousarray(t[:, CAM_A:CAM_A + 6]).view(np.uint32) & np.uint32(0x7FFFFFFF)
    t[:, CAM_TAME] = (mag.max(axis=1) <= np.uint32(0x7149F2CA)).astype(np.float32)
    t[:, 30:32] = 0.0
    t[:, 63] = 0.0
    return tab


def _content(t) -> bytes:
    a = _np(t)
    return a.dtype.str.encode() + a.tobytes()


def meta_cache_key(meta: Sequence[dict], flip_xcoords, img_size) -> tuple:
    """CONTENT key of everything ``pack_cameras`` reads: the raw bytes of the few-KB calibration / crop arrays
    (~45 us for 5 views x 4 samples, a tenth of the pack itself).  Identity-based keys (id / data_ptr / _version)
    cannot see in-place edits of numpy entries and can collide when a freed batch's addresses are reused; bytes
    cannot."""
    key: List = [tuple(int(v) for v in img_size)]
    for m in meta:
        parts = [_content(m["center"]), _content(m["scale"]), _content(m["rotation"])]
        parts += [_content(m["camera"][k]) for k in ("R", "T", "fx", "fy", "cx", "cy", "k", "p")]
        key.append(b"|".join(parts))
    key.append(None if flip_xcoords is None else _content(flip_xcoords))
    return tuple(key)
