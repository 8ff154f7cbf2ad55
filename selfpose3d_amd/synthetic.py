"""Deterministic synthetic multi-view scene.

This is INPUT generation for tests / bench / golden vectors: a ring of V calibrated
pin-hole + distortion cameras around the Panoptic capture space, the per-view ``meta``
dictionaries in the batched layout the reference's DataLoader collate produces
(/root/reference/lib/dataset/JointsDataset.py:211-223, JointsDatasetSSV.py:540-587),
and per-view heatmaps (uniform-random or Gaussian "people").

Everything is generated from ``numpy.random.default_rng(seed)`` (PCG64: the stream is
stable across machines and numpy versions), so the GPU box regenerates bit-identical
inputs to the ones the committed golden vectors were computed on.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

# Panoptic-like defaults (configs/panoptic/resnet50/prn64_cpn80x80x20_960x512_cam5.yaml:71-94)
SPACE_SIZE = (8000.0, 8000.0, 2000.0)
SPACE_CENTER = (0.0, -500.0, 800.0)
INITIAL_CUBE_SIZE = (80, 80, 20)
FINE_GRID_SIZE = (2000.0, 2000.0, 2000.0)
FINE_CUBE_SIZE = (64, 64, 64)
ORIG_IMAGE = (1920, 1080)


def get_scale(image_size, resized_size):
    """Aspect-preserving crop scale /200 (reference: lib/utils/transforms.py:151-162)."""
    w, h = image_size
    wr, hr = resized_size
    if w / wr < h / hr:
        w_pad, h_pad = h / hr * wr, h
    else:
        w_pad, h_pad = w, w / wr * hr
    return np.array([w_pad / 200.0, h_pad / 200.0], dtype=np.float32)


def ring_cameras(num_views: int, radius: float = 3000.0, height: float = 2000.0,
                 target=(0.0, -500.0, 800.0)) -> List[dict]:
    """V cameras on a ring looking at ``target``; world z is up, units mm.

    Dict fields follow the reference's camera dict (lib/dataset/panoptic.py:221-233):
    ``R`` world->camera rotation, ``T`` camera centre in world (Xc = R (X - T)),
    ``fx fy cx cy``, ``k`` (3,1) radial, ``p`` (2,1) tangential.
    """
    cams = []
    tgt = np.asarray(target, dtype=np.float64)
    for i in range(num_views):
        ang = 2.0 * math.pi * i / num_views
        c = np.array([tgt[0] + radius * math.cos(ang), tgt[1] + radius * math.sin(ang), height])
        fwd = tgt - c
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], axis=0)
        cams.append({
            "R": R.astype(np.float64),
            "T": c.reshape(3, 1).astype(np.float64),
            "fx": np.float64(1400.0), "fy": np.float64(1400.0),
            "cx": np.float64(960.0), "cy": np.float64(540.0),
            "k": np.array([[-0.2], [0.1], [0.0]], dtype=np.float64),
            "p": np.array([[1e-3], [-1e-3]], dtype=np.float64),
        })
    return cams


def make_meta(batch: int, num_views: int, image_size: Sequence[int],
              rotations: Optional[Sequence[float]] = None,
              scale_mults: Optional[Sequence[float]] = None,
              ssv_style: bool = False) -> List[dict]:
    """Per-view meta dicts with a leading batch dim (what default collate emits).

    ``rotations`` / ``scale_mults`` are per-sample augmentation (degrees, multiplier);
    None = the un-augmented validation path (rotation 0, int64 like the reference).
    ``ssv_style`` emits fp32 camera tensors (JointsDatasetSSV.py:230-237).
    """
    cams = ring_cameras(num_views)
    base_scale = get_scale(ORIG_IMAGE, image_size)
    metas = []
    for v in range(num_views):
        cam = cams[v]
        cdt = torch.float32 if ssv_style else torch.float64
        camera = {
            "R": torch.as_tensor(np.repeat(cam["R"][None], batch, 0), dtype=cdt),
            "T": torch.as_tensor(np.repeat(cam["T"][None], batch, 0), dtype=cdt),
            "fx": torch.full((batch,), float(cam["fx"]), dtype=cdt),
            "fy": torch.full((batch,), float(cam["fy"]), dtype=cdt),
            "cx": torch.full((batch,), float(cam["cx"]), dtype=cdt),
            "cy": torch.full((batch,), float(cam["cy"]), dtype=cdt),
            "k": torch.as_tensor(np.repeat(cam["k"][None], batch, 0), dtype=cdt),
            "p": torch.as_tensor(np.repeat(cam["p"][None], batch, 0), dtype=cdt),
        }
        if ssv_style:                                          # JointsDatasetSSV.py:230-237 also stores the pairs
            camera["f"] = torch.stack([camera["fx"], camera["fy"]], -1).reshape(batch, 2, 1)
            camera["c"] = torch.stack([camera["cx"], camera["cy"]], -1).reshape(batch, 2, 1)
        center = torch.tensor([[ORIG_IMAGE[0] / 2.0, ORIG_IMAGE[1] / 2.0]] * batch, dtype=torch.float64)
        scale = np.repeat(base_scale[None], batch, 0).copy()
        if scale_mults is not None:
            scale = scale * np.asarray(scale_mults, dtype=np.float32)[:, None]
        if rotations is None:
            rotation = torch.zeros(batch, dtype=torch.int64)
        else:
            rotation = torch.tensor(list(rotations), dtype=torch.float64)
        metas.append({
            "center": center,
            "scale": torch.as_tensor(scale, dtype=torch.float32),
            "rotation": rotation,
            "camera": camera,
        })
    return metas


def random_heatmaps(batch, num_views, num_joints, hm_h, hm_w, seed=0, device="cpu") -> List[torch.Tensor]:
    """U[0,1) heatmaps, list[V] of (B,J,h,w) fp32 (worst case for the gather)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(num_views):
        a = rng.random((batch, num_joints, hm_h, hm_w), dtype=np.float32)
        out.append(torch.from_numpy(a).to(device))
    return out


def _project_f64(X, cam):
    """float64 pin-hole + distortion projection of (P,3) world points (input generation only)."""
    R, T = cam["R"], cam["T"].reshape(3)
    xc = (X - T) @ R.T
    y = xc[:, :2] / (xc[:, 2:3] + 1e-5)
    r2 = np.minimum((y ** 2).sum(1), 1e10)
    k, p = cam["k"].reshape(3), cam["p"].reshape(2)
    radial = 1 + k[0] * r2 + k[1] * r2 ** 2 + k[2] * r2 ** 3
    tan = p[0] * y[:, 1] + p[1] * y[:, 0]
    y = y * (radial + 2 * tan)[:, None] + np.outer(r2, np.array([p[1], p[0]]))
    return np.stack([cam["fx"] * y[:, 0] + cam["cx"], cam["fy"] * y[:, 1] + cam["cy"]], 1)


def people_points(batch, num_joints, seed=0, min_people=2, max_people=5):
    """Random 3D 'skeletons': per sample P roots >=900 mm apart inside the capture space,
    each with J joints scattered +-250 mm around the root.  Returns list[B] of (P,J,3)."""
    rng = np.random.default_rng(seed + 7919)
    out = []
    for _ in range(batch):
        P = int(rng.integers(min_people, max_people + 1))
        roots = []
        while len(roots) < P:
            c = np.array([rng.uniform(-1500, 1500), rng.uniform(-2000, 1000), rng.uniform(700, 1100)])
            if all(np.linalg.norm(c[:2] - r[:2]) > 900.0 for r in roots):
                roots.append(c)
        roots = np.stack(roots)
        joints = roots[:, None, :] + rng.uniform(-250, 250, size=(P, num_joints, 3))
        joints[:, min(2, num_joints - 1)] = roots          # root joint (ROOTIDX=2) sits on the root
        out.append(joints)
    return out


def people_heatmaps(batch, num_views, num_joints, hm_h, hm_w, image_size, seed=0, sigma=3.0,
                    device="cpu"):
    """Gaussian-blob heatmaps of random 3D people seen by the ring cameras (rot=0 crop).

    Mirrors how the reference renders synthetic 2D heatmaps
    (lib/models/cuboid_proposal_net_soft.py:209-227).  Returns (heatmaps list[V], points list[B]).
    """
    pts = people_points(batch, num_joints, seed)
    cams = ring_cameras(num_views)
    s = get_scale(ORIG_IMAGE, image_size).astype(np.float64) * 200.0
    a = image_size[0] / s[0] if s[0] >= s[1] else image_size[1] / s[1]
    t = np.array([image_size[0] / 2.0, image_size[1] / 2.0]) - a * np.array(ORIG_IMAGE) / 2.0
    ys, xs = np.mgrid[0:hm_h, 0:hm_w].astype(np.float64)
    hms = []
    for v in range(num_views):
        hm = np.zeros((batch, num_joints, hm_h, hm_w), dtype=np.float64)
        for b in range(batch):
            P = pts[b].shape[0]
            px = _project_f64(pts[b].reshape(-1, 3), cams[v]).reshape(P, num_joints, 2)
            q = (px * a + t) * np.array([hm_w, hm_h]) / np.array(image_size, dtype=np.float64)
            for pidx in range(P):
                for j in range(num_joints):
                    g = np.exp(-((xs - q[pidx, j, 0]) ** 2 + (ys - q[pidx, j, 1]) ** 2) / (2 * sigma ** 2))
                    hm[b, j] = np.maximum(hm[b, j], g)
        hms.append(torch.from_numpy(np.clip(hm, 0, 1).astype(np.float32)).to(device))
    return hms, pts


def fill_parameters_deterministic(module: torch.nn.Module, seed: int = 0, scale: float = 0.05):
    """Fill every parameter/buffer of ``module`` from a numpy stream in sorted-key order.

    Used so an independently-constructed twin of a reference module (same state_dict keys)
    gets bit-identical weights without shipping a checkpoint.  BatchNorm running_var is
    kept positive; num_batches_tracked untouched.
    """
    rng = np.random.default_rng(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for key in sorted(sd.keys()):
            t = sd[key]
            if not torch.is_floating_point(t):
                continue
            a = rng.standard_normal(tuple(t.shape)).astype(np.float32) * scale
            if key.endswith("running_var"):
                a = np.abs(a) + 1.0
            elif key.endswith("weight") and t.dim() == 1:      # BN gamma
                a = a + 1.0
            t.copy_(torch.from_numpy(a))
    return module


def random_meta(batch: int, num_views: int, image_size, seed: int = 0, augment: bool = True, ssv_style: bool = False):
    """Random calibrated rigs for parity sweeps: per (sample, view) a camera at a random position around the
    capture space looking roughly at it, random intrinsics / distortion, random original-image size, and
    (optionally) random crop rotation / scale.  Same dict layout as ``make_meta``."""
    rng = np.random.default_rng(seed)
    cdt = torch.float32 if ssv_style else torch.float64
    metas = []
    for _ in range(num_views):
        R, T, fx, fy, cx, cy, k, p, cen, sc, rot = ([] for _ in range(11))
        for _b in range(batch):
            ang, rad, hgt = rng.uniform(0, 2 * math.pi), rng.uniform(2500, 6000), rng.uniform(800, 3000)
            c = np.array([rad * math.cos(ang), -500 + rad * math.sin(ang), hgt])
            tgt = np.array([rng.uniform(-800, 800), -500 + rng.uniform(-800, 800), rng.uniform(500, 1200)])
            fwd = (tgt - c) / np.linalg.norm(tgt - c)
            right = np.cross(fwd, np.array([0.0, 0.0, 1.0])); right /= np.linalg.norm(right)
            roll = rng.uniform(-0.2, 0.2)
            down = np.cross(fwd, right)
            right, down = math.cos(roll) * right + math.sin(roll) * down, -math.sin(roll) * right + math.cos(roll) * down
            W0, H0 = [(1920, 1080), (1032, 776), (360, 288), (1280, 720)][int(rng.integers(4))]
            f = rng.uniform(0.6, 1.4) * W0
            R.append(np.stack([right, down, fwd])); T.append(c.reshape(3, 1))
            fx.append(f); fy.append(f * rng.uniform(0.97, 1.03))
            cx.append(W0 / 2 + rng.uniform(-20, 20)); cy.append(H0 / 2 + rng.uniform(-20, 20))
            k.append(np.array([[rng.uniform(-0.3, 0.1)], [rng.uniform(-0.1, 0.2)], [rng.uniform(-0.05, 0.05)]]))
            p.append(np.array([[rng.uniform(-2e-3, 2e-3)], [rng.uniform(-2e-3, 2e-3)]]))
            cen.append([W0 / 2.0, H0 / 2.0])
            s = get_scale((W0, H0), image_size)
            if augment:
                s = (s * np.float32(rng.uniform(0.7, 1.35))).astype(np.float32)
            sc.append(s)
            rot.append(float(rng.uniform(-40, 40)) if augment and rng.random() < 0.7 else 0.0)
        camera = {"R": torch.as_tensor(np.stack(R), dtype=cdt), "T": torch.as_tensor(np.stack(T), dtype=cdt),
                  "fx": torch.as_tensor(np.array(fx), dtype=cdt), "fy": torch.as_tensor(np.array(fy), dtype=cdt),
                  "cx": torch.as_tensor(np.array(cx), dtype=cdt), "cy": torch.as_tensor(np.array(cy), dtype=cdt),
                  "k": torch.as_tensor(np.stack(k), dtype=cdt), "p": torch.as_tensor(np.stack(p), dtype=cdt)}
        metas.append({"center": torch.tensor(cen, dtype=torch.float64), "scale": torch.as_tensor(np.stack(sc), dtype=torch.float32),
                      "rotation": torch.tensor(rot, dtype=torch.float64), "camera": camera})
    return metas
