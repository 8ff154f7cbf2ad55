"""Synthetic Panoptic-shape dataset: the sample tuple and ``meta`` schema the reference's datasets
emit (/root/reference/lib/dataset/JointsDataset.py:102-225, panoptic.py:221-233) without any real data
(the CMU Panoptic files and OpenCV are not in the image).  Deterministic per (seed, index).

item = (inputs[V] (3,H,W), target_heatmaps[V] (J,h,w), target_weights[V] (J,1), targets_3d[V] (X,Y,Z),
        meta[V] dict, input_heatmaps[V] (J,h,w))  - default_collate adds the batch dim.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

from . import synthetic as syn


class SyntheticPanoptic(Dataset):
    def __init__(self, cfg, num_frames: int = 32, seed: int = 0, max_people: int = 4, images: bool = True):
        self.cfg = cfg
        self.n = int(num_frames)
        self.seed = int(seed)
        self.V = int(cfg.DATASET.CAMERA_NUM)
        self.J = int(cfg.NETWORK.NUM_JOINTS)
        self.img = [int(v) for v in cfg.NETWORK.IMAGE_SIZE]
        self.hm = [int(v) for v in cfg.NETWORK.HEATMAP_SIZE]
        self.space_size = np.asarray(cfg.MULTI_PERSON.SPACE_SIZE, np.float64)
        self.space_center = np.asarray(cfg.MULTI_PERSON.SPACE_CENTER, np.float64)
        self.cube = [int(v) for v in cfg.MULTI_PERSON.INITIAL_CUBE_SIZE]
        self.maxp = int(cfg.MULTI_PERSON.MAX_PEOPLE_NUM)
        self.max_people = min(max_people, self.maxp)
        self.root_id = int(cfg.DATASET.ROOTIDX) if not isinstance(cfg.DATASET.ROOTIDX, (list, tuple)) else 2
        self.images = images
        self.cams = syn.ring_cameras(self.V)
        self.scale = syn.get_scale(syn.ORIG_IMAGE, self.img)

    def __len__(self):
        return self.n

    def _image(self, idx, v):
        """N(0,1) camera image from a small pool (content is irrelevant to the geometry; avoids 7 M normal
        draws per frame on the host)"""
        if not hasattr(self, "_pool"):
            rng = np.random.default_rng(self.seed + 12345)
            self._pool = [torch.from_numpy(rng.standard_normal((3, self.img[1], self.img[0]), dtype=np.float32))
                          for _ in range(4)]
        return self._pool[(idx * self.V + v) % len(self._pool)]

    def _target_3d(self, roots):
        """max of 3D Gaussians (sigma 200 mm) at the roots (JointsDataset.generate_3d_target:304-341)"""
        X, Y, Z = self.cube
        gx = np.linspace(-self.space_size[0] / 2, self.space_size[0] / 2, X) + self.space_center[0]
        gy = np.linspace(-self.space_size[1] / 2, self.space_size[1] / 2, Y) + self.space_center[1]
        gz = np.linspace(-self.space_size[2] / 2, self.space_size[2] / 2, Z) + self.space_center[2]
        t = np.zeros((X, Y, Z), np.float32)
        for r in roots:                                   # separable: exp on the three 1-D profiles
            ex = np.exp(-((gx - r[0]) ** 2) / (2 * 200.0 ** 2))
            ey = np.exp(-((gy - r[1]) ** 2) / (2 * 200.0 ** 2))
            ez = np.exp(-((gz - r[2]) ** 2) / (2 * 200.0 ** 2))
            t = np.maximum(t, (ex[:, None, None] * ey[None, :, None] * ez[None, None, :]).astype(np.float32))
        return np.clip(t, 0, 1)

    def _render(self, joints):
        """separable Gaussian rendering (exp on 1-D profiles, outer product, max over people): list[V] (J,h,w)"""
        w, h = self.hm
        sigma = float(self.cfg.NETWORK.SIGMA)
        s = self.scale.astype(np.float64) * 200.0
        a = self.img[0] / s[0] if s[0] >= s[1] else self.img[1] / s[1]
        t = np.array([self.img[0] / 2.0, self.img[1] / 2.0]) - a * np.array(syn.ORIG_IMAGE) / 2.0
        xs, ys = np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64)
        P = joints.shape[0]
        out = []
        for cam in self.cams:
            px = syn._project_f64(joints.reshape(-1, 3), cam).reshape(P, self.J, 2)
            qq = (px * a + t) * np.array([w, h]) / np.array(self.img, dtype=np.float64)
            gx = np.exp(-((xs[None, None, :] - qq[..., 0:1]) ** 2) / (2 * sigma ** 2))      # (P,J,w)
            gy = np.exp(-((ys[None, None, :] - qq[..., 1:2]) ** 2) / (2 * sigma ** 2))      # (P,J,h)
            hm = (gy[..., :, None] * gx[..., None, :]).max(axis=0)                            # (J,h,w)
            out.append(torch.from_numpy(np.clip(hm, 0, 1).astype(np.float32)))
        return out

    def __getitem__(self, idx):
        w, h = self.hm
        pts = syn.people_points(1, self.J, self.seed * 100003 + idx)
        joints = pts[0][:self.max_people]                              # (P,J,3)
        hms = [x[None] for x in self._render(joints)]
        P = joints.shape[0]
        j3d = np.zeros((self.maxp, self.J, 3)); j3d[:P] = joints
        vis = np.zeros((self.maxp, self.J, 3)); vis[:P] = 1.0
        roots = j3d[:, self.root_id]
        t3d = torch.from_numpy(self._target_3d(roots[:P]))
        inputs, targets, weights, t3ds, metas, ihm = [], [], [], [], [], []
        for v in range(self.V):
            cam = self.cams[v]
            img = self._image(idx, v) if self.images else torch.zeros(3, 1, 1)
            inputs.append(img)
            targets.append(hms[v][0])
            weights.append(torch.ones(self.J, 1))
            t3ds.append(t3d)
            ihm.append(hms[v][0])
            metas.append({
                "image": f"synthetic/{idx:06d}_{v}", "num_person": P, "joints_3d": j3d, "joints_3d_vis": vis,
                "roots_3d": roots, "center": np.array([syn.ORIG_IMAGE[0] / 2.0, syn.ORIG_IMAGE[1] / 2.0]),
                "scale": self.scale.copy(), "rotation": 0,
                "camera": {k: (np.asarray(val)) for k, val in cam.items()},
            })
        return inputs, targets, weights, t3ds, metas, ihm


class SyntheticPanopticSSV(SyntheticPanoptic):
    """The 18-tuple the reference's self-supervised loop consumes (lib/core/function.py:50-69; sample schema of
    lib/dataset/JointsDatasetSSV.py:540-587): the same frame as THREE view sets - two under independent random crop
    rotation / scale (``DATASET.ROT_FACTOR1/2`` degrees, ``SCALE_FACTOR1/2``), one plain - each as
    (inputs, target_heatmaps, target_weights, targets_3d, meta, input_heatmaps).  The target heat-maps play the role of
    the pseudo labels (here: Gaussians of the projected ground-truth joints under that set's crop); ``meta`` carries the
    SSV extras: fp32 camera tensors incl. ``f`` / ``c``, the crop affine ``trans``, ``hflip`` (always False here),
    the 2D pseudo-label joints ``joints`` / ``joints_vis`` in network-input pixels.  RandAugment / cut-out act on image
    CONTENT only and are not modelled (the images are noise)."""

    def _set(self, idx, joints, rot, mult):
        from .camera_pack import get_affine_transform_batch
        w, h = self.hm
        sigma = float(self.cfg.NETWORK.SIGMA)
        P = joints.shape[0]
        center = np.array([syn.ORIG_IMAGE[0] / 2.0, syn.ORIG_IMAGE[1] / 2.0])
        scale = (self.scale * np.float32(mult)).astype(np.float32)
        trans = get_affine_transform_batch(center[None], scale[None], np.array([rot], np.float64), self.img)[0]
        j3d = np.zeros((self.maxp, self.J, 3)); j3d[:P] = joints
        vis3 = np.zeros((self.maxp, self.J, 3)); vis3[:P] = 1.0
        roots = j3d[:, self.root_id]
        t3d = torch.from_numpy(self._target_3d(roots[:P]))
        xs, ys = np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64)
        stride = np.array([self.img[0] / w, self.img[1] / h])
        inputs, targets, weights, t3ds, metas, ihm = [], [], [], [], [], []
        for v, cam in enumerate(self.cams):
            px = syn._project_f64(joints.reshape(-1, 3), cam).reshape(P, self.J, 2)
            q = px @ trans[:, :2].T + trans[:, 2]                              # network-input pixels
            qq = q / stride
            gx = np.exp(-((xs[None, None, :] - qq[..., 0:1]) ** 2) / (2 * sigma ** 2))
            gy = np.exp(-((ys[None, None, :] - qq[..., 1:2]) ** 2) / (2 * sigma ** 2))
            hm = torch.from_numpy(np.clip((gy[..., :, None] * gx[..., None, :]).max(axis=0), 0, 1).astype(np.float32))
            j2d = np.zeros((self.maxp, self.J, 2), np.float32); j2d[:P] = q
            v2d = np.zeros((self.maxp, self.J, 2), np.float32); v2d[:P] = 1.0
            camera = {k: np.asarray(val, np.float32) for k, val in cam.items()}
            camera["f"] = np.array([[cam["fx"]], [cam["fy"]]], np.float32)
            camera["c"] = np.array([[cam["cx"]], [cam["cy"]]], np.float32)
            inputs.append(self._image(idx, v) if self.images else torch.zeros(3, 1, 1))
            targets.append(hm); weights.append(torch.ones(self.J, 1)); t3ds.append(t3d); ihm.append(hm)
            metas.append({"image": f"synthetic/{idx:06d}_{v}", "num_person": P, "joints_3d": j3d, "joints_3d_vis": vis3,
                          "roots_3d": roots, "center": center.copy(), "scale": scale.copy(), "rotation": float(rot),
                          "camera": camera, "trans": trans.astype(np.float32), "hflip": False,
                          "joints": j2d, "joints_vis": v2d, "mis_count": 0})
        return inputs, targets, weights, t3ds, metas, ihm

    def __getitem__(self, idx):
        ds = self.cfg.DATASET
        rng = np.random.default_rng(self.seed * 7919 + idx)
        pts = syn.people_points(1, self.J, self.seed * 100003 + idx)
        joints = pts[0][:self.max_people]
        out = ()
        for k in (1, 2):
            rf, sf = float(ds.get(f"ROT_FACTOR{k}", 45)), float(ds.get(f"SCALE_FACTOR{k}", 0.35))
            rot = float(np.clip(rng.standard_normal() * rf, -2 * rf, 2 * rf)) if rng.random() < 0.6 else 0.0
            mult = float(np.clip(rng.standard_normal() * sf + 1.0, 1 - sf, 1 + sf))
            out = out + self._set(idx, joints, rot, mult)
        return out + self._set(idx, joints, 0.0, 1.0)
