"""SURVEY §8 f3 pinned to the REFERENCE (round-5 review item 6): the heat-maps the self-supervised loss renders from the
predicted 3D poses, and their gradient w.r.t. those poses, caught inside the reference's own train forward
(lib/models/multi_person_posenet_ssv.py:433-465 + lib/utils/cameras.py:58-118) at full heat-map size 240x128 -
tests/golden/render_ssv_full.npz, generator tests/golden/make_goldens_r6.py::g_render_ssv_full.  This repo's path:
reprojection.project_joints (torch) -> sp3d_render_joints_fwd / _bwd (HIP)."""
import json
import os

import numpy as np
import pytest
import torch

from tests import golden_io as gio

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rendered_maps_and_pose_gradients_vs_reference_forward():
    from selfpose3d_amd.camera_pack import pack_cameras
    from selfpose3d_amd.reprojection import reprojection_heatmaps
    dev = torch.device("cuda:0")
    g = gio.load("render_ssv_full")
    cfg = gio.render_cfg()
    batch = gio.train_batch(cfg, B=2, seed=int(g["data_seed"]), ssv=True)
    m1, m2 = batch[4], batch[10]
    assert np.array_equal(m1[0]["trans"].numpy(), g["trans1"]) and np.array_equal(m2[0]["trans"].numpy(), g["trans2"])
    w, h = cfg.NETWORK.HEATMAP_SIZE
    B = 2
    cam = torch.from_numpy(pack_cameras(m1, B, list(cfg.NETWORK.IMAGE_SIZE))).to(dev)      # proj_cameras come from set 1 (:397)
    count = torch.from_numpy(g["count"]).to(dev)
    j1 = torch.from_numpy(g["joints1"]).to(dev).requires_grad_(True)
    j2 = torch.from_numpy(g["joints2"]).to(dev).requires_grad_(True)
    maps21 = reprojection_heatmaps(j2, count, cam, h, w, 4.0, 3.0, torch.from_numpy(g["trans1"]).to(dev))
    maps12 = reprojection_heatmaps(j1, count, cam, h, w, 4.0, 3.0, torch.from_numpy(g["trans2"]).to(dev))
    V, J = maps21.shape[0], maps21.shape[2]
    assert tuple(maps21.shape) == (V, B, J, h, w) == (3, 2, 15, 128, 240)
    rng = np.random.default_rng(int(g["weight_seed"]))
    w21 = torch.from_numpy(rng.standard_normal(tuple(maps21.shape)).astype(np.float32)).to(dev)
    w12 = torch.from_numpy(rng.standard_normal(tuple(maps12.shape)).astype(np.float32)).to(dev)
    ((maps21 * w21).sum() + (maps12 * w12).sum()).backward()
    pw = np.random.default_rng(int(g["pos_seed"])).standard_normal((h, w))
    rec = {}
    for nm, mp_ in (("maps21", maps21), ("maps12", maps12)):
        a = mp_.detach().cpu().numpy()
        a64 = a.astype(np.float64)
        rec[nm] = dict(full=float(np.abs(a[:, 0][:, [0, 2, J - 1]] - g[nm + "_full_b0"]).max()),
                       sub=float(np.abs(a.reshape(-1)[::int(g["sub_stride"])] - g[nm + "_sub"]).max()),
                       sum=float(np.abs(a64.sum(axis=(3, 4)) - g[nm + "_sum"]).max()),
                       pos=float(np.abs((a64 * pw).sum(axis=(3, 4)) - g[nm + "_pos_sum"]).max()),
                       at_one=int(np.abs((a == 1).sum(axis=(3, 4)) - g[nm + "_at_one"]).sum()))
        assert rec[nm]["full"] <= 2e-5 and rec[nm]["sub"] <= 2e-5, rec          # every pixel of 9 maps + 1 in 53 of all
        assert rec[nm]["sum"] <= 2e-5 * h * w and rec[nm]["pos"] <= 2e-5 * h * w, rec
        assert np.abs(a.max(axis=(3, 4)) - g[nm + "_max"]).max() <= 2e-5
        assert float(a.max()) == 1.0 and rec[nm]["at_one"] <= 2              # the clip at 1 is active in the reference's maps too
    for nm, t, ref in (("grad_joints1", j1.grad, g["grad_joints1"]), ("grad_joints2", j2.grad, g["grad_joints2"])):
        got = t.cpu().numpy()
        scale = float(np.abs(ref).max())
        assert scale > 1e-3
        rec[nm] = dict(err=float(np.abs(got - ref).max()), scale=scale)
        # fp32 on both sides: a gradient is a sum over ~5e3 pixels of (weight x d exp / d x), the reference's through
        # autograd over (P,J,h,w) temporaries, ours analytic per joint; observed ~1e-5 of the largest component
        assert rec[nm]["err"] <= 2e-4 * scale, rec
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "render_reference_pin.json"), "w") as f:
        json.dump(rec, f, indent=1)
