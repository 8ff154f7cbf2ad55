#!/usr/bin/env python3
"""Time sp3d_unproject_bwd (HIP events) on the coarse and fine workloads."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from selfpose3d_amd import _lib, synthetic as syn
from selfpose3d_amd.camera_pack import pack_cameras
if os.environ.get("SP3D_LIB"):                      # a measurement build instead of the shipped library
    _lib.LIB_PATH = os.path.abspath(os.environ["SP3D_LIB"])
dev = torch.device("cuda:0")
def timed(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
img, (w, h), J = (960, 512), (240, 128), 15
res = {}
for name, B, V, cube, gs, fine in (("coarse_b2", 2, 5, syn.INITIAL_CUBE_SIZE, syn.SPACE_SIZE, False), ("coarse_b4", 4, 5, syn.INITIAL_CUBE_SIZE, syn.SPACE_SIZE, False),
                                   ("fine_p4_b2", 2, 5, syn.FINE_CUBE_SIZE, syn.FINE_GRID_SIZE, True)):
    if os.environ.get("BWD_ONLY") and name not in os.environ["BWD_ONLY"].split(","):
        continue
    meta = syn.make_meta(B, V, img)
    cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
    hms = [x.to(dev) for x in syn.people_heatmaps(B, V, J, h, w, img, seed=3)[0]]
    if fine:
        P = 4
        rng = np.random.default_rng(0)
        c = np.stack([rng.uniform(-1500, 1500, P), rng.uniform(-2000, 1000, P), rng.uniform(700, 1100, P)], 1).astype(np.float32)
        centers = torch.from_numpy(c).to(dev); sample_of = torch.tensor([0, 0, 1, 1], dtype=torch.int32, device=dev)
    else:
        P = B
        centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev); sample_of = None
    valid = torch.ones(P, dtype=torch.uint8, device=dev)
    g = torch.randn(P, J, *cube, device=dev)
    res[name] = {"bwd_us": round(timed(lambda: _lib.unproject_bwd(hms, cam, centers, valid, g, cube, gs, img, sample_of=sample_of)), 1)}
    packed = _lib.pack_heatmaps(hms, jp=16)
    mask = torch.empty((P, cube[0] * cube[1] * cube[2]), dtype=torch.int16, device=dev)
    _lib.unproject_fwd([packed[c] for c in range(V)], _lib.LAYOUT_NHWC, 16, cam, centers, valid, P, J, h, w, cube, gs, img,
                       False, sample_of=sample_of, pass_mask=mask)
    res[name]["bwd_packed_us"] = round(timed(lambda: _lib.unproject_bwd_packed(cam, centers, valid, g, mask, B, V, J, 16, h, w,
                                                                                cube, gs, img, sample_of=sample_of)), 1)
    res[name]["bwd_packed_deterministic_us"] = round(timed(lambda: _lib.unproject_bwd_packed(
        cam, centers, valid, g, mask, B, V, J, 16, h, w, cube, gs, img, sample_of=sample_of, deterministic=True)), 1)
    for nm, sc in (("per_tap", _lib.SCATTER_PER_TAP), ("merge", _lib.SCATTER_MERGE)):      # round 6: both kernels on every grid
        res[name][f"bwd_packed_{nm}_us"] = round(timed(lambda: _lib.unproject_bwd_packed(
            cam, centers, valid, g, mask, B, V, J, 16, h, w, cube, gs, img, sample_of=sample_of, scatter=sc)), 1)
    res[name]["fwd_train_us"] = round(timed(lambda: _lib.unproject_fwd([packed[c] for c in range(V)], _lib.LAYOUT_NHWC, 16, cam,
                                                                        centers, valid, P, J, h, w, cube, gs, img, False,
                                                                        sample_of=sample_of, pass_mask=mask)), 1)
print(json.dumps(res))
