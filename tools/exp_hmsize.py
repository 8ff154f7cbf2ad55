#!/usr/bin/env python3
"""Sensitivity of the unprojection kernel to the heat-map footprint: same voxels, views and tap count, heat-map
resolution scaled down so the per-sample maps fit the XCD's L2 (measurement only)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from selfpose3d_amd import _lib, synthetic as syn
from selfpose3d_amd.camera_pack import pack_cameras
if os.environ.get("SP3D_EXP_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "selfpose3d_amd", os.environ["SP3D_EXP_LIB"])
dev = torch.device("cuda:0")
img, J, B, V = (960, 512), 15, 4, 5
cube, gs = syn.INITIAL_CUBE_SIZE, syn.SPACE_SIZE
meta = syn.make_meta(B, V, img)
cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
valid = torch.ones(B, dtype=torch.uint8, device=dev)
out = {}
for (w, h) in [(240, 128), (120, 64), (60, 32), (30, 16)]:
    hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=7)]
    packed = _lib.pack_heatmaps(hms, jp=16)
    views = [packed[c] for c in range(V)]
    CL = bool(int(os.environ.get("SP3D_EXP_CL", "0")))
    run = lambda: _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16 if CL else J, h, w, cube, gs, img, False, channels_last=CL)
    for _ in range(20): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    out[f"{w}x{h}"] = {"us": round(e0.elapsed_time(e1) * 5, 2), "maps_MB_per_sample": round(V * h * w * 64 / 1e6, 2)}
print(json.dumps(out, indent=1))
