#!/usr/bin/env python3
"""Pose stage of the inference path: PoseRegressionNet.forward_batched on P = B*K person proposals of the
bench rig (5 views, 64^3 fine cubes around each proposal): one indexed unprojection launch, V2V in chunks of 8 cubes,
soft-argmax with in-kernel grids.  Prints ms per batch and per cube, plus the plan switches' A/B."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from selfpose3d_amd import _lib, synthetic as syn  # noqa: E402
from selfpose3d_amd.config import load_config  # noqa: E402
from selfpose3d_amd.pose_regression_net import PoseRegressionNet  # noqa: E402
from selfpose3d_amd.project_layer import nhwc_heatmap_views  # noqa: E402

dev = torch.device("cuda:0")
B, K = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = load_config(None)
V, J = int(cfg.DATASET.CAMERA_NUM), int(cfg.NETWORK.NUM_JOINTS)
w, h = cfg.NETWORK.HEATMAP_SIZE
meta = syn.make_meta(B, V, cfg.NETWORK.IMAGE_SIZE)
hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=5)]
hms = nhwc_heatmap_views(_lib.pack_heatmaps(hms, jp=16), J)
net = PoseRegressionNet(cfg)
syn.fill_parameters_deterministic(net, seed=73, scale=0.05)
net.eval().to(dev).use_channels_last(True)
g = torch.Generator().manual_seed(3)
gc = torch.zeros(B, K, 5)
gc[..., 0] = (torch.rand(B, K, generator=g) - 0.5) * 4000
gc[..., 1] = (torch.rand(B, K, generator=g) - 0.5) * 4000
gc[..., 2] = 800 + torch.rand(B, K, generator=g) * 400
gc = gc.to(dev)


def run():
    with torch.no_grad():
        return net.forward_batched(hms, meta, gc)


def timeit(n=10, w=3):
    for _ in range(w):
        run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


out = {"cubes": B * K}
ref = run().clone()
out["ms_per_batch"] = round(timeit(), 3)
out["ms_per_cube"] = round(out["ms_per_batch"] / (B * K), 4)
for flag in (() if os.environ.get("SP3D_POSE_ONLY") else ("direct_conv", "wino_split", "winograd", "fft_front")):
    setattr(net.v2v_net, flag, False)
    net.v2v_net.invalidate_plan()
    alt = run()
    out[f"ms_without_{flag}"] = round(timeit(5, 2), 3)
    out[f"max_abs_dev_without_{flag}_mm"] = float((alt - ref).abs().max())
    setattr(net.v2v_net, flag, True)
    net.v2v_net.invalidate_plan()
if os.environ.get("SP3D_POSE_ONLY"):
    timeit(3, 0)            # the tail of a kernel trace is then three steady-state batches
print(json.dumps(out))
