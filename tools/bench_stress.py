#!/usr/bin/env python3
"""root-net forward at BASELINE configs[3] (10 views, 160x160x40, B=1) and at 48x48x12: eager, HIP events.
    python tools/bench_stress.py > gpurun_out/stress_rootnet.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from selfpose3d_amd import synthetic as syn
from selfpose3d_amd.config import load_config
from selfpose3d_amd.cuboid_proposal_net import CuboidProposalNet
from selfpose3d_amd.v2v_net import V2VNet

dev = torch.device("cuda:0")
V2VNet.tune_gemms(True)
out = {}
for name, (B, V, cube) in {"configs3_b1_v10_160x160x40": (1, 10, (160, 160, 40)), "b4_v5_48x48x12": (4, 5, (48, 48, 12)),
                           "b4_v5_80x80x20": (4, 5, (80, 80, 20))}.items():
    img, hm, J = (960, 512), (240, 128), 15
    cfg = load_config(None, MULTI_PERSON__INITIAL_CUBE_SIZE=list(cube))
    net = CuboidProposalNet(cfg)
    syn.fill_parameters_deterministic(net, seed=71, scale=0.05)
    net.eval().to(dev)
    net.use_channels_last(True)
    meta = syn.make_meta(B, V, img)
    hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, hm[1], hm[0], seed=7)]
    with torch.no_grad():
        for _ in range(3):
            net(hms, meta)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            net(hms, meta)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    out[name] = {"ms_per_forward_eager": round(ms, 3), "samples_per_s": round(B / ms * 1e3, 1)}
print(json.dumps(out, indent=1))
