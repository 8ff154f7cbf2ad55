#!/usr/bin/env python3
"""Half-resolution 3x3x3 layers of the root net (C = 32 | 64 -> 64 on 40x40x10, B = 4; plus the 32^3 pose-net size): the
wave-specialised fused Winograd kernel (round 5, default) against the round-2 kernel (mode bit 8), HIP events, interleaved."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F
from selfpose3d_amd import _lib

dev = torch.device("cuda:0")
out = {}
for name, (B, C, S) in {"root_c64_40x40x10_b4": (4, 64, (40, 40, 10)), "root_c32_40x40x10_b4": (4, 32, (40, 40, 10)),
                        "pose_c64_32x32x32_b8": (8, 64, (32, 32, 32))}.items():
    g = torch.Generator().manual_seed(1)
    x = (torch.randn((B, C) + S, generator=g) * 2).to(dev).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn((64, C, 3, 3, 3), generator=g) * 0.05).to(dev)
    shift = torch.randn(64, generator=g).to(dev)
    res = torch.randn((B, 64) + S, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    U = _lib.wino_weights(w)
    U3 = _lib.wino_weights_split(U, 16)
    fn = {"specialised": lambda: _lib.wino_fused_conv3d_(x, U, shift, 2, res, U3),
          "round2": lambda: _lib.wino_fused_conv3d_(x, U, shift, 2, res, U3, legacy16=True)}
    ref = (F.conv3d(x.double(), w.double(), padding=1) + shift.double().view(1, 64, 1, 1, 1) + res.double()).clamp_min(0)
    rec = {k: {"max_err_vs_f64": float((f().double() - ref).abs().max())} for k, f in fn.items()}
    rec["max_abs_diff_between_them"] = float((fn["specialised"]() - fn["round2"]()).abs().max())
    t = {k: [] for k in fn}
    for rep in range(5):
        for k, f in fn.items():
            for _ in range(5):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                f()
            e1.record()
            torch.cuda.synchronize()
            t[k].append(e0.elapsed_time(e1) / 40 * 1e3)
    for k in fn:
        rec[k]["us"] = round(float(np.median(t[k])), 2)
    out[name] = rec
print(json.dumps(out))
