#!/usr/bin/env python3
"""Half-resolution fused Winograd layers alone (C = 32 | 64 -> 64 on 40x40x10, B = 4; the 32^3 pose-net size): HIP events,
median of 5 x 40 launches per epilogue mode, and the maximum error against a float64 convolution.  A few seconds of GPU time:
the A/B loop for changes to wino_fused16_kernel.       python tools/bench_wino16.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from selfpose3d_amd import _lib  # noqa: E402

if os.environ.get("SP3D_BENCH_LIB"):               # a measurement build (selfpose3d_amd.build.build_variant) instead of the product
    _lib.LIB_PATH = os.path.abspath(os.environ["SP3D_BENCH_LIB"])

dev = torch.device("cuda:0")
out = {"library": os.path.basename(_lib.LIB_PATH)}
for name, (B, C, S) in {"root_c64_40x40x10_b4": (4, 64, (40, 40, 10)), "root_c32_40x40x10_b4": (4, 32, (40, 40, 10)),
                        "pose_c64_32x32x32_b8": (8, 64, (32, 32, 32))}.items():
    g = torch.Generator().manual_seed(1)
    x = (torch.randn((B, C) + S, generator=g) * 2).to(dev).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn((64, C, 3, 3, 3), generator=g) * 0.05).to(dev)
    shift = torch.randn(64, generator=g).to(dev)
    res = torch.randn((B, 64) + S, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    U = _lib.wino_weights(w)
    U3 = _lib.wino_weights_split(U, 16)
    fn = {"relu": lambda: _lib.wino_fused_conv3d_(x, U, shift, 1, None, U3),
          "residual_relu": lambda: _lib.wino_fused_conv3d_(x, U, shift, 2, res, U3)}
    conv = F.conv3d(x.double(), w.double(), padding=1) + shift.double().view(1, 64, 1, 1, 1)
    ref = {"relu": conv.clamp_min(0), "residual_relu": (conv + res.double()).clamp_min(0)}
    rec = {k: {"max_err_vs_f64": float((f().double() - ref[k]).abs().max()),
               "checksum": float(f().double().sum())} for k, f in fn.items()}
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
