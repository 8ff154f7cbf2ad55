#!/usr/bin/env python3
"""Grouped BatchNorm kernels (sp3d_gbn_*) against their HBM roofline on the train step's shapes, next to the library's
BatchNorm (+ ReLU) on the same tensors.  Algorithmic bytes: forward 3 x 4 B per element (read x twice, write y), backward 5 x
4 B (read x and dy twice, write dx); mode 2 adds one read (residual / y) per pass and one write (grad_residual).
HIP events, inputs rotated through 3 buffers (the tensors exceed the 256 MB Infinity Cache only for the largest layer)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn as nn
from selfpose3d_amd.grouped_bn import GroupedBatchNorm2d, GroupedBatchNorm3d, GroupSpec

dev = torch.device("cuda:0")
PEAK = 8000.0
SHAPES = {"backbone_layer1_bn3 (10 x 256 x 128x240, groups n % 5)": (2, (10, 256, 128, 240), [n % 5 for n in range(10)]),
          "backbone_layer2_bn1 (10 x 128 x 64x120)": (2, (10, 128, 64, 120), [n % 5 for n in range(10)]),
          "pose_v2v_full_res (5 x 32 x 64^3, slots 2+2+1)": (3, (5, 32, 64, 64, 64), [0, 0, 1, 1, 2]),
          "pose_v2v_half_res (5 x 64 x 32^3)": (3, (5, 64, 32, 32, 32), [0, 0, 1, 1, 2])}


def t_us(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


out = {}
for name, (dims, shape, group_of) in SHAPES.items():
    fmt = torch.channels_last if dims == 2 else torch.channels_last_3d
    G = max(group_of) + 1
    spec = GroupSpec([group_of.count(g) for g in range(G)], dev, group_of=group_of)
    C = shape[1]
    bn = (GroupedBatchNorm2d if dims == 2 else GroupedBatchNorm3d)(C).to(dev).train()
    ref = (nn.BatchNorm2d if dims == 2 else nn.BatchNorm3d)(C).to(dev).train().to(memory_format=fmt)
    xs = [torch.randn(shape, device=dev).contiguous(memory_format=fmt).requires_grad_(True) for _ in range(3)]
    gy = torch.randn(shape, device=dev).contiguous(memory_format=fmt)
    nbytes = xs[0].numel() * 4
    rec = {"elements": xs[0].numel(), "tensor_MB": round(nbytes / 1e6, 1)}
    state = {"i": 0}

    def nxt():
        state["i"] += 1
        return xs[state["i"] % 3]
    for mode, relu in (("bn", False), ("bn_relu", True)):
        bn.groups = spec
        with torch.no_grad():
            tf = t_us(lambda: bn.grouped_forward(nxt(), relu=relu))
            tl = t_us(lambda: (torch.relu_(ref(nxt())) if relu else ref(nxt())))

        def fb_grouped():
            x = nxt(); x.grad = None
            bn.grouped_forward(x, relu=relu).backward(gy)

        def fb_lib():
            x = nxt(); x.grad = None
            y = ref(x)
            (torch.relu(y) if relu else y).backward(gy)
        tfb, tlb = t_us(fb_grouped, 15), t_us(fb_lib, 15)
        rec[mode] = {"grouped_fwd_us": round(tf, 1), "grouped_fwd_frac_hbm": round(3 * nbytes / (tf * 1e-6) / 1e9 / PEAK, 3),
                     "library_fwd_us": round(tl, 1),
                     "grouped_fwd_bwd_us": round(tfb, 1), "grouped_bwd_frac_hbm": round(5 * nbytes / ((tfb - tf) * 1e-6) / 1e9 / PEAK, 3),
                     "library_fwd_bwd_us": round(tlb, 1)}
    out[name] = rec
    del xs, gy
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
