#!/usr/bin/env python3
"""How fast is the batched fp32 GEMM a Winograd F(2,3) formulation of the V2V 3x3x3 convs would need (measurement only)."""
import json, os, sys
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True


def timed(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

out = {}
for name, tiles, C, O, grid in (("quarter_128", 4 * 10 * 10 * 3, 128, 128, (20, 20, 5)), ("half_64", 4 * 20 * 20 * 5, 64, 64, (40, 40, 10)),
                                ("full_32", 4 * 40 * 40 * 10, 32, 32, (80, 80, 20))):
    a = torch.randn(64, tiles, C, device=dev); w = torch.randn(64, C, O, device=dev)
    t = timed(lambda: torch.bmm(a, w))
    x = torch.randn(4, C, *grid, device=dev).contiguous(memory_format=torch.channels_last_3d)
    wt = torch.randn(O, C, 3, 3, 3, device=dev).contiguous(memory_format=torch.channels_last_3d)
    td = timed(lambda: F.conv3d(x, wt, padding=1))
    fl = 64 * tiles * C * O * 2
    out[name] = {"bmm_us": round(t, 1), "bmm_TFLOPs": round(fl / t / 1e6, 1), "direct_conv_us": round(td, 1),
                 "direct_TFLOPs": round(4 * grid[0] * grid[1] * grid[2] * 27 * C * O * 2 / td / 1e6, 1),
                 "transformed_act_MB": round(64 * tiles * C * 4 / 1e6, 1)}
print(json.dumps(out, indent=1))
