#!/usr/bin/env python3
"""Winograd F(2,3) 3D conv (HIP transforms + rocBLAS bmm) vs MIOpen on the V2V low-resolution layers (measurement only)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from selfpose3d_amd import _lib
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
for name, B, C, O, grid in (("quarter_128_128", 4, 128, 128, (20, 20, 5)), ("quarter_64_128", 4, 64, 128, (20, 20, 5)),
                            ("half_64_64", 4, 64, 64, (40, 40, 10)), ("pose_quarter_128", 8, 128, 128, (16, 16, 16)),
                            ("pose_half_64", 8, 64, 64, (32, 32, 32))):
    x = torch.randn(B, C, *grid, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(O, C, 3, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last_3d)
    s = torch.randn(O, device=dev)
    U = _lib.wino_weights(w)
    direct = lambda: _lib.channel_shift_act_(F.conv3d(x, w, padding=1), s, 1)
    wino = lambda: _lib.wino_conv3d_(x, U, s, 1)
    err = float((direct() - wino()).abs().max())
    out[name] = {"miopen_plus_epilogue_us": round(timed(direct), 1), "winograd_us": round(timed(wino), 1), "max_abs_diff": err}
for name, B, C, O, grid in (("full_32_32", 4, 32, 32, (80, 80, 20)), ("full_16_32", 4, 16, 32, (80, 80, 20)), ("pose_full_32", 8, 32, 32, (64, 64, 64))):
    x = torch.randn(B, C, *grid, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(O, C, 3, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last_3d)
    s = torch.randn(O, device=dev)
    U = _lib.wino_weights(w)
    direct = lambda: _lib.channel_shift_act_(F.conv3d(x, w, padding=1), s, 1)
    fused = lambda: _lib.wino_fused_conv3d_(x, U, s, 1)
    err = float((direct() - fused()).abs().max())
    out[name] = {"miopen_plus_epilogue_us": round(timed(direct), 1), "winograd_fused_us": round(timed(fused), 1), "max_abs_diff": err}
print(json.dumps(out, indent=1))
