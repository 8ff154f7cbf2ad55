#!/usr/bin/env python3
"""Low-resolution Winograd layer (HIP transforms + library GEMM): whole batch at once vs per-sample chunks (does keeping
the transformed tensors V / M inside the Infinity Cache pay?).  Measurement only.
Result (round 2): (4,64,40,40,10) 132 -> 119 us in isolation with two halves, but no gain inside the step (2.44 vs 2.43 ms):
not adopted."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from selfpose3d_amd import _lib
dev = torch.device("cuda:0")


def timed(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 1)


tun = torch.cuda.tunable
tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30)
out = {}
for (B, C, X, Y, Z) in ((4, 64, 40, 40, 10), (4, 128, 20, 20, 5)):
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    U = _lib.wino_weights(w); s = torch.randn(C, device=dev)
    x = torch.randn(B, C, X, Y, Z, device=dev).contiguous(memory_format=torch.channels_last_3d)
    res = torch.randn_like(x)
    xs = [x[b:b + 1].contiguous(memory_format=torch.channels_last_3d) for b in range(B)]
    rs = [res[b:b + 1].contiguous(memory_format=torch.channels_last_3d) for b in range(B)]
    x2 = [x[b:b + 2].contiguous(memory_format=torch.channels_last_3d) for b in range(0, B, 2)]
    r2 = [res[b:b + 2].contiguous(memory_format=torch.channels_last_3d) for b in range(0, B, 2)]
    whole = lambda: _lib.wino_conv3d_(x, U, s, 2, res)
    per1 = lambda: [_lib.wino_conv3d_(a, U, s, 2, r) for a, r in zip(xs, rs)]
    per2 = lambda: [_lib.wino_conv3d_(a, U, s, 2, r) for a, r in zip(x2, r2)]
    whole(); per1(); per2()          # tune the GEMM shapes
    out[f"{B}x{C}x{X}x{Y}x{Z}"] = {"whole_batch_us": timed(whole), "two_chunks_us": timed(per2), "per_sample_us": timed(per1)}
print(json.dumps(out, indent=1))
