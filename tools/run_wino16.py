"""Launch the half-resolution fused Winograd layer (C = 64 -> 64 on 40x40x10, B = 4) a few times per epilogue mode: target for
rocprofv3 --pmc / --kernel-trace (tools/pmc_wino16.sh).    python tools/run_wino16.py [--iters 12]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from selfpose3d_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=12)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B, C, S = 4, 64, (40, 40, 10)
x = (torch.randn((B, C) + S, generator=g) * 2).to(dev).contiguous(memory_format=torch.channels_last_3d)
w = (torch.randn((64, C, 3, 3, 3), generator=g) * 0.05).to(dev)
shift = torch.randn(64, generator=g).to(dev)
res = torch.randn((B, 64) + S, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
U = _lib.wino_weights(w)
U3 = _lib.wino_weights_split(U, 16)
for _ in range(a.iters):
    _lib.wino_fused_conv3d_(x, U, shift, 1, None, U3)
    _lib.wino_fused_conv3d_(x, U, shift, 2, res, U3)
torch.cuda.synchronize()
print("done", a.iters)
