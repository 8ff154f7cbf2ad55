#!/usr/bin/env python3
"""run the fused Winograd kernel a few times (target for rocprofv3 --pmc passes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from selfpose3d_amd import _lib
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
U = _lib.wino_weights(w); s = torch.randn(32, device=dev)
x = torch.randn(B, 32, 80, 80, 20, device=dev).contiguous(memory_format=torch.channels_last_3d)
for _ in range(12):
    _lib.wino_fused_conv3d_(x, U, s, 1)
torch.cuda.synchronize()
