#!/usr/bin/env python3
"""run a fused Winograd kernel a few times (target for rocprofv3 --pmc passes)
    run_wino_fused.py B [kind]     kind: fp32 (32->32 full res, v_mfma_f32_32x32x2_f32), split (same layer, bf16 x 3),
                                         half (64->64 at 40x40x10, 16x16x32 bf16 x 3),
                                         direct (32->32 full res, implicit GEMM, bf16 x 3)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from selfpose3d_amd import _lib
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kind = sys.argv[2] if len(sys.argv) > 2 else "fp32"
if kind == "half":
    w = torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05
    U = _lib.wino_weights(w); s = torch.randn(64, device=dev); U3 = _lib.wino_weights_split(U, 16)
    x = torch.randn(B, 64, 40, 40, 10, device=dev).contiguous(memory_format=torch.channels_last_3d)
else:
    w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
    U = _lib.wino_weights(w); s = torch.randn(32, device=dev); U3 = _lib.wino_weights_split(U) if kind == "split" else None
    x = torch.randn(B, 32, 80, 80, 20, device=dev).contiguous(memory_format=torch.channels_last_3d)
W3 = _lib.conv_weights_split(w) if kind == "direct" else None
for _ in range(12):
    if kind == "direct":
        _lib.conv3_split_(x, W3, s, 1)
    else:
        _lib.wino_fused_conv3d_(x, U, s, 1, None, U3)
torch.cuda.synchronize()
