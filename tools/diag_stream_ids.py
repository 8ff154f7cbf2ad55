#!/usr/bin/env python3
"""run under `rocprofv3 --kernel-trace`: library kernels and libsp3d kernels on torch's DEFAULT stream (handle 0) and on a
side stream - do they land on the same HIP stream / HSA queue?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from selfpose3d_amd import _lib
dev = torch.device("cuda:0")
x = torch.rand(2, 16, 8, 8, 4, device=dev).contiguous(memory_format=torch.channels_last_3d)
shift = torch.rand(16, device=dev)
print("default stream handle", torch.cuda.current_stream(dev).cuda_stream)
for _ in range(3):
    y = torch.relu(x) * 2.0                      # library kernels
    y = _lib.channel_shift_act_(y, shift, 1)     # libsp3d kernel, handle of torch's current stream
    y = _lib.maxpool2x(y)
torch.cuda.synchronize()
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    print("side stream handle", torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(3):
        y = torch.sigmoid(x) * 3.0
        y = _lib.channel_shift_act_(y, shift, 1)
        y = _lib.maxpool2x(y)
torch.cuda.synchronize()
