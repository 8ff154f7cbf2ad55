#!/usr/bin/env python3
"""Time the V2V stack (MIOpen) under different tensor layouts / paddings (fp32), forward only."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn
import torch.nn.functional as F
from selfpose3d_amd.v2v_net import V2VNet

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True


def timeit(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


res = {}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shape = (B, 15, 80, 80, 20)
x = torch.rand(shape, device=dev)
m = V2VNet(15, 1).eval().to(dev)
with torch.no_grad():
    res["v2v_ncdhw"] = timeit(lambda: m(x))
    mcl = V2VNet(15, 1).eval().to(dev).to(memory_format=torch.channels_last_3d)
    xcl = x.to(memory_format=torch.channels_last_3d)
    res["v2v_channels_last_3d"] = timeit(lambda: mcl(xcl))
    # first conv alone: Cin 15 vs 16, layouts
    for cin in (15, 16):
        conv = nn.Conv3d(cin, 16, 7, 1, 3).to(dev)
        xi = torch.rand((B, cin, 80, 80, 20), device=dev)
        res[f"conv7_cin{cin}_ncdhw"] = timeit(lambda: conv(xi))
        convc = conv.to(memory_format=torch.channels_last_3d)
        xic = xi.to(memory_format=torch.channels_last_3d)
        res[f"conv7_cin{cin}_cl3d"] = timeit(lambda: convc(xic))
    # 7^3 conv as 7 accumulated 2D-ish convs? (kernel (7,7,1) over z-shifted inputs) - same math, different kernels
    conv = nn.Conv3d(16, 16, 7, 1, 3).to(dev)
    xi = torch.rand((B, 16, 80, 80, 20), device=dev)
    def split_z():
        xp = F.pad(xi, (3, 3, 0, 0, 0, 0))
        out = None
        for k in range(7):
            o = F.conv3d(xp[..., k:k + 20], conv.weight[..., k:k + 1], None, 1, (3, 3, 0))
            out = o if out is None else out + o
        return out
    res["conv7_cin16_split_z7"] = timeit(split_z)
    # 3x3x3 64->64 at 40x40x10 and 128->128 at 20x20x5, 32->32 at 80x80x20
    for (c, s) in ((32, (80, 80, 20)), (64, (40, 40, 10)), (128, (20, 20, 5))):
        conv = nn.Conv3d(c, c, 3, 1, 1).to(dev)
        xi = torch.rand((B, c) + s, device=dev)
        res[f"conv3_c{c}_ncdhw"] = timeit(lambda: conv(xi))
        convc = conv.to(memory_format=torch.channels_last_3d); xic = xi.to(memory_format=torch.channels_last_3d)
        res[f"conv3_c{c}_cl3d"] = timeit(lambda: convc(xic))
print(json.dumps(res, indent=1))
