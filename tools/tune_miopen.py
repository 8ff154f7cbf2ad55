#!/usr/bin/env python3
"""Let MIOpen exhaustively tune its conv solvers for the V2V shapes of the bench workload and keep the
resulting user perf-db under <out>/ (MIOPEN_USER_DB_PATH).  Usage on the GPU box:
    python tools/tune_miopen.py gpurun_out/miopen_db [enforce]
"""
import os
import sys
import time

out = os.path.abspath(sys.argv[1])
os.makedirs(out, exist_ok=True)
os.environ["MIOPEN_USER_DB_PATH"] = out
os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(out, "cache")
if len(sys.argv) > 2:
    os.environ["MIOPEN_FIND_ENFORCE"] = sys.argv[2]      # 3 = SEARCH, 4 = SEARCH_DB_UPDATE
    os.environ["MIOPEN_FIND_MODE"] = "1"                 # NORMAL (full) find
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from selfpose3d_amd.v2v_net import V2VNet

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
net = V2VNet(15, 1).eval().to(dev).to(memory_format=torch.channels_last_3d)
x = torch.rand(4, 16, 80, 80, 20, device=dev).contiguous(memory_format=torch.channels_last_3d)
t0 = time.time()
with torch.no_grad():
    net(x)
torch.cuda.synchronize()
print(f"first forward (find/tune) took {time.time() - t0:.1f} s", flush=True)
with torch.no_grad():
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        net(x)
    torch.cuda.synchronize()
print(f"V2V fwd B=4: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms")
print(os.listdir(out))
