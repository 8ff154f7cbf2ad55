import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F
from selfpose3d_amd import _lib
dev = torch.device("cuda:0")
def timed(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out = {}
w = (torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05)
U = _lib.wino_weights(w); s = torch.randn(32, device=dev)
for B in (1, 2, 3, 4, 6, 7, 8):
    x = torch.randn(B, 32, 80, 80, 20, device=dev).contiguous(memory_format=torch.channels_last_3d)
    out[f"B{B}_wgs{B*500}"] = round(timed(lambda: _lib.wino_fused_conv3d_(x, U, s, 1)), 1)
print(json.dumps(out))
