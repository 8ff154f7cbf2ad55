import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from selfpose3d_amd import _lib
dev = torch.device("cuda:0")
x = torch.rand(4, 80, 80, 20, device=dev)
def timed(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
print(json.dumps({"nms_topk_b4_us": round(timed(lambda: _lib.nms_topk(x, 10, [8000., 8000., 2000.], [0., -500., 800.])), 1)}))
