"""Launch sp3d_nms_proposals on the bench volume (80x80x20, k = 10) for rocprofv3 (tools/kstats.sh nms tools/run_nms.py)."""
import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from selfpose3d_amd import _lib
dev = torch.device("cuda:0")
x = torch.rand(1, 80, 80, 20, device=dev)
gs = [8000.0, 8000.0, 2000.0]; gc = [0.0, -500.0, 800.0]
for _ in range(30):
    _lib.nms_proposals(x, 10, gs, gc, 0.3)
torch.cuda.synchronize()
print("done")
