"""Is the training backbone faster in channels_last?  ResNet-50 PoseResNet forward + backward on the train leg's shape
(B*V = 10 images of 512x960) with plain per-batch BatchNorm (no view grouping: timing only), NCHW against channels_last,
MIOpen search on.  Round 5: decides whether the backbone's per-view BatchNorm moves to the grouped channels-last kernels."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from selfpose3d_amd.config import load_config
from selfpose3d_amd import pose_resnet
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
cfg = load_config(os.path.join(ROOT, "configs", "panoptic_synthetic_960x512_cam5.yaml"))
out = {}
for fmt_name, fmt in (("nchw", torch.contiguous_format), ("channels_last", torch.channels_last)):
    net = pose_resnet.get_pose_net(cfg, is_train=True).to(dev).train().to(memory_format=fmt)
    x = torch.randn(10, 3, 512, 960, device=dev).contiguous(memory_format=fmt)
    def step():
        y = net(x)
        y.square().mean().backward()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    out[fmt_name] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
    del net, x
    torch.cuda.empty_cache()
print(json.dumps({"backbone_fwd_bwd_ms": out}))
