#!/usr/bin/env python3
"""per-phase cycle stamps of the fused Winograd kernel (stamped build, measurement only)"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from selfpose3d_amd import _lib, build as _build
TL = os.path.join(ROOT, "selfpose3d_amd", "libsp3d_wftl.so")
if "--build-only" in sys.argv or not os.path.exists(TL):
    _build.build_variant(TL, ["-DSP3D_WF_TIMELINE"])
    if "--build-only" in sys.argv:
        sys.exit(0)
_lib.LIB_PATH = TL
lib = _lib.load()
dev = torch.device("cuda:0")
B = 4
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
U = _lib.wino_weights(w); s = torch.randn(32, device=dev)
x = torch.randn(B, 32, 80, 80, 20, device=dev).contiguous(memory_format=torch.channels_last_3d)
for _ in range(3): _lib.wino_fused_conv3d_(x, U, s, 1)
buf = torch.zeros(2000 * 80, dtype=torch.int64, device=dev)
lib.sp3d_debug_wino_fused_timeline.argtypes = [ctypes.c_void_p]
assert lib.sp3d_debug_wino_fused_timeline(buf.data_ptr()) == 0
_lib.wino_fused_conv3d_(x, U, s, 1); torch.cuda.synchronize()
lib.sp3d_debug_wino_fused_timeline(None)
t = buf.cpu().numpy().reshape(-1, 80).astype(np.float64)
t = t[t[:, 0] != 0]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); _lib.wino_fused_conv3d_(x, U, s, 1); e1.record(); torch.cuda.synchronize()
res = {"waves": len(t), "kernel_us": round(e0.elapsed_time(e1) * 1e3, 1), "total_ticks_per_wave": float((t[:, 5] - t[:, 0]).mean()),
       "prep(ready - prev acc)": float(np.mean([(t[:, 8 + 4 * (jk + 1)] - t[:, 10 + 4 * jk]).mean() for jk in range(15)])),
       "mfma_issue(issued - ready)": float(np.mean([(t[:, 9 + 4 * jk] - t[:, 8 + 4 * jk]).mean() for jk in range(16)])),
       "accumulate(acc - issued)": float(np.mean([(t[:, 10 + 4 * jk] - t[:, 9 + 4 * jk]).mean() for jk in range(16)])),
       "step_period": float(np.mean([(t[:, 8 + 4 * (jk + 1)] - t[:, 8 + 4 * jk]).mean() for jk in range(15)]))}
print(json.dumps(res, indent=1))
