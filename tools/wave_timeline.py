#!/usr/bin/env python3
"""Per-wave s_memtime timeline of the pipelined unprojection kernel (measurement only)."""
import ctypes, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from selfpose3d_amd import _lib, synthetic as syn
from selfpose3d_amd.camera_pack import pack_cameras
from selfpose3d_amd import build as _build
TL = os.path.join(ROOT, "selfpose3d_amd", "libsp3d_timeline.so")      # stamped build, separate from the shipped library
if "--build-only" in sys.argv or not os.path.exists(TL) or os.path.getmtime(TL) < os.path.getmtime(_build.LIB):
    _build.build_variant(TL, ["-DSP3D_TIMELINE"])
    if "--build-only" in sys.argv:
        sys.exit(0)
_lib.LIB_PATH = TL
lib = _lib.load()
img, (w, h), J = (960, 512), (240, 128), 15
dev = torch.device("cuda:0")
B, V = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 5
cube, gs = syn.INITIAL_CUBE_SIZE, syn.SPACE_SIZE
meta = syn.make_meta(B, V, img)
cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
valid = torch.ones(B, dtype=torch.uint8, device=dev)
hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=7)]
packed = _lib.pack_heatmaps(hms, jp=16); views = [packed[c] for c in range(V)]
run = lambda: _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, J, h, w, cube, gs, img, False)
for _ in range(5): run()
nblk = 8 * 4096
S = 32
buf = torch.zeros(nblk * S, dtype=torch.int64, device=dev)
lib.sp3d_debug_set_timeline.argtypes = [ctypes.c_void_p]
assert lib.sp3d_debug_set_timeline(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
assert lib.sp3d_debug_set_timeline(None) == 0
t = buf.cpu().numpy().reshape(-1, S)
t = t[t[:, 0] != 0]
start, p1, end = t[:, 0], t[:, 1], t[:, 30]
life = end - start
span = end.max() - start.min()
us = e0.elapsed_time(e1) * 1e3
# stamps per view c: [2+4c] view start, [3+4c] tap loads issued, [4+4c] P1(c+1) done; next view start = FMAs done
nxt = lambda c: t[:, 2 + 4 * (c + 1)] if c + 1 < V else end
res = {"waves": int(len(t)), "kernel_us_event": round(us, 1), "kernel_span_ticks": int(span), "ticks_per_us": round(span / us, 1),
       "wave_life_ticks": {"mean": float(life.mean()), "p10": float(np.percentile(life, 10)), "p90": float(np.percentile(life, 90))},
       "avg_resident_waves_per_cu": round(float(life.sum() / span / 256), 2),
       "P1_0_ticks_mean": float((p1 - start).mean()),
       "issue_loads_ticks_mean": [float((t[:, 3 + 4 * c] - t[:, 2 + 4 * c]).mean()) for c in range(V)],
       "P1_next_ticks_mean": [float((t[:, 4 + 4 * c] - t[:, 3 + 4 * c]).mean()) for c in range(V)],
       "wait_fma_ticks_mean": [float((nxt(c) - t[:, 4 + 4 * c]).mean()) for c in range(V)],
       "start_spread_ticks": {"p50": float(np.percentile(start - start.min(), 50)), "p90": float(np.percentile(start - start.min(), 90)), "max": float((start - start.min()).max())},
       "life_by_bound_views": {int(k): float(life[t[:, 31] == k].mean()) for k in np.unique(t[:, 31])}}
print(json.dumps(res, indent=1))
