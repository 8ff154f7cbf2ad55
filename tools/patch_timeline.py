#!/usr/bin/env python3
"""Per-wave s_memtime stamps of the LDS-staged patch kernel (measurement only; stamped copy of the library).

    python tools/patch_timeline.py [fine|stress|coarse] > gpurun_out/patch_timeline.json
"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from selfpose3d_amd import _lib, synthetic as syn
from selfpose3d_amd.camera_pack import pack_cameras
from selfpose3d_amd import build as _build
TL = os.path.join(ROOT, "selfpose3d_amd", "libsp3d_patchtl.so")
if "--build-only" in sys.argv or not os.path.exists(TL) or os.path.getmtime(TL) < os.path.getmtime(_build.LIB):
    _build.build_variant(TL, ["-DSP3D_PATCH_TL"])
    if "--build-only" in sys.argv:
        sys.exit(0)
_lib.LIB_PATH = TL
lib = _lib.load()
which = [a for a in sys.argv[1:] if not a.startswith("--")]
which = which[0] if which else "fine"
img, (w, h), J = (960, 512), (240, 128), 15
dev = torch.device("cuda:0")
if which == "fine":
    B, V, cube, gs = 10, 5, syn.FINE_CUBE_SIZE, syn.FINE_GRID_SIZE
    rng = np.random.default_rng(0)
    c = np.stack([rng.uniform(-1500, 1500, B), rng.uniform(-2000, 1000, B), rng.uniform(700, 1100, B)], 1)
    centers = torch.from_numpy(c.astype(np.float32)).to(dev)
elif which == "stress":
    B, V, cube, gs = 1, 10, (160, 160, 40), syn.SPACE_SIZE
    centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
else:
    B, V, cube, gs = 4, 5, syn.INITIAL_CUBE_SIZE, syn.SPACE_SIZE
    centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
meta = syn.make_meta(B, V, img)
cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
valid = torch.ones(B, dtype=torch.uint8, device=dev)
hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=7)]
packed = _lib.pack_heatmaps(hms, jp=16); views = [packed[c] for c in range(V)]
run = lambda: _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16, h, w, cube, gs, img, False,
                                 variant=128, channels_last=True)
for _ in range(5): run()
S = 32
nwave = 8 * 60000
buf = torch.zeros(nwave * S, dtype=torch.int64, device=dev)
lib.sp3d_debug_set_patch_timeline.argtypes = [ctypes.c_void_p]
assert lib.sp3d_debug_set_patch_timeline(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
assert lib.sp3d_debug_set_patch_timeline(None) == 0
t = buf.cpu().numpy().reshape(-1, S).astype(np.float64)
t = t[t[:, 0] != 0]
d = lambda a, b: float(np.mean(t[:, b] - t[:, a]))
out = {"workload": which, "waves": int(len(t)), "kernel_us_event": round(e0.elapsed_time(e1) * 1e3, 1),
       "cycles": {"life": d(0, 29), "setup+P1(0)+issue(0)": d(0, 1)}}
nv = min(V, 6)
for c in range(nv):
    b0 = 2 + 4 * c
    prev = 1 if c == 0 else 5 + 4 * (c - 1)
    out["cycles"][f"view{c}"] = {"P1_next": d(prev, b0), "wait_patch+tap_reads_issue": d(b0, b0 + 1),
                                 "wait_taps+issue_next_patch": d(b0 + 1, b0 + 2), "interp": d(b0 + 2, b0 + 3)}
out["cycles"]["fusion+stores"] = d(28, 29)
print(json.dumps(out, indent=1))
