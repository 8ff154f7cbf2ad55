#!/usr/bin/env python3
"""per-item s_memtime stamps of workgroup 0 of the direct convolution kernel (measurement build -DSP3D_CD_TIMELINE):
consumer waves 0-3: [0] tap loop start, [1] tap loop end, [2] before barrier, [3] after barrier; producer waves 4-5: [0] start
of staging, [2] before barrier, [3] after.    python tools/conv3_timeline.py --build-only | (GPU) python tools/conv3_timeline.py"""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "selfpose3d_amd", "ablate", "libsp3d_cdtl.so")
if "--build-only" in sys.argv:
    from selfpose3d_amd import build as _b
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call([_b.HIPCC] + _b.FLAGS + ["-DSP3D_CD_TIMELINE", os.path.join(_b.CSRC, "sp3d_winograd.hip"), "-o", LIB])
    subprocess.check_call([_b.HIPCC] + _b.FLAGS + ["-DSP3D_CD_TIMELINE", "-DSP3D_W16_ABLATE=16", os.path.join(_b.CSRC, "sp3d_winograd.hip"), "-o", LIB.replace(".so", "_nostore.so")])
    for tag, fl in (("_nt", ['-DSP3D_CD_DMA_POLICY=" nt"']), ("_sc", ['-DSP3D_CD_DMA_POLICY=" sc0 sc1"']), ("_same", ["-DSP3D_CD_DMA_SAME"]), ("_sl2", ["-DSP3D_CD_DMA_SLEEP=2"]), ("_sl4", ["-DSP3D_CD_DMA_SLEEP=4"]), ("_sl8", ["-DSP3D_CD_DMA_SLEEP=8"])):
        if "--only" in sys.argv and tag[1:] not in sys.argv:
            continue
        subprocess.check_call([_b.HIPCC] + _b.FLAGS + ["-DSP3D_CD_TIMELINE"] + fl + [os.path.join(_b.CSRC, "sp3d_winograd.hip"), "-o", LIB.replace(".so", tag + ".so")])
    sys.exit(0)
import torch
from selfpose3d_amd import _lib
B, C, X, Y, Z = 4, 32, 80, 80, 20
x = torch.randn(B, C, X, Y, Z).cuda().contiguous(memory_format=torch.channels_last_3d)
w = (torch.randn(32, C, 3, 3, 3) * 0.05).cuda()
W3 = _lib.conv_weights_split(w)
shift = torch.randn(32).cuda()
y = torch.empty(B, X, Y, Z, 32, device="cuda")
VAR = [a[6:] for a in sys.argv if a.startswith("--var=")]
L = ctypes.CDLL(LIB.replace(".so", "_nostore.so") if "--no-store" in sys.argv else (LIB.replace(".so", "_" + VAR[0] + ".so") if VAR else LIB))
f = L.sp3d_conv3_split_ex
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
x3 = _lib.conv3_split_(x, W3, shift, 1, want_f32=False, want_s3=True)[1] if "--s3" in sys.argv else None      # zero-bordered
tl = torch.zeros(8 * 64 * 4, dtype=torch.int64, device="cuda")
run = lambda: f(None if x3 is not None else x.data_ptr(), x3.data_ptr() if x3 is not None else None, W3.data_ptr(),
                y.data_ptr(), None, shift.data_ptr(), None, 1, B, X, Y, Z, C, 32, None)
for _ in range(3):
    assert run() == 0
torch.cuda.synchronize()
assert L.sp3d_debug_conv3_timeline(ctypes.c_void_p(tl.data_ptr())) == 0
assert run() == 0
torch.cuda.synchronize()
t = tl.cpu().view(8, 64, 4)
w0, c0, w1, c1 = int(t[7, 62, 0]), int(t[7, 62, 1]), int(t[7, 63, 0]), int(t[7, 63, 1])
t[7, 62] = 0
t[7, 63] = 0
if w1 > w0:
    print(json.dumps({"wall_us_100MHz_counter": (w1 - w0) / 100.0, "s_memtime_ticks": c1 - c0, "ticks_per_us": round((c1 - c0) / ((w1 - w0) / 100.0), 1)}))
t0 = int(t[t > 0].min())
rows = []
for item in range(16):
    r = {"item": item}
    for wv in (0, 3, 4, 5):
        a = [int(v) - t0 if int(v) > 0 else None for v in t[wv, item]]
        r[f"w{wv}"] = a
    rows.append(r)
if "--rows" in sys.argv:
    for r in rows:
        print(json.dumps(r))
import statistics
c = t[0]
tap = [int(c[i, 1] - c[i, 0]) for i in range(16)]
epi = [int(c[i, 2] - c[i, 1]) for i in range(16)]
bar = [int(c[i, 3] - c[i, 2]) for i in range(16)]
p = t[4]
prod = [int(p[i, 2] - p[i, 0]) for i in range(16)]
prod_split = [int(p[i, 1] - p[i, 3]) for i in range(16)]
prod_wait = [int(p[i, 3] - p[i, 0]) for i in range(16)]
prod_issue = [int(p[i, 2] - p[i, 1]) for i in range(16)]
print(json.dumps({"consumer_tap_loop_clk": tap, "consumer_epilogue_clk": epi, "consumer_barrier_wait_clk": bar,
                  "producer_stage_clk": prod, "producer_load_wait_clk": prod_wait, "producer_split_write_clk": prod_split, "producer_issue_next_clk": prod_issue, "total_clk": int(t.max()) - t0}))
