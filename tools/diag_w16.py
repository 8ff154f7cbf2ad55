#!/usr/bin/env python3
"""Ablation timing of the half-resolution fused Winograd kernel (measurement only; never shipped): one library per
-DSP3D_W16_ABLATE=<mask>, a part of the kernel removed in each, timed with HIP events.
    python tools/diag_w16.py --build-only      # CPU box
    python tools/diag_w16.py > gpurun_out/diag_w16.json
"""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
MASKS = {"full": 0, "no_mfma": 1, "no_weight_loads": 2, "no_split": 4, "no_lds_reads": 8,
         "no_mfma_no_split": 5, "no_mfma_no_loads_no_lds": 11, "staging_epilogue_valu_only": 15, "mfma_only": 14}
LIBDIR = os.path.join(ROOT, "selfpose3d_amd", "ablate")


def lib_path(m):
    return os.path.join(LIBDIR, f"libsp3d_w16_{m}.so")


def build_all():
    from selfpose3d_amd import build as _b
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(_b.CSRC, "sp3d_winograd.hip")
    procs = [subprocess.Popen([_b.HIPCC] + _b.FLAGS + [f"-DSP3D_W16_ABLATE={m}", src, "-o", lib_path(m)])
             for m in sorted(set(MASKS.values()))]
    for p in procs:
        assert p.wait() == 0


if "--build-only" in sys.argv:
    build_all()
    sys.exit(0)
import torch
from selfpose3d_amd import _lib
if "--direct" in sys.argv:         # the direct convolution kernel: 2 no weight loads, 16 no result stores
    B, C, X, Y, Z = 4, 32, 80, 80, 20
    x = torch.randn(B, C, X, Y, Z).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(32, C, 3, 3, 3) * 0.05).cuda()
    W3 = _lib.conv_weights_split(w)
    shift = torch.randn(32).cuda()
    y = torch.empty(B, X, Y, Z, 32, device="cuda")
    out = {}
    for name, m in {"full": 0, "no_weight_loads": 2, "no_stores": 16, "no_weight_loads_no_stores": 18}.items():
        L = ctypes.CDLL(lib_path(m))
        f = L.sp3d_conv3_split
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
        run = lambda: f(x.data_ptr(), W3.data_ptr(), y.data_ptr(), shift.data_ptr(), None, 1, B, X, Y, Z, C, 32, None)
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record()
        torch.cuda.synchronize()
        out[name] = round(a.elapsed_time(b) * 1e3 / 20, 1)
    print(json.dumps(out, indent=1))
    sys.exit(0)
if "--full-res" in sys.argv:       # the full-resolution kernel (wino_fused3_kernel) under the same switches
    B, C, X, Y, Z = 4, 32, 80, 80, 20
    x = torch.randn(B, C, X, Y, Z).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(32, C, 3, 3, 3) * 0.05).cuda()
    U3 = _lib.wino_weights_split(_lib.wino_weights(w))
    shift = torch.randn(32).cuda()
    y = torch.empty(B, X, Y, Z, 32, device="cuda")
    out = {}
    for name, m in MASKS.items():
        L = ctypes.CDLL(lib_path(m))
        f = L.sp3d_wino_fused_split
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
        run = lambda: f(x.data_ptr(), U3.data_ptr(), y.data_ptr(), shift.data_ptr(), None, 1, B, X, Y, Z, C, 32, None)
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record()
        torch.cuda.synchronize()
        out[name] = round(a.elapsed_time(b) * 1e3 / 20, 1)
    print(json.dumps(out, indent=1))
    sys.exit(0)
B, C, X, Y, Z = 4, 64, 40, 40, 10
x = torch.randn(B, C, X, Y, Z).cuda().contiguous(memory_format=torch.channels_last_3d)
w = (torch.randn(64, C, 3, 3, 3) * 0.05).cuda()
U3 = _lib.wino_weights_split(_lib.wino_weights(w), 16)
shift = torch.randn(64).cuda()
y = torch.empty(B, X, Y, Z, 64, device="cuda")
out = {}
for name, m in MASKS.items():
    L = ctypes.CDLL(lib_path(m))
    f = L.sp3d_wino_fused_split64
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
    r = {}
    for nbw, ks in ((2, 2), (4, 2), (2, 1)):
        L.sp3d_debug_set_w16_nbw(nbw, ks)
        run = lambda: f(x.data_ptr(), U3.data_ptr(), y.data_ptr(), shift.data_ptr(), None, 1, B, X, Y, Z, C, 64, None)
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            run()
        b.record()
        torch.cuda.synchronize()
        r[f"nbw{nbw}_ks{ks}_us"] = round(a.elapsed_time(b) * 1e3 / 30, 1)
    out[name] = r
print(json.dumps(out, indent=1))
