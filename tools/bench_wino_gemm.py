#!/usr/bin/env python3
"""quarter-resolution Winograd layer: input transform / products (own split GEMM vs torch.bmm) / output transform, HIP events"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from selfpose3d_amd import _lib

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 2)

out = {}
for C in (128, 64):
    x = torch.randn(4, C, 20, 20, 5).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(128, C, 3, 3, 3) * 0.03).cuda()
    shift = torch.randn(128).cuda()
    U = _lib.wino_weights(w); W3 = _lib.wino_gemm_weights_split(U)
    T = 4 * 10 * 10 * 3
    V = torch.randn(64, T, C).cuda()
    out[f"C{C}"] = {"gemm_split_us": timeit(lambda: _lib.wino_gemm_split(V, W3)), "gemm_bmm_us": timeit(lambda: torch.bmm(V, U)),
                    "layer_split_us": timeit(lambda: _lib.wino_conv3d_(x, U, shift, 1, None, W3)),
                    "layer_bmm_us": timeit(lambda: _lib.wino_conv3d_(x, U, shift, 1, None))}
for path in sys.argv[1:]:          # measurement builds of sp3d_winograd.hip alone (-DSP3D_WG_ABLATE=mask): the GEMM only
    import ctypes
    L = ctypes.CDLL(os.path.abspath(path))
    f = L.sp3d_wino_gemm_split
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    V = torch.randn(64, 1200, 128).cuda()
    W3 = _lib.wino_gemm_weights_split((torch.randn(64, 128, 128) * 0.05).cuda())
    M = torch.empty(64, 1200, 128).cuda()
    out[os.path.basename(path)] = timeit(lambda: f(V.data_ptr(), W3.data_ptr(), M.data_ptr(), 64, 1200, 128, 128, None))
print(json.dumps(out, indent=1))
