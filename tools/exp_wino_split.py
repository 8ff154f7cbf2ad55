"""fused Winograd kernel: fp32 MFMA against the exact three-piece bf16 split, time and error vs float64 (bench shapes)"""
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from selfpose3d_amd import _lib  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) * 1e3 / iters, 1)


out = {}
for name, (B, C, X, Y, Z, mode) in {} if "--half" in sys.argv else {"root_32_relu": (4, 32, 80, 80, 20, 1), "root_32_res": (4, 32, 80, 80, 20, 2),
                                    "root_16_relu": (4, 16, 80, 80, 20, 1), "pose8_32_res": (8, 32, 64, 64, 64, 2)}.items():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, X, Y, Z, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(32, C, 3, 3, 3, generator=g) * 0.05).cuda()
    shift = torch.randn(32, generator=g).cuda()
    res = torch.randn(B, 32, X, Y, Z, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d) if mode >= 2 else None
    U = _lib.wino_weights(w)
    U3 = _lib.wino_weights_split(U)
    W3 = _lib.conv_weights_split(w)
    r = {"fp32_mfma_us": timeit(lambda: _lib.wino_fused_conv3d_(x, U, shift, mode, res)),
         "bf16x3_us": timeit(lambda: _lib.wino_fused_conv3d_(x, U, shift, mode, res, U3)),
         "direct_bf16x3_us": timeit(lambda: _lib.conv3_split_(x, W3, shift, mode, res))}
    if C == 32:
        x3 = _lib.conv3_split_(x, W3, shift, 1, want_f32=False, want_s3=True)[1]
        o3 = _lib.conv3_s3_empty(B, X, Y, Z, 32, x.device)          # reused: allocation + zeroing are not part of a layer
        dm = (X, Y, Z)
        r["direct_s3_in_f32_out_us"] = timeit(lambda: _lib.conv3_split_(None, W3, shift, mode, res, x_s3=x3, dims=dm))
        r["direct_s3_in_both_out_us"] = timeit(lambda: _lib.conv3_split_(None, W3, shift, mode, res, x_s3=x3, want_s3=True, out_s3=o3, dims=dm))
        r["direct_s3_in_s3_out_us"] = timeit(lambda: _lib.conv3_split_(None, W3, shift, mode, res, x_s3=x3, want_f32=False, want_s3=True, out_s3=o3, dims=dm))
        r["direct_f32_in_s3_out_us"] = timeit(lambda: _lib.conv3_split_(x, W3, shift, mode, res, want_f32=False, want_s3=True, out_s3=o3))
    if B * X * Y * Z <= 600000:
        ref = F.conv3d(x[:1].double(), w.double(), padding=1) + shift.double().view(1, 32, 1, 1, 1)
        if mode == 2:
            ref = (ref + res[:1].double())
        ref = ref.clamp_min(0)
        y32 = _lib.wino_fused_conv3d_(x, U, shift, mode, res)[:1].double()
        y3 = _lib.wino_fused_conv3d_(x, U, shift, mode, res, U3)[:1].double()
        d = F.conv3d(x[:1], w, padding=1) + shift.view(1, 32, 1, 1, 1)
        if mode == 2:
            d = d + res[:1]
        d = d.clamp_min(0).double()
        yd = _lib.conv3_split_(x, W3, shift, mode, res)[:1].double()
        r.update(max_err_direct_bf16x3=float((yd - ref).abs().max()))
        r.update(max_err_fp32_mfma=float((y32 - ref).abs().max()), max_err_bf16x3=float((y3 - ref).abs().max()),
                 max_err_direct_fp32_conv=float((d - ref).abs().max()), ref_max=float(ref.abs().max()))
    out[name] = r
for name, (B, C, X, Y, Z, mode) in {"half_64_res": (4, 64, 40, 40, 10, 2), "half_64_relu": (4, 64, 40, 40, 10, 1),
                                    "half_32to64_relu": (4, 32, 40, 40, 10, 1), "pose8_half_64": (8, 64, 32, 32, 32, 2)}.items():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, X, Y, Z, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(64, C, 3, 3, 3, generator=g) * 0.05).cuda()
    shift = torch.randn(64, generator=g).cuda()
    res = torch.randn(B, 64, X, Y, Z, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d) if mode >= 2 else None
    U = _lib.wino_weights(w)
    U3 = _lib.wino_weights_split(U, 16)
    r = {}
    for nbw, ks in ((2, 2),):
        _lib.load().sp3d_debug_set_w16_nbw(nbw, ks)
        r[f"fused_bf16x3_nbw{nbw}_ks{ks}_us"] = timeit(lambda: _lib.wino_fused_conv3d_(x, U, shift, mode, res, U3))
    _lib.load().sp3d_debug_set_w16_nbw(0, 1)
    r["fused_bf16x3_auto_us"] = timeit(lambda: _lib.wino_fused_conv3d_(x, U, shift, mode, res, U3))
    out[name] = {**r, "three_launch_fp32_us": timeit(lambda: _lib.wino_conv3d_(x, U, shift, mode, res)),
                 "miopen_direct_us": timeit(lambda: F.conv3d(x, w.contiguous(memory_format=torch.channels_last_3d), None, 1, 1))}
print(json.dumps(out, indent=1))
