#!/usr/bin/env python3
"""7x7x7 stride-1 'same' Conv3d on the V2V grids: MIOpen direct conv vs rFFT -> sp3d_freq_contract -> irFFT
(rocFFT through torch.fft).  Reports time per stage and max |diff| against float64 (measurement only)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from selfpose3d_amd import _lib
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)


def timed(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(B, C, O, grid, k, sizes):
    X, Y, Z = grid
    x = torch.rand(B, C, X, Y, Z, device=dev)
    w = torch.randn(O, C, k, k, k, device=dev) * 0.05
    p = k // 2
    Cp = (C + 3) // 4 * 4
    xcl = F.pad(x, (0, 0, 0, 0, 0, 0, 0, Cp - C)).contiguous(memory_format=torch.channels_last_3d)
    wcl = F.pad(w, (0, 0, 0, 0, 0, 0, 0, Cp - C)).contiguous(memory_format=torch.channels_last_3d)
    ref = F.conv3d(xcl, wcl, padding=p)
    out = {"direct_cl3d_us": round(timed(lambda: F.conv3d(xcl, wcl, padding=p)), 1)}
    ref64 = F.conv3d(x.double(), w.double(), padding=p)
    out["direct_err_vs_f64"] = float((ref.double() - ref64).abs().max())
    for S in sizes:
        Wf = torch.conj(torch.fft.rfftn(w, s=S, dim=(2, 3, 4))).resolve_conj().contiguous()     # (O,C,F...)
        def fft_conv():
            xp = F.pad(x, (p, S[2] - Z - p, p, S[1] - Y - p, p, S[0] - X - p))
            Xf = torch.fft.rfftn(xp, dim=(2, 3, 4))
            Yf = _lib.freq_contract(Xf, Wf)
            y = torch.fft.irfftn(Yf, s=S, dim=(2, 3, 4))[..., :X, :Y, :Z]
            return y.contiguous(memory_format=torch.channels_last_3d)
        y = fft_conv()
        key = "fft_%dx%dx%d" % S
        out[key + "_us"] = round(timed(fft_conv), 1)
        out[key + "_err_vs_f64"] = float((y.double() - ref64).abs().max())
        xp = F.pad(x, (p, S[2] - Z - p, p, S[1] - Y - p, p, S[0] - X - p))
        out[key + "_pad_us"] = round(timed(lambda: F.pad(x, (p, S[2] - Z - p, p, S[1] - Y - p, p, S[0] - X - p))), 1)
        out[key + "_rfftn_us"] = round(timed(lambda: torch.fft.rfftn(xp, dim=(2, 3, 4))), 1)
        Xf = torch.fft.rfftn(xp, dim=(2, 3, 4))
        out[key + "_contract_us"] = round(timed(lambda: _lib.freq_contract(Xf, Wf)), 1)
        Yf = _lib.freq_contract(Xf, Wf)
        out[key + "_irfftn_us"] = round(timed(lambda: torch.fft.irfftn(Yf, s=S, dim=(2, 3, 4))), 1)
        yy = torch.fft.irfftn(Yf, s=S, dim=(2, 3, 4))
        out[key + "_crop_cl_us"] = round(timed(lambda: yy[..., :X, :Y, :Z].contiguous(memory_format=torch.channels_last_3d)), 1)
    out["ref_abs_max"] = float(ref.abs().max())
    return out


res = {"root_b4_c15": run(4, 15, 16, (80, 80, 20), 7, [(88, 88, 28), (88, 88, 26), (88, 88, 32), (90, 90, 28), (96, 96, 28), (96, 96, 32), (100, 100, 28), (104, 104, 28), (112, 112, 28), (128, 128, 32)]),
       "pose_p6_c15": run(6, 15, 16, (64, 64, 64), 7, [(70, 70, 72), (72, 72, 72), (80, 80, 80), (75, 75, 72), (77, 77, 72), (96, 96, 96)])}
print(json.dumps(res, indent=1))
