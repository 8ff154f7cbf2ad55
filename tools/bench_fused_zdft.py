#!/usr/bin/env python3
"""A/B of the root grid's unprojection + z pass (BASELINE configs[1]: B=4, 5 views, 240x128 -> 80x80x20), HIP events,
back-to-back launches on warm inputs AND behind a 512 MiB fill (cold caches, closer to the step):
  two kernels : sp3d_unproject_fwd (channels-last cubes) -> sp3d_zdft_fwd_cl -> sp3d_cfft2d_ex(rows_in = 80)
  fused       : sp3d_unproject_fwd_zdft -> sp3d_cfft2d_88_tiled
    python tools/bench_fused_zdft.py > gpurun_out/r06_fused_zdft_ab.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from selfpose3d_amd import _lib, synthetic as syn
from selfpose3d_amd.camera_pack import pack_cameras

if os.environ.get("SP3D_LIB"):                      # a measurement build instead of the shipped library
    _lib.LIB_PATH = os.path.abspath(os.environ["SP3D_LIB"])
dev = torch.device("cuda:0")
B, V, J, img, hm, cube = 4, 5, 15, (960, 512), (240, 128), (80, 80, 20)
S = (88, 88, 28)
meta = syn.make_meta(B, V, img)
cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
cen = torch.tensor([list(syn.SPACE_CENTER)] * B, dtype=torch.float32, device=dev)
val = torch.ones(B, dtype=torch.uint8, device=dev)
packed = _lib.pack_heatmaps([x.to(dev) for x in syn.random_heatmaps(B, V, J, hm[1], hm[0], seed=0)], jp=16)
views = [packed[c] for c in range(V)]
gs = list(syn.SPACE_SIZE)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def unproj():
    return _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, cen, val, B, 16, hm[1], hm[0], cube, gs, img, False, channels_last=True)[0]


def fused():
    return _lib.unproject_fwd_zdft(views, 16, cam, cen, val, B, J, hm[1], hm[0], cube, gs, img, 28)


cubes = unproj()
spec2 = _lib.zdft_fwd_cl(cubes, J, S)
spec1 = fused()


def t(fn, n=200, cold=False):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    if not cold:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return round(1e3 * e0.elapsed_time(e1) / n, 2)
    n = 40
    for _ in range(n):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return round(1e3 * tot / n, 2)


out = {"workload": "configs[1] root grid, B=4, V=5, J=15, 240x128 -> 80x80x20", "unit": "us"}
for cold in (False, True):
    k = "cold" if cold else "warm"
    out[k] = {
        "unproject_cl": t(unproj, cold=cold),
        "zdft_fwd_cl": t(lambda: _lib.zdft_fwd_cl(cubes, J, S), cold=cold),
        "cfft2d_padded": t(lambda: _lib.cfft2d_(spec2, False, rows_in=80), cold=cold),
        "unproject_zdft_fused": t(fused, cold=cold),
        "cfft2d_88_tiled": t(lambda: _lib.cfft2d_88_tiled(spec1, 80, 80), cold=cold),
        "chain_two_kernels": t(lambda: _lib.cfft2d_(_lib.zdft_fwd_cl(unproj(), J, S), False, rows_in=80), cold=cold),
        "chain_fused": t(lambda: _lib.cfft2d_88_tiled(fused(), 80, 80), cold=cold),
    }
# round 6, second step: the channel contraction with the weight spectrum's y transform rebuilt per bin
from selfpose3d_amd.v2v_net import V2VNet, _FoldedV2V
net = V2VNet(15, 1).to(dev).eval()
plan = _FoldedV2V(net); plan._build(); plan.key = plan._key(net)
w0, _s0 = plan.t["front"]
Wz = plan._weights_z(w0, S); Tt, tw = plan._weights_ty(w0, S)
Xs = _lib.cfft2d_88_tiled(spec1, 80, 80)
for cold in (False, True):
    k = "cold" if cold else "warm"
    out[k]["freq_contract_full_spectrum"] = t(lambda: _lib.freq_contract(Xs, Wz), cold=cold)
    out[k]["freq_contract_ty"] = t(lambda: _lib.freq_contract_ty(Xs, Tt, tw), cold=cold)
alg = 4 * B * (V * J * hm[0] * hm[1] + J * 80 * 80 * 20)
out["algorithmic_bytes_unprojection"] = alg
out["bytes_fused_kernel_actually_moves"] = 4 * B * V * 16 * hm[0] * hm[1] + 8 * B * J * 15 * 80 * 80
out["frac_hbm_warm"] = {"unproject_cl": round(alg / (out["warm"]["unproject_cl"] * 1e-6) / 8e12, 4),
                        "fused_on_the_same_bytes": round(alg / (out["warm"]["unproject_zdft_fused"] * 1e-6) / 8e12, 4)}
print(json.dumps(out, indent=1))
