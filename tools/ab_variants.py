#!/usr/bin/env python3
"""Within-process interleaved A/B of the unprojection kernel variants (HIP-event timing).

    python tools/ab_variants.py [--rounds 5 --iters 100] > gpurun_out/ab.json

Workloads: root-net coarse grid (B=4 and B=1, 5 views, 240x128), the V=10 160x160x40 stress
config, a batch of 10 fine 64^3 proposal cubes.  Reports median/min microseconds per launch and
algorithmic GB/s (4*B*(V*J*h*w + J*N) bytes, grids not requested).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from selfpose3d_amd import _lib, synthetic as syn
from selfpose3d_amd.camera_pack import pack_cameras


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--variants", type=str, default="24,56,120")
    ap.add_argument("--lib", type=str, default=None, help="a measurement build of the library (selfpose3d_amd/ablate/...)")
    ap.add_argument("--only", type=str, default=None, help="comma list of workloads")
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    dev = torch.device("cuda:0")
    variants = [int(v) for v in args.variants.split(",")]
    img, (w, h), J = (960, 512), (240, 128), 15
    workloads = {
        "coarse_b4_v5": dict(B=4, V=5, cube=syn.INITIAL_CUBE_SIZE, gs=syn.SPACE_SIZE, fine=False),
        "coarse_b1_v5": dict(B=1, V=5, cube=syn.INITIAL_CUBE_SIZE, gs=syn.SPACE_SIZE, fine=False),
        "stress_b1_v10": dict(B=1, V=10, cube=(160, 160, 40), gs=syn.SPACE_SIZE, fine=False),
        "fine_b10_v5": dict(B=10, V=5, cube=syn.FINE_CUBE_SIZE, gs=syn.FINE_GRID_SIZE, fine=True),
        # BASELINE configs[4]: 3-4 view rigs (Campus / Shelf style), fine per-person cubes; the bf16 rows are its storage mode
        "fine_b10_v3": dict(B=10, V=3, cube=syn.FINE_CUBE_SIZE, gs=syn.FINE_GRID_SIZE, fine=True),
        "fine_b10_v4": dict(B=10, V=4, cube=syn.FINE_CUBE_SIZE, gs=syn.FINE_GRID_SIZE, fine=True),
    }
    report = {}
    if args.only:
        workloads = {k: v for k, v in workloads.items() if k in args.only.split(",")}
    for name, wl in workloads.items():
        B, V, cube, gs = wl["B"], wl["V"], wl["cube"], wl["gs"]
        N = cube[0] * cube[1] * cube[2]
        meta = syn.make_meta(B, V, img)
        cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
        if wl["fine"]:
            rng = np.random.default_rng(0)
            c = np.stack([rng.uniform(-1500, 1500, B), rng.uniform(-2000, 1000, B), rng.uniform(700, 1100, B)], 1)
            centers = torch.from_numpy(c.astype(np.float32)).to(dev)
        else:
            centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
        valid = torch.ones(B, dtype=torch.uint8, device=dev)
        hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=7)]
        packed = _lib.pack_heatmaps(hms, jp=16)
        views = [packed[c] for c in range(V)]
        fns = {"planar": lambda: _lib.unproject_fwd(hms, _lib.LAYOUT_PLANAR, 0, cam, centers, valid, B, J, h, w, cube,
                                                    gs, img, False),
               "pack": lambda: _lib.pack_heatmaps(hms, jp=16, out=packed)}
        # library defaults (selfpose3d_amd/csrc/sp3d_unproject.hip: default_variant)
        fns["nhwc_default"] = lambda: _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, J, h, w, cube,
                                                         gs, img, False)
        fns["nhwc_default_cl"] = lambda: _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16, h, w,
                                                            cube, gs, img, False, channels_last=True)
        for v in variants:
            fns[f"nhwc_v{v}"] = (lambda v=v: _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, J,
                                                                h, w, cube, gs, img, False, variant=v))
            # channels-last result (all 16 packed channels: the layout MIOpen is fed)
            fns[f"nhwc_v{v}_cl"] = (lambda v=v: _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16,
                                                                   h, w, cube, gs, img, False, variant=v,
                                                                   channels_last=True))
        packed16 = _lib.pack_heatmaps(hms, jp=16, out_dtype=torch.bfloat16)
        views16 = [packed16[c] for c in range(V)]
        fns["nhwc_bf16_in"] = lambda: _lib.unproject_fwd(views16, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, J, h, w,
                                                         cube, gs, img, False)
        fns["nhwc_bf16_io"] = lambda: _lib.unproject_fwd(views16, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, J, h, w,
                                                         cube, gs, img, False, out_dtype=torch.bfloat16)
        fns["nhwc_bf16_io_chlast"] = lambda: _lib.unproject_fwd(views16, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16, h, w,
                                                                cube, gs, img, False, channels_last=True, out_dtype=torch.bfloat16)
        fns["pack_bf16"] = lambda: _lib.pack_heatmaps(hms, jp=16, out=packed16)
        times = {k: [] for k in fns}
        ref_out = None
        for k, fn in fns.items():
            timed(fn, 5)
            if k.endswith("_cl"):                   # every channels-last variant must produce the same bits
                out = fn()[0]
                torch.cuda.synchronize()
                if ref_out is None:
                    ref_out = out.clone()
                else:
                    assert torch.equal(out, ref_out), f"variant {k} differs from the default on {name}"
        for _ in range(args.rounds):
            for k, fn in fns.items():
                times[k].append(timed(fn, args.iters if k != "planar" else max(5, args.iters // 10)))
        alg = 4.0 * B * (V * J * h * w + J * N)
        # 2 bytes per element for tensors stored as bf16
        alg_by = {"nhwc_bf16_in": B * (2.0 * V * J * h * w + 4.0 * J * N), "nhwc_bf16_io": 2.0 * B * (V * J * h * w + J * N),
                  "nhwc_bf16_io_chlast": 2.0 * B * (V * J * h * w + J * N)}
        report[name] = {"algorithmic_MB": round(alg / 1e6, 2),
                        "algorithmic_MB_bf16_in": round(alg_by["nhwc_bf16_in"] / 1e6, 2),
                        "algorithmic_MB_bf16_io": round(alg_by["nhwc_bf16_io"] / 1e6, 2)}
        for k, ts in times.items():
            med, mn = float(np.median(ts)), float(np.min(ts))
            a = alg_by.get(k, alg)
            report[name][k] = {"median_us": round(med, 2), "min_us": round(mn, 2),
                               "alg_GBps_at_median": round(a / med / 1e3, 1) if not k.startswith("pack") else None,
                               "frac_of_8TBps": round(a / med / 1e3 / 8000.0, 4) if not k.startswith("pack") else None}
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
