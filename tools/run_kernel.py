#!/usr/bin/env python3
"""Launch the unprojection kernel(s) of one workload a few times (target for rocprofv3 --pmc / --kernel-trace).

    python tools/run_kernel.py --workload coarse_b4_v5 --variant 8 --iters 20 [--rotate 8]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from selfpose3d_amd import _lib, synthetic as syn
from selfpose3d_amd.camera_pack import pack_cameras

WORKLOADS = {
    "coarse_b4_v5": dict(B=4, V=5, cube=syn.INITIAL_CUBE_SIZE, gs=syn.SPACE_SIZE, fine=False),
    "coarse_b1_v5": dict(B=1, V=5, cube=syn.INITIAL_CUBE_SIZE, gs=syn.SPACE_SIZE, fine=False),
    "stress_b1_v10": dict(B=1, V=10, cube=(160, 160, 40), gs=syn.SPACE_SIZE, fine=False),
    "fine_b10_v5": dict(B=10, V=5, cube=syn.FINE_CUBE_SIZE, gs=syn.FINE_GRID_SIZE, fine=True),
    "fine_b10_v4": dict(B=10, V=4, cube=syn.FINE_CUBE_SIZE, gs=syn.FINE_GRID_SIZE, fine=True),      # BASELINE configs[4]
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="coarse_b4_v5")
    ap.add_argument("--variant", type=int, default=-1, help="-1: library default")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rotate", type=int, default=1, help="number of distinct input sets cycled through")
    ap.add_argument("--planar", action="store_true")
    ap.add_argument("--cl", action="store_true", help="channels-last result (16 channels)")
    ap.add_argument("--zdft", action="store_true", help="root grid only: the unprojection fused with the opening conv's z pass")
    ap.add_argument("--bf16", action="store_true", help="bf16 heat-maps and cubes (configs[4]: unproject_brick_h_kernel)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    wl = WORKLOADS[args.workload]
    img, (w, h), J = (960, 512), (240, 128), 15
    B, V, cube, gs = wl["B"], wl["V"], wl["cube"], wl["gs"]
    meta = syn.make_meta(B, V, img)
    cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
    if wl["fine"]:
        rng = np.random.default_rng(0)
        c = np.stack([rng.uniform(-1500, 1500, B), rng.uniform(-2000, 1000, B), rng.uniform(700, 1100, B)], 1)
        centers = torch.from_numpy(c.astype(np.float32)).to(dev)
    else:
        centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
    valid = torch.ones(B, dtype=torch.uint8, device=dev)
    sets = []
    for r in range(args.rotate):
        hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=7 + r)]
        sets.append((hms, _lib.pack_heatmaps(hms, jp=16, out_dtype=torch.bfloat16 if args.bf16 else torch.float32)))
    torch.cuda.synchronize()
    for it in range(args.iters):
        hms, packed = sets[it % args.rotate]
        if args.zdft:
            _lib.unproject_fwd_zdft([packed[c] for c in range(V)], 16, cam, centers, valid, B, J, h, w, cube, gs, img, 28)
        elif args.planar:
            _lib.unproject_fwd(hms, _lib.LAYOUT_PLANAR, 0, cam, centers, valid, B, J, h, w, cube, gs, img, False)
        else:
            _lib.unproject_fwd([packed[c] for c in range(V)], _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, J, h, w,
                               cube, gs, img, False, variant=None if args.variant < 0 else args.variant) if not args.cl else \
                _lib.unproject_fwd([packed[c] for c in range(V)], _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16, h, w,
                                   cube, gs, img, False, variant=None if args.variant < 0 else args.variant,
                                   channels_last=True, out_dtype=torch.bfloat16 if args.bf16 else torch.float32)
    torch.cuda.synchronize()
    print("done", args.workload, args.variant, args.iters)


if __name__ == "__main__":
    main()
