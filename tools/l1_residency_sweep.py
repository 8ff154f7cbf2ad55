#!/usr/bin/env python3
"""Round-5 bounded experiment on the root-grid unprojection kernel: does L1 residency buy time?

The brick kernel's workgroups are made view-synchronous (tuning bit 10: a workgroup barrier per view) and / or fewer per CU
(bits 11-13: n x 20 KB of unused LDS per workgroup), with stacks of 5 bricks (default) or single bricks (bit 6), so that
the lines a CU's resident waves want at any one time shrink from ~280 KB towards the 32 KB of its L1.  Every variant is
bit-identical to the shipped kernel (asserted here).

    python tools/l1_residency_sweep.py                 # HIP-event timing of all variants, interleaved -> JSON
    python tools/l1_residency_sweep.py --only NAME     # 12 launches of one variant (target of rocprofv3 --pmc)
tools/l1_residency_sweep.sh runs both and writes the table (profiles/r05_l1_residency.md)."""
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

BASE = 120                       # library default for a channels-last result: pipelined (8) + one wave per tile (16) + bricks (32)
                                 # + every brick its own one-wave workgroup (64)
VSYNC, BALLAST, STACK = 1 << 10, 1 << 11, -64       # STACK: clear bit 6 -> a workgroup is a z-stack of 5 bricks (5 waves)
VARIANTS = {
    # name: (bits to set, bits to clear)
    "shipped: single-brick workgroups (16 resident per CU)": None,
    "single bricks, +20 KB LDS (6 per CU)": (1 * BALLAST, 0),
    "single bricks, +40 KB LDS (3 per CU)": (2 * BALLAST, 0),
    "single bricks, +60 KB LDS (2 per CU)": (3 * BALLAST, 0),
    "single bricks, +140 KB LDS (1 per CU)": (7 * BALLAST, 0),
    "stacks of 5 bricks (3 workgroups = 15 waves per CU)": (0, 64),
    "stacks, view-synchronous": (VSYNC, 64),
    "stacks, +40 KB LDS (2 per CU)": (2 * BALLAST, 64),
    "stacks, +40 KB LDS, view-synchronous": (2 * BALLAST | VSYNC, 64),
    "stacks, +80 KB LDS (1 per CU)": (4 * BALLAST, 64),
    "stacks, +80 KB LDS, view-synchronous": (4 * BALLAST | VSYNC, 64),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--list", action="store_true")
    a = ap.parse_args()
    if a.list:
        print("\n".join(VARIANTS))
        return
    dev = torch.device("cuda:0")
    B, V, J, img, (w, h) = 4, 5, 15, (960, 512), (240, 128)
    cube, gs = syn.INITIAL_CUBE_SIZE, syn.SPACE_SIZE
    meta = syn.make_meta(B, V, img)
    cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
    centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
    valid = torch.ones(B, dtype=torch.uint8, device=dev)
    hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=7)]
    packed = _lib.pack_heatmaps(hms, jp=16)
    views = [packed[c] for c in range(V)]
    default_variant = None

    def run(bits):
        variant = None if bits is None else ((base | bits[0]) & ~bits[1])
        return _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16, h, w, cube, gs, img, False,
                                  variant=variant, channels_last=True)[0]
    # the default's bit pattern: tuning bits are OR-ed onto it
    base = int(os.environ.get("SP3D_BASE_VARIANT", BASE))
    ref = run(None)
    torch.cuda.synchronize()
    if a.only:
        bits = VARIANTS[a.only]
        for _ in range(12):
            run(bits)
        torch.cuda.synchronize()
        print("done", a.only)
        return
    res = {}
    for name, bits in VARIANTS.items():
        out = run(bits)
        torch.cuda.synchronize()
        res[name] = {"bit_identical": bool(torch.equal(out, ref))}
    times = {n: [] for n in VARIANTS}
    for rep in range(5):
        for name, bits in VARIANTS.items():
            for _ in range(5):
                run(bits)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters // 5):
                run(bits)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) * 1e3 / (a.iters // 5))
    for name in VARIANTS:
        res[name]["us"] = round(float(np.median(times[name])), 2)
        res[name]["us_min"] = round(float(min(times[name])), 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
