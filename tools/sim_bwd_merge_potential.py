#!/usr/bin/env python3
"""How many memory atomics can ANY merge-before-scatter backward save on the root grid?  (analysis only, CPU)
The unprojection backward adds every (voxel, view, tap) contribution g * w to a 64-byte pixel record (16 channels) of
grad_hm; memory atomics retire at ~20.7 G (instruction, 64-byte segment) pairs per second on MI355X
(tools/global_atomic_bench.hip), so time >= segments / 20.7e9.  A workgroup that first merges the taps of a BLOCK of voxels
in LDS issues one atomic per DISTINCT pixel of the block's footprint.  This script counts, for BASELINE configs[1]'s grid
(80x80x20, 5 views, 240x128) and the dense grids, taps and distinct pixels per block shape, and the LDS patch a block needs.
    python tools/sim_bwd_merge_potential.py > profiles/r06_bwd_merge_potential.md"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from selfpose3d_amd import synthetic as syn

RATE = 20.7e9


def taps_of(V, cube, gsize, center):
    (w, h), img = (240, 128), (960, 512)
    X, Y, Z = cube
    cams = syn.ring_cameras(V)
    ax = [np.linspace(-gsize[i] / 2, gsize[i] / 2, cube[i]) + center[i] for i in range(3)]
    P = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    a = img[0] / (200 * syn.get_scale(syn.ORIG_IMAGE, img)[0])
    out = []
    for c in range(V):
        px = syn._project_f64(P, cams[c])
        with np.errstate(invalid="ignore"):
            bound = (px[:, 0] >= 0) & (px[:, 1] >= 0) & (px[:, 0] < 1920) & (px[:, 1] < 1080)
            q = (px - np.array([960, 540])) * a + np.array([img[0] / 2, img[1] / 2])
            ix, iy = q[:, 0] * w / img[0], q[:, 1] * h / img[1]
            x0 = np.clip(np.floor(np.nan_to_num(ix)).astype(int), 0, w - 2)
            y0 = np.clip(np.floor(np.nan_to_num(iy)).astype(int), 0, h - 2)
        ids = [np.where(bound, c * h * w + (y0 + dy) * w + (x0 + dx), -1) for dy in (0, 1) for dx in (0, 1)]
        out.append(np.stack(ids, 1))
    return np.stack(out, 1)          # (N, V, 4) pixel ids or -1


def table(name, V, cube, gsize, center, shapes):
    t = taps_of(V, cube, gsize, center)
    X, Y, Z = cube
    n = np.arange(X * Y * Z)
    vx, vy, vz = n // (Y * Z), (n // Z) % Y, n % Z
    total = int((t >= 0).sum())
    print(f"\n### {name}: {V} views, grid {cube}, {total / 1e6:.2f} M taps per sample = {total / RATE * 1e6:.0f} us of memory atomics per sample\n")
    print("| block of voxels merged before the scatter | taps / distinct pixels | atomics left per sample | floor us per sample | largest footprint (pixels of one view) | LDS patch, 16 ch x int64 |")
    print("|---|---:|---:|---:|---:|---:|")
    for (bx, by, bz) in shapes:
        key = ((vx // bx) * ((Y + by - 1) // by) + vy // by) * ((Z + bz - 1) // bz) + vz // bz
        nb = int(key.max()) + 1
        distinct, worst = 0, 0
        for c in range(V):
            ids = t[:, c, :]
            k = np.repeat(key[:, None], 4, 1)[ids >= 0].astype(np.int64)
            p = ids[ids >= 0].astype(np.int64)
            u = np.unique(k * (1 << 32) + p)
            distinct += len(u)
            worst = max(worst, int(np.bincount((u >> 32).astype(np.int64), minlength=nb).max()))
        print(f"| {bx} x {by} x {bz} | {total / distinct:.2f} | {distinct / 1e6:.2f} M | {distinct / RATE * 1e6:.0f} | {worst} | {worst * 128 / 1024:.0f} KB |")


print("# Merge potential of the unprojection backward (what a non-atomic or merged scatter could save)\n")
print(f"memory atomics: {RATE / 1e9:.1f} G (instruction, 64-byte segment) pairs per second (tools/global_atomic_bench.hip, profiles/r04_backward_kernels.md)")
table("root grid (BASELINE configs[1] / [2])", 5, (80, 80, 20), syn.SPACE_SIZE, syn.SPACE_CENTER,
      [(1, 1, 1), (4, 4, 4), (8, 8, 4), (8, 8, 20), (16, 16, 20), (80, 80, 20)])
table("person cube 64^3 (2 m)", 5, (64, 64, 64), syn.FINE_GRID_SIZE, (300.0, -800.0, 900.0),
      [(1, 1, 1), (4, 4, 4), (8, 8, 4), (16, 16, 8), (64, 64, 64)])
