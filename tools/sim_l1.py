#!/usr/bin/env python3
"""CPU model (no GPU): how many DISTINCT 128-B heat-map lines does one wave's view-gather touch, per voxel-view, for
different shapes of the 64 voxels a wave owns?  (L1 misses of the unprojection kernels = this figure x voxel-views:
profiles/r02_pmc_cache_linear_vs_brick.json.)  Root grid, the 160x160x40 stress grid and a 64^3 person cube on the
synthetic 5/10-camera ring; line = 2 horizontally adjacent 64-B pixels of the (h,w,16) fp32 layout.

    python tools/sim_l1.py
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from selfpose3d_amd import synthetic as syn
def setup(V, cube, space, center, w=240, h=128, img=(960,512)):
    X, Y, Z = cube
    cams = syn.ring_cameras(V)
    gx = np.linspace(-space[0]/2, space[0]/2, X) + center[0]
    gy = np.linspace(-space[1]/2, space[1]/2, Y) + center[1]
    gz = np.linspace(-space[2]/2, space[2]/2, Z) + center[2]
    P = np.stack(np.meshgrid(gx, gy, gz, indexing="ij"), -1).reshape(-1, 3)
    a = img[0] / (200 * syn.get_scale(syn.ORIG_IMAGE, img)[0])
    lines = []
    for c in range(V):
        px = syn._project_f64(P, cams[c])
        bound = (px[:, 0] >= 0) & (px[:, 1] >= 0) & (px[:, 0] < 1920) & (px[:, 1] < 1080)
        px = np.nan_to_num(np.clip(px, -1, 1920))
        q = (px - np.array([960, 540])) * a + np.array([img[0] / 2, img[1] / 2])
        ix, iy = q[:, 0] * w / img[0], q[:, 1] * h / img[1]
        x0 = np.clip(np.floor(ix).astype(int), 0, w - 2); y0 = np.clip(np.floor(iy).astype(int), 0, h - 2)
        ids = []
        for dy in (0, 1):
            for dx in (0, 1):
                pix = (y0 + dy) * w + (x0 + dx)
                ids.append(np.where(bound, c * (h * w // 2) + pix // 2, -1))
        lines.append(np.stack(ids, 1))
    return lines
def groups(cube, shp):
    X, Y, Z = cube; tx, ty, tz = shp
    idx = np.arange(X*Y*Z).reshape(X, Y, Z)
    return idx.reshape(X//tx, tx, Y//ty, ty, Z//tz, tz).transpose(0,2,4,1,3,5).reshape(-1, tx*ty*tz)
def stats(lines, t):
    uniq = acc = gv = 0
    for L in lines:
        L = L[t].reshape(len(t), -1)
        L = np.sort(L, axis=1)
        valid = L >= 0
        newv = np.ones_like(L, bool); newv[:,1:] = L[:,1:] != L[:,:-1]
        uniq += (newv & valid).sum(); acc += valid.sum(); gv += valid.any(1).sum()
    return gv, acc, uniq
def run(name, V, cube, space, center, shapes):
    lines = setup(V, cube, space, center)
    N = cube[0]*cube[1]*cube[2]
    cur = np.arange(N).reshape(-1, 64)
    gv, acc, uq = stats(lines, cur)
    print(f"== {name}: voxel-views {acc//4}  ({acc/4/N/V:.2f} visible)")
    print(f"  current 64 consecutive: wave-views {gv}, lines/voxel-view {uq/(acc/4):.2f}")
    for shp in shapes:
        t = groups(cube, shp)
        gv, acc, uq = stats(lines, t)
        print(f"  {shp} ({np.prod(shp)} vox): group-views {gv}, lines/voxel-view {uq/(acc/4):.2f}")
run("coarse 80x80x20 V5", 5, (80,80,20), syn.SPACE_SIZE, syn.SPACE_CENTER,
    [(4,4,4),(4,8,2),(8,4,2),(8,8,1),(2,8,4),(2,2,4),(4,4,20),(8,8,20),(4,8,20),(8,8,4),(8,8,10),(16,16,20),(2,4,20),(4,4,10),(5,5,20),(10,10,20)])
run("stress 160x160x40 V10", 10, (160,160,40), syn.SPACE_SIZE, syn.SPACE_CENTER,
    [(4,4,4),(4,8,2),(8,8,1),(4,4,40),(8,8,40),(2,2,40), (8,8,8)])
run("fine 64^3 V5 center", 5, (64,64,64), syn.FINE_GRID_SIZE, (0.,-500.,800.),
    [(4,4,4),(4,8,2),(8,8,1),(2,2,16),(4,4,64),(8,8,8),(8,8,64),(2,4,64)])
