#!/usr/bin/env python3
"""CPU model (analysis only, no GPU) of the L2 fill traffic of the brick unprojection kernel under different
brick -> XCD assignments, for BASELINE configs[3] (10 views, 160x160x40, B=1) and configs[1]'s grid at B=1.
Each XCD: 4 MiB L2 = 32768 lines of 128 B, LRU at the granularity of a GENERATION of bricks in flight (32 CUs x 16
waves = 512 single-wave workgroups); a generation's distinct lines are fetched once unless still resident.
Optimistic (no thrash inside a generation, no other traffic in the L2), meant for RANKING assignments.
    python tools/sim_xcd_map.py [stress|root]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from selfpose3d_amd import synthetic as syn

which = sys.argv[1] if len(sys.argv) > 1 else "stress"
V, cube = (10, (160, 160, 40)) if which == "stress" else (5, (80, 80, 20))
(w, h), img = (240, 128), (960, 512)
X, Y, Z = cube
cams = syn.ring_cameras(V)
gx = np.linspace(-4000, 4000, X) + syn.SPACE_CENTER[0]
gy = np.linspace(-4000, 4000, Y) + syn.SPACE_CENTER[1]
gz = np.linspace(-1000, 1000, Z) + syn.SPACE_CENTER[2]
P = np.stack(np.meshgrid(gx, gy, gz, indexing="ij"), -1).reshape(-1, 3)
N = len(P)
a = img[0] / (200 * syn.get_scale(syn.ORIG_IMAGE, img)[0])
LPV = h * w // 2                      # 128-byte lines per view (2 pixels of 64 B)
tap = []
for c in range(V):
    px = syn._project_f64(P, cams[c])
    bound = (px[:, 0] >= 0) & (px[:, 1] >= 0) & (px[:, 0] < 1920) & (px[:, 1] < 1080)
    q = (px - np.array([960, 540])) * a + np.array([img[0] / 2, img[1] / 2])
    ix, iy = q[:, 0] * w / img[0], q[:, 1] * h / img[1]
    x0 = np.clip(np.floor(ix).astype(int), 0, w - 2); y0 = np.clip(np.floor(iy).astype(int), 0, h - 2)
    ids = [np.where(bound, c * LPV + ((y0 + dy) * w + (x0 + dx)) // 2, -1) for dy in (0, 1) for dx in (0, 1)]
    tap.append(np.stack(ids, 1))
tap = np.concatenate(tap, 1)          # (N, 4V) line ids or -1
vx, vy, vz = np.arange(N) // (Y * Z), (np.arange(N) // Z) % Y, np.arange(N) % Z
bx, by, bz = vx // 4, vy // 4, vz // 4
nbx, nby, nbz = X // 4, Y // 4, Z // 4
brick = (bz * nbx + bx) * nby + by     # the kernel's default order: z slowest, then x, then y
nbr = nbx * nby * nbz
order = np.argsort(brick, kind="stable")
vox_of = order.reshape(nbr, 64)       # voxels of brick i
TOTAL = V * LPV
maps_mb = TOTAL * 128 / 1e6
print(f"{which}: {V} views, grid {cube}, {nbr} bricks, heat-maps {maps_mb:.1f} MB, bound fraction {np.mean(tap[:, ::4] >= 0):.3f}")


def simulate(assign, seq=None, cap=32768, gen=512):
    """assign: (nbr,) XCD of every brick; seq: (nbr,) sort key inside an XCD (default: brick id).  -> fill MB, worst XCD"""
    fills = []
    for x in range(8):
        mine = np.flatnonzero(assign == x)
        if seq is not None:
            mine = mine[np.argsort(seq[mine], kind="stable")]
        last = np.full(TOTAL, -1, np.int64)          # generation of last use
        resident = np.zeros(TOTAL, bool)
        miss = 0
        for g0 in range(0, len(mine), gen):
            ids = tap[vox_of[mine[g0:g0 + gen]].reshape(-1)].reshape(-1)
            ids = np.unique(ids[ids >= 0])
            miss += int(np.count_nonzero(~resident[ids]))
            resident[ids] = True
            last[ids] = g0
            n = int(resident.sum())
            if n > cap:                               # evict the least recently used
                res = np.flatnonzero(resident)
                drop = res[np.argsort(last[res], kind="stable")[:n - cap]]
                resident[drop] = False
        fills.append(miss * 128 / 1e6)
    return sum(fills), max(fills), min(fills)


b = np.arange(nbr)
bzc, bxc, byc = b // (nbx * nby), (b // nby) % nbx, b % nby
res = {}
for K in (64, 256, 1024):
    res[f"chunks of {K} consecutive bricks, round robin (the kernel's map; default K = 256 at B=1)"] = simulate((b // K) % 8)
res["x slabs (1/8 of X each)"] = simulate(bxc * 8 // nbx)
res["y slabs"] = simulate(byc * 8 // nby)
res["2 x 4 (x, y) columns, all z"] = simulate((bxc * 2 // nbx) * 4 + byc * 4 // nby)
res["z slabs of the voxels' z (uneven: nbz = %d layers)" % nbz] = simulate(bzc * 8 // nbz)
ang = np.arctan2((byc + 0.5) / nby - 0.5, (bxc + 0.5) / nbx - 0.5)
res["8 angular wedges around the vertical axis, swept outwards"] = simulate(((ang + np.pi) / (2 * np.pi) * 8).astype(int) % 8,
                                                                            seq=np.hypot(bxc - nbx / 2, byc - nby / 2) * nbz + bzc)
res["8 angular wedges, swept z-major"] = simulate(((ang + np.pi) / (2 * np.pi) * 8).astype(int) % 8)
rad = np.hypot((bxc + 0.5) / nbx - 0.5, (byc + 0.5) / nby - 0.5)
rk = np.argsort(np.argsort(rad + 1e-6 * b))
res["8 rings of equal brick count around the vertical axis"] = simulate(rk * 8 // (nbx * nby * nbz) if False else (np.argsort(np.argsort(rad * 1000 + bzc * 1e-3 + 1e-9 * b)) * 8 // nbr))
res["checkerboard of 2x2x1-brick cells (worst case: everything everywhere)"] = simulate((bxc // 2 + byc // 2 + bzc) % 8)
print(f"\n| brick -> XCD assignment | L2 fills MB | x heat-maps | worst / best XCD MB |\n|---|---:|---:|---:|")
for k, (tot, mx, mn) in res.items():
    print(f"| {k} | {tot:.1f} | {tot / maps_mb:.2f} | {mx:.1f} / {mn:.1f} |")

# ---- second table: rectangular (x, y) column blocks, work balance and sweep order inside an XCD -----------------------
work_v = (tap[:, ::4] >= 0).sum(1) + 1.0                       # visible views per voxel (+ the fixed cost of a voxel)
work_b = work_v[vox_of].sum(1)
print(f"\n| assignment | sweep inside an XCD | L2 fills MB | x heat-maps | busiest XCD's work / mean |\n|---|---|---:|---:|---:|")


def report(name, assign, seqs):
    wk = np.array([work_b[assign == x].sum() for x in range(8)])
    for sname, seq in seqs.items():
        tot, mx, mn = simulate(assign, seq=seq)
        print(f"| {name} | {sname} | {tot:.1f} | {tot / maps_mb:.2f} | {wk.max() / wk.mean():.3f} |")


sweeps = {"z slowest, then x, then y (the kernel's order)": None,
          "x slowest, then y, then z (columns)": (bxc * nby + byc) * nbz + bzc,
          "y slowest, then x, then z": (byc * nbx + bxc) * nbz + bzc}
report("round robin chunks of 256", (b // 256) % 8, {"-": None})
report("2 x 4 blocks", (bxc * 2 // nbx) * 4 + byc * 4 // nby, sweeps)
report("4 x 2 blocks", (bxc * 4 // nbx) * 2 + byc * 2 // nby, sweeps)
report("8 x 1 (x slabs)", bxc * 8 // nbx, sweeps)
# interleaved: 16 blocks (4 x 4), XCD x gets blocks x and 15 - x (a far and a near one: balances the work)
blk = (bxc * 4 // nbx) * 4 + byc * 4 // nby
report("4 x 4 blocks, XCD x = blocks {x, 15 - x}", np.where(blk < 8, blk, 15 - blk), sweeps)

# ---- third table: octants (quadrants split along their diagonal): equal column counts, symmetric for ring rigs ---------
cx, cy = nbx / 2, nby / 2
u = np.where(bxc < cx, cx - 1 - bxc, bxc - cx).astype(int)        # distance from the centre lines, 0 .. nbx/2-1
v = np.where(byc < cy, cy - 1 - byc, byc - cy).astype(int)
quad = (bxc >= cx).astype(int) * 2 + (byc >= cy).astype(int)
upper = (v < u) | ((v == u) & (u % 2 == 0))
octant = quad * 2 + upper.astype(int)
print(f"\n| assignment | sweep inside an XCD | L2 fills MB | x heat-maps | busiest XCD's work / mean |\n|---|---|---:|---:|---:|")
report("8 octants (quadrants cut along the diagonal)", octant,
       {"z slowest, rows outwards": bzc * 10000 + np.maximum(u, v) * 100 + np.minimum(u, v),
        "rows outwards, z fastest": (np.maximum(u, v) * 100 + np.minimum(u, v)) * nbz + bzc})
report("2 x 2 quadrants, two XCDs interleaved by column parity", quad * 2 + ((bxc + byc) & 1), {"z slowest": None})
