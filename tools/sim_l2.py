#!/usr/bin/env python3
"""CPU model of one XCD's L2 (LRU over 128-B lines) under different tile orders of the unprojection kernel
(analysis only; no GPU).  One sample of the bench workload, its tiles served by `nx` XCDs."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from selfpose3d_amd import synthetic as syn

V, (w, h), img = 5, (240, 128), (960, 512)
X, Y, Z = syn.INITIAL_CUBE_SIZE
cams = syn.ring_cameras(V)
gx = np.linspace(-4000, 4000, X) + syn.SPACE_CENTER[0]
gy = np.linspace(-4000, 4000, Y) + syn.SPACE_CENTER[1]
gz = np.linspace(-1000, 1000, Z) + syn.SPACE_CENTER[2]
P = np.stack(np.meshgrid(gx, gy, gz, indexing="ij"), -1).reshape(-1, 3)
N = len(P)
a = img[0] / (200 * syn.get_scale(syn.ORIG_IMAGE, img)[0])
lines = []   # per view: (N, 4) line ids or -1
for c in range(V):
    px = syn._project_f64(P, cams[c])
    bound = (px[:, 0] >= 0) & (px[:, 1] >= 0) & (px[:, 0] < 1920) & (px[:, 1] < 1080)
    q = (px - np.array([960, 540])) * a + np.array([img[0] / 2, img[1] / 2])
    ix, iy = q[:, 0] * w / img[0], q[:, 1] * h / img[1]
    x0 = np.clip(np.floor(ix).astype(int), 0, w - 2); y0 = np.clip(np.floor(iy).astype(int), 0, h - 2)
    ids = []
    for dy in (0, 1):
        for dx in (0, 1):
            pix = (y0 + dy) * w + (x0 + dx)
            ids.append(np.where(bound, c * (h * w // 2) + pix // 2, -1))   # 2 pixels of 64 B per 128-B line
    lines.append(np.stack(ids, 1))
print("bound frac", np.mean([np.mean(l[:, 0] >= 0) for l in lines]))
tiles = N // 64

def simulate(order, inflight=512, cap=32768):
    """order: tile ids in dispatch order for this XCD"""
    lru = collections.OrderedDict(); miss = acc = 0
    for g0 in range(0, len(order), inflight):
        grp = order[g0:g0 + inflight]
        vox = (np.asarray(grp)[:, None] * 64 + np.arange(64)[None]).reshape(-1)
        for c in range(V):
            ids = lines[c][vox].reshape(-1)
            ids = ids[ids >= 0]
            # collapse immediate duplicates (L1 does that), keep order
            for i in ids.tolist():
                acc += 1
                if i in lru: lru.move_to_end(i)
                else:
                    miss += 1; lru[i] = 1
                    if len(lru) > cap: lru.popitem(last=False)
    return miss, acc

nx = 2
res = {}
# (1) chunks of K tiles dealt alternately
for K in (64, 256, 512, 1024):
    o = [t for t in range(tiles) if (t // K) % nx == 0]
    res[f"chunk{K}"] = simulate(o)
# (2) blocked order: tiles regrouped into (bx x by) column blocks, each XCD takes alternate blocks
def blocked(bx, by):
    # a tile is 64 consecutive voxels = 3.2 z-columns; approximate by ordering tiles by the block of their first voxel
    t = np.arange(tiles); v0 = t * 64
    xi = v0 // (Y * Z); yi = (v0 % (Y * Z)) // Z
    key = (xi // bx) * 1000 + (yi // by)
    blocks = {}
    for tt, k in zip(t, key): blocks.setdefault(k, []).append(tt)
    ks = sorted(blocks)
    return [tt for i, k in enumerate(ks) if i % nx == 0 for tt in blocks[k]]
for bx, by in ((8, 8), (16, 16), (20, 20), (40, 40), (10, 80), (80, 10)):
    res[f"block{bx}x{by}"] = simulate(blocked(bx, by))
# (3) z-banded: not expressible with 64-consecutive-voxel tiles; skip
for k, (m, a_) in res.items():
    print(f"{k:12s} misses {m:8d}  accesses {a_:9d}  miss-rate {m / a_:.3f}  fill MB {m * 128 / 1e6:.1f}")
print("compulsory lines per view-set:", len(set(np.concatenate([l[l >= 0] for l in lines]).tolist())) * 128 / 1e6, "MB")

# footprint overlap of two-way spatial splits of one sample (lines each half touches, all views)
vz = np.arange(N) % Z; vy = (np.arange(N) // Z) % Y; vx = np.arange(N) // (Y * Z)
def fp(mask):
    return set(np.concatenate([l[mask][l[mask] >= 0] for l in lines]).tolist())
for name, m in (("x<40", vx < 40), ("y<40", vy < 40), ("z<10", vz < 10), ("z<8", vz < 8), ("x+y<80", vx + vy < 80)):
    a_, b_ = fp(m), fp(~m)
    print(f"split {name:7s}: A {len(a_) * 128 / 1e6:.2f} MB  B {len(b_) * 128 / 1e6:.2f} MB  sum {(len(a_) + len(b_)) * 128 / 1e6:.2f} MB  union {len(a_ | b_) * 128 / 1e6:.2f} MB")
