#!/usr/bin/env python3
"""CPU model behind profiles/r04_backward_kernels.md (no GPU): for the block-merge backward scatter on a 64^3 person cube,
  * per voxel-block shape: size of the tap rectangle of a (block, view), distinct pixels touched vs taps issued (= by how much
    merging in LDS cuts the memory atomics), share of rectangles above a patch capacity;
  * per lane-to-voxel map and patch row stride: lanes on the busiest of the 32 double-word LDS banks per half-wave tap
    instruction (what SQ_LDS_BANK_CONFLICT measures).
Projections are the oracle's (oracle.project_points) on the synthetic 5-camera rig, heat-map pixels = image pixels / 8.

    python tools/sim_bwd_patch.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from selfpose3d_amd import synthetic as syn               # noqa: E402
from selfpose3d_amd.camera_pack import pack_cameras       # noqa: E402
from oracle import oracle                                 # noqa: E402


def tap_origins(num_cubes=2):
    img = (960, 512)
    cam = pack_cameras(syn.make_meta(2, 5, img), 2, img)
    rng = np.random.default_rng(0)
    c = np.stack([rng.uniform(-1500, 1500, 4), rng.uniform(-2000, 1000, 4), rng.uniform(700, 1100, 4)], 1).astype(np.float32)
    ax = [oracle.linspace(syn.FINE_GRID_SIZE[i], syn.FINE_CUBE_SIZE[i]) for i in range(3)]
    out = []
    for p in range(num_cubes):
        X, Y, Z = np.meshgrid(ax[0] + c[p, 0], ax[1] + c[p, 1], ax[2] + c[p, 2], indexing="ij")
        pts = np.stack([X, Y, Z], -1).reshape(-1, 3)
        for v in range(5):
            uv = oracle.project_points(cam[0, v], pts).reshape(64, 64, 64, 2) / 8.0
            out.append((np.floor(uv[..., 0]).astype(int), np.floor(uv[..., 1]).astype(int)))
    return out


def rectangles(origins):
    print("block shape: rectangle pixels mean / p50 / p90 / p99 | distinct pixels of taps (merge factor) | share above 256 / 384 / 512 px")
    for shape in [(8, 8, 4), (4, 4, 16), (4, 8, 8), (8, 8, 8), (16, 16, 4)]:
        rows = []
        bx, by, bz = shape
        for x0, y0 in origins:
            for i in range(0, 64, bx):
                for j in range(0, 64, by):
                    for k in range(0, 64, bz):
                        xs, ys = x0[i:i + bx, j:j + by, k:k + bz], y0[i:i + bx, j:j + by, k:k + bz]
                        ok = (xs >= 0) & (xs < 239) & (ys >= 0) & (ys < 127)
                        if not ok.any():
                            continue
                        pw, ph = xs[ok].max() - xs[ok].min() + 2, ys[ok].max() - ys[ok].min() + 2
                        key = ys[ok] * 1000 + xs[ok]
                        d = set()
                        for q in key.ravel():
                            d.update((q, q + 1, q + 1000, q + 1001))
                        rows.append((pw * ph, len(d), ok.sum() * 4))
        r = np.array(rows)
        print(f"{shape}: {r[:, 0].mean():.0f} / {np.percentile(r[:, 0], 50):.0f} / {np.percentile(r[:, 0], 90):.0f} / "
              f"{np.percentile(r[:, 0], 99):.0f} | {r[:, 1].mean():.0f} of {r[:, 2].mean():.0f} (x{r[:, 2].sum() / r[:, 1].sum():.1f}) | "
              f"{(r[:, 0] > 256).mean():.3f} / {(r[:, 0] > 384).mean():.3f} / {(r[:, 0] > 512).mean():.3f}")


def bank_load(origins):
    t = np.arange(256)
    maps = {"z fastest (shipped)": (t >> 5, (t >> 2) & 7, t & 3), "z slowest (a wave = a layer)": ((t >> 3) & 7, t & 7, t >> 6),
            "y fastest": (t >> 5, t & 7, (t >> 3) & 3)}
    strides = {"width": lambda w: w, "width | 1": lambda w: w | 1, "== 5 mod 8": lambda w: w + ((5 - w) % 8),
               "== 9 mod 16": lambda w: w + ((9 - w) % 16)}
    print("lane map, row stride: lanes on the busiest of 32 double-word banks per half-wave tap instruction (mean)")
    for name, (mx, my, mz) in maps.items():
        acc = {k: [] for k in strides}
        for x0, y0 in origins[:5]:
            for bx in range(0, 64, 8):
                for by in range(0, 64, 8):
                    for bz in range(0, 64, 16):
                        xs, ys = x0[bx + mx, by + my, bz + mz], y0[bx + mx, by + my, bz + mz]
                        ok = (xs >= 0) & (xs < 239) & (ys >= 0) & (ys < 127)
                        if ok.sum() < 200:
                            continue
                        rx0, ry0, ww = xs[ok].min(), ys[ok].min(), xs[ok].max() - xs[ok].min() + 2
                        for pol, f in strides.items():
                            S, tot = f(ww), 0
                            for tap in range(4):
                                pidx = (ys + (tap >> 1) - ry0) * S + (xs + (tap & 1) - rx0)
                                for h in range(8):
                                    p, o = pidx[h * 32:(h + 1) * 32], ok[h * 32:(h + 1) * 32]
                                    if o.any():
                                        tot += np.bincount(p[o] % 32, minlength=32).max()
                            acc[pol].append(tot / 32.0)
        print("  %-30s " % name + "   ".join(f"{k}: {np.mean(v):.2f}" for k, v in acc.items()))


if __name__ == "__main__":
    o = tap_origins()
    rectangles(o)
    bank_load(o)
