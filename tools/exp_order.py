#!/usr/bin/env python3
"""Does the dispatch ORDER of the bricks matter?  (measurement only)  Shared-rig path: per-brick view counts are read back
from the K1 records, bricks are ordered heaviest-first (LPT) / lightest-first / shuffled, K2 is timed for each order."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from selfpose3d_amd import _lib, synthetic as syn
from selfpose3d_amd.camera_pack import pack_cameras


def timed(fn, iters=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return round(best, 2)


if __name__ != "__main__":
    raise SystemExit
dev = torch.device("cuda:0")
img, (w, h), J = (960, 512), (240, 128), 15
out = {}
for name, B, V, cube in (("coarse_b4", 4, 5, syn.INITIAL_CUBE_SIZE), ("coarse_b1", 1, 5, syn.INITIAL_CUBE_SIZE), ("stress_b1", 1, 10, (160, 160, 40))):
    gs = syn.SPACE_SIZE
    meta = syn.make_meta(B, V, img)
    cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
    centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
    valid = torch.ones(B, dtype=torch.uint8, device=dev)
    hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=7)]
    packed = _lib.pack_heatmaps(hms, jp=16); views = [packed[c] for c in range(V)]
    rec = _lib.build_records(cam[0], centers[0], V, 16, h, w, cube, gs, img)
    nb = [(c + 3) // 4 for c in cube]
    per = V * 320 + 128
    hdr = rec.view(-1, per)[:, V * 320 + 64].contiguous().view(torch.int32).cpu().numpy()
    cnt = np.array([bin(int(x) & 0xffff).count("1") for x in hdr])          # views with taps per brick
    res = {"bricks": int(len(cnt)), "views_hist": np.bincount(cnt, minlength=V + 1).tolist()}
    for cl in (True, False):
        if cl:
            weight = cnt
        else:           # planar: workgroup = z-stack
            nwz = nb[2]; nzc = (nwz + 7) // 8; zw = (nwz + nzc - 1) // nzc
            c3 = cnt.reshape(nb[0] * nb[1], nwz)
            pad = nzc * zw - nwz
            if pad: c3 = np.concatenate([c3, np.zeros((c3.shape[0], pad), int)], 1)
            weight = c3.reshape(-1, nzc, zw).sum(2).reshape(-1)
        n = len(weight)
        orders = {"natural": None, "lpt": np.argsort(-weight, kind="stable"), "light_first": np.argsort(weight, kind="stable"),
                  "shuffled": np.random.default_rng(0).permutation(n)}
        # interleave: heavy and light alternate (keeps every CU mixed)
        lpt = np.argsort(-weight, kind="stable"); half = (n + 1) // 2
        mix = np.empty(n, int); mix[0::2] = lpt[:half]; mix[1::2] = lpt[half:][::-1]
        orders["heavy_light_pairs"] = mix
        row = {}
        for k, o in orders.items():
            od = None if o is None else torch.from_numpy(o.astype(np.int32)).to(dev)
            for chunk in ((0,) if o is None else (0, 1)):
                fn = lambda: _lib.unproject_fwd_records(views, 16, rec, valid, B, 16 if cl else J, h, w, cube, channels_last=cl,
                                                        order=od, xcd_chunk=chunk)
                row[f"{k}{'_chunk1' if chunk else ''}"] = timed(fn)
        for occ, tag in ((1, "U2_5waves"), (2, "U2_6waves"), (3, "U1_8waves")):
            fn = lambda: _lib.unproject_fwd_records(views, 16, rec, valid, B, 16 if cl else J, h, w, cube, channels_last=cl,
                                                    xcd_chunk=occ << 16)
            row["natural_" + tag] = timed(fn)
        if cl:
            for pers in (2, 3, 4, 5, 6):
                for u in (0, 1):
                    fn = lambda: _lib.unproject_fwd_records(views, 16, rec, valid, B, 16, h, w, cube, channels_last=True,
                                                            xcd_chunk=(pers << 18) | (u << 16))
                    row[f"persist_{4 * pers}wavesPerCU_{'U2' if u else 'U4'}"] = timed(fn)
        res["cl" if cl else "planar"] = row
    out[name] = res
print(json.dumps(out, indent=1))
