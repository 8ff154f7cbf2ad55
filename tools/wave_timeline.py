#!/usr/bin/env python3
"""Per-wave s_memtime timeline of the pipelined unprojection kernel (measurement only)."""
import ctypes, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from selfpose3d_amd import _lib, synthetic as syn
from selfpose3d_amd.camera_pack import pack_cameras
from selfpose3d_amd import build as _build
LEVEL = 1 if "--occupancy" in sys.argv else 2
TL = os.path.join(ROOT, "selfpose3d_amd", "libsp3d_timeline%d.so" % LEVEL)      # stamped build, separate from the shipped library
if "--build-only" in sys.argv or not os.path.exists(TL) or os.path.getmtime(TL) < os.path.getmtime(_build.LIB):
    _build.build_variant(TL, ["-DSP3D_TIMELINE=%d" % LEVEL])
    if "--build-only" in sys.argv:
        sys.exit(0)
_lib.LIB_PATH = TL
lib = _lib.load()
img, (w, h), J = (960, 512), (240, 128), 15
dev = torch.device("cuda:0")
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
B, V = int(_pos[0]) if _pos else 4, 5
cube, gs = syn.INITIAL_CUBE_SIZE, syn.SPACE_SIZE
meta = syn.make_meta(B, V, img)
cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
valid = torch.ones(B, dtype=torch.uint8, device=dev)
hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=7)]
packed = _lib.pack_heatmaps(hms, jp=16); views = [packed[c] for c in range(V)]
run = lambda: _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16, h, w, cube, gs, img, False, channels_last=True)
for _ in range(5): run()
nblk = 8 * 4096
S = 32
buf = torch.zeros(nblk * S, dtype=torch.int64, device=dev)
lib.sp3d_debug_set_timeline.argtypes = [ctypes.c_void_p]
assert lib.sp3d_debug_set_timeline(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
assert lib.sp3d_debug_set_timeline(None) == 0
t = buf.cpu().numpy().reshape(-1, S)
t = t[t[:, 0] != 0]
start, p1, end = t[:, 0], t[:, 1], t[:, 30]
life = end - start
span = end.max() - start.min()
us = e0.elapsed_time(e1) * 1e3
if LEVEL == 1:
    xcc = (t[:, 29] & 0xf).astype(int)
    hw = t[:, 28].astype(np.int64)
    cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7)      # cu_id, se_id, sh_id
    out = {"waves": int(len(t)), "kernel_us_event": round(us, 1), "xcds": {}}
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        s0, e0_ = t[m, 26].astype(np.float64) * 10.0, t[m, 27].astype(np.float64) * 10.0      # ns
        t0, t1 = s0.min(), e0_.max()
        grid = np.linspace(t0, t1, 21)
        live = [int(((s0 <= g) & (e0_ > g)).sum()) for g in grid[:-1]]
        ncu = len(set(cu[m].tolist()))
        out["xcds"][int(x)] = {"waves": int(m.sum()), "cus_seen": ncu, "span_us": round((t1 - t0) / 1e3, 2), "first_start_us": round((t0 - t[:, 26].min() * 10.0) / 1e3, 2),
                               "wave_life_us": round(float((e0_ - s0).mean()) / 1e3, 2),
                               "last_start_frac": round(float((s0.max() - t0) / (t1 - t0)), 3),
                               "live_waves_per_cu_at_5pct_steps": [round(v / max(ncu, 1), 1) for v in live]}
        st_ = np.sort(s0); en_ = np.sort(e0_); tg = t[:, 26].min() * 10.0
        out["xcds"][int(x)]["epilogue_us_mean"] = round(float((t[m, 27] - t[m, 25]).mean()) * 10.0 / 1e3, 2)
        out["xcds"][int(x)]["first_end_us"] = round(float(en_[0] - tg) / 1e3, 2)
        out["xcds"][int(x)]["end_q10_us"] = round(float(en_[len(en_) // 10] - tg) / 1e3, 2)
        out["xcds"][int(x)]["gen2_first_start_us"] = round(float(st_[min(512, len(st_) - 1)] - tg) / 1e3, 2)
        if x == 0:      # per-CU / per-SE spread inside one XCD at the same instants
            cus = sorted(set(cu[m].tolist())); cum = cu[m]; se = (cum >> 4) & 7
            rows = []
            for g in grid[1:-1:2]:
                alive = (s0 <= g) & (e0_ > g)
                per = np.array([int((alive & (cum == c_)).sum()) for c_ in cus])
                per_se = [round(float(alive[se == k].sum()) / max(1, len(set(cum[se == k].tolist()))), 1) for k in sorted(set(se.tolist()))]
                rows.append({"t_frac": round(float((g - t0) / (t1 - t0)), 2), "cu_min": int(per.min()), "cu_mean": round(float(per.mean()), 1),
                             "cu_max": int(per.max()), "per_se": per_se})
            out["xcd0_per_cu"] = rows
            st = np.sort(s0 - t0) / 1e3
            out["xcd0_start_us_quantiles"] = [round(float(st[int(q * (len(st) - 1))]), 2) for q in (0, .1, .25, .5, .51, .55, .6, .7, .8, .9, 1.0)]
    print(json.dumps(out, indent=1))
    sys.exit(0)
# stamps per view c: [2+4c] view start, [3+4c] tap loads issued, [4+4c] P1(c+1) done; next view start = FMAs done
nxt = lambda c: t[:, 2 + 4 * (c + 1)] if c + 1 < V else end
res = {"waves": int(len(t)), "kernel_us_event": round(us, 1), "kernel_span_ticks": int(span), "ticks_per_us": round(span / us, 1),
       "wave_life_ticks": {"mean": float(life.mean()), "p10": float(np.percentile(life, 10)), "p90": float(np.percentile(life, 90))},
       "avg_resident_waves_per_cu": round(float(life.sum() / span / 256), 2),
       "P1_0_ticks_mean": float((p1 - start).mean()),
       "issue_loads_ticks_mean": [float((t[:, 3 + 4 * c] - t[:, 2 + 4 * c]).mean()) for c in range(V)],
       "P1_next_ticks_mean": [float((t[:, 4 + 4 * c] - t[:, 3 + 4 * c]).mean()) for c in range(V)],
       "wait_fma_ticks_mean": [float((nxt(c) - t[:, 4 + 4 * c]).mean()) for c in range(V)],
       "start_spread_ticks": {"p50": float(np.percentile(start - start.min(), 50)), "p90": float(np.percentile(start - start.min(), 90)), "max": float((start - start.min()).max())},
       "life_by_bound_views": {int(k): float(life[t[:, 31] == k].mean()) for k in np.unique(t[:, 31])}}
# phase concurrency per CU (round 4): how many of a CU's resident waves are in the SAME phase at the same instant?
# (cycle counters are per XCD: compare waves of one XCD only).  issue = [view start, tap loads issued), p1 = [loads issued,
# next view projected), fma = [projected, next view start)
xcc = (t[:, 29] & 0xf).astype(int)
hw = t[:, 28].astype(np.int64)
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7)
m = xcc == 0
tt, cux = t[m].astype(np.float64), cu[m]
# the cycle counters are not aligned between CUs: place every wave on the chip-wide 100 MHz clock (stamp 26 = wave start)
# and measure inside the wave with its own cycle counter
fclk = float(np.median((tt[:, 30] - tt[:, 0]) / np.maximum(1.0, (tt[:, 25] - tt[:, 26]))))      # cycles per 10 ns
base = tt[:, 26] * fclk - tt[:, 0]
for col in list(range(0, 25)) + [30]:
    tt[:, col] = np.where(tt[:, col] > 0, tt[:, col] + base, 0.0)
res["cycles_per_us"] = round(fclk * 100.0, 1)
iv = {"issue": [], "p1": [], "fma": []}
for c in range(V):
    s0, s1, s2 = tt[:, 2 + 4 * c], tt[:, 3 + 4 * c], tt[:, 4 + 4 * c]
    s3 = tt[:, 2 + 4 * (c + 1)] if c + 1 < V else tt[:, 30]
    ok = (s0 > 0) & (s1 >= s0) & (s2 >= s1) & (s3 >= s2)
    for k, (a, b) in (("issue", (s0, s1)), ("p1", (s1, s2)), ("fma", (s2, s3))):
        iv[k].append(np.stack([a[ok], b[ok], cux[ok]], 1))
iv = {k: np.concatenate(v) for k, v in iv.items()}
t0, t1 = tt[:, 0].min(), tt[:, 30].max()
grid = np.linspace(t0 + 0.1 * (t1 - t0), t0 + 0.6 * (t1 - t0), 400)         # the busy middle of the launch
conc = {}
for k, v in iv.items():
    counts = []
    for c_ in sorted(set(cux.tolist()))[:32]:
        r = v[v[:, 2] == c_]
        counts.append([int(((r[:, 0] <= g_) & (r[:, 1] > g_)).sum()) for g_ in grid])
    counts = np.array(counts)
    conc[k] = {"mean_waves_in_phase_per_cu": round(float(counts.mean()), 2), "p10": float(np.percentile(counts, 10)),
               "p50": float(np.percentile(counts, 50)), "p90": float(np.percentile(counts, 90)), "max": int(counts.max()),
               "mean_phase_ticks": round(float((v[:, 1] - v[:, 0]).mean()), 1)}
live = np.array([[int(((tt[cux == c_, 0] <= g_) & (tt[cux == c_, 30] > g_)).sum()) for g_ in grid] for c_ in sorted(set(cux.tolist()))[:32]])
res["phase_concurrency_xcd0"] = conc
res["resident_waves_per_cu_mean_in_window"] = round(float(live.mean()), 2)
print(json.dumps(res, indent=1))
