#!/usr/bin/env python3
"""aggregate the LAST `--ms` milliseconds of a rocprofv3 kernel trace (csv) by kernel name: steady-state steps of a run whose
beginning is MIOpen find-mode noise.   python tools/trace_tail.py <dir> --ms 400 [--top 40]"""
import argparse, csv, glob, os, collections
ap = argparse.ArgumentParser()
ap.add_argument("dir"); ap.add_argument("--ms", type=float, default=400.0); ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()
f = sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
end = max(r[1] for r in rows)
t0 = end - int(a.ms * 1e6)
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for s, e, n in rows:
    if s >= t0:
        agg[n][0] += 1; agg[n][1] += e - s; busy += e - s
print(f"# last {a.ms:.0f} ms of the trace: GPU-busy {busy / 1e6:.1f} ms in {sum(v[0] for v in agg.values())} dispatches\n")
print("| kernel | calls | total ms | avg us | % of busy |\n|---|---:|---:|---:|---:|")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"| `{n[:110]}` | {c} | {t / 1e6:.2f} | {t / c / 1e3:.1f} | {100.0 * t / busy:.1f} |")
