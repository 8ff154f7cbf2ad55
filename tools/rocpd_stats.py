#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace (CSV or rocpd .db) into per-kernel stats + a steady-state step view.

    python tools/rocpd_stats.py <dir-or-file> [--step-end sp3d::nms_merge_kernel] > profiles/rNN_*.md
"""
import csv
import glob
import os
import sqlite3
import sys
from collections import OrderedDict


def load(path):
    rows = []
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    csvs = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    if csvs:
        for f in csvs:
            for r in csv.DictReader(open(f)):
                rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    elif dbs:
        con = sqlite3.connect(dbs[0])
        rows = con.execute("select name,start,end from kernels order by start").fetchall()
    rows.sort(key=lambda r: r[1])
    return rows


def table(rows, title, top=30):
    agg = OrderedDict()
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1.0
    print(f"\n### {title}\n")
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"| `{n[:110]}` | {a[0]} | {a[1]:.1f} | {a[1]/a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100*a[1]/tot:.1f} |")


def main():
    path = sys.argv[1]
    step_end = "sp3d::nms_merge_kernel"
    if "--step-end" in sys.argv:
        step_end = sys.argv[sys.argv.index("--step-end") + 1]
    rows = load(path)
    print(f"# rocprofv3 kernel trace summary ({os.path.basename(path.rstrip('/'))}): {len(rows)} dispatches")
    table([r for r in rows if "sp3d" in r[0]], "sp3d kernels (all dispatches)")
    # The unprojection kernel is launched in two contexts: inside the step (behind the previous step's V2V: cold L2 /
    # Infinity Cache) and back to back in bench.py's roofline leg (the `roofline.kernel_us` of the bench line).
    b2b = [rows[i] for i in range(1, len(rows)) if "unproject" in rows[i][0] and rows[i - 1][0] == rows[i][0]]
    ins = [rows[i] for i in range(1, len(rows)) if "unproject" in rows[i][0] and rows[i - 1][0] != rows[i][0]]
    if b2b:
        table(b2b, "unprojection kernels, back-to-back launches only (bench.py roofline leg: warm inputs)")
    if ins:
        table(ins, "unprojection kernels, launches behind another kernel (inside the step: cold caches)")
    ends = [i for i, r in enumerate(rows) if r[0].startswith(step_end)]
    if len(ends) >= 3:
        a, b = ends[-2], ends[-1]
        step = rows[a + 1:b + 1]
        span = (step[-1][2] - step[0][1]) / 1e3
        busy = sum(e - s for _, s, e in step) / 1e3
        table(step, f"last steady-state step: {len(step)} kernels, span {span:.1f} us, GPU-busy {busy:.1f} us")
        if "--sequence" in sys.argv:
            print("\n### last steady-state step in launch order\n")
            print("| # | start us | dur us | gap before us | kernel |")
            print("|---:|---:|---:|---:|---|")
            t0, prev = step[0][1], None
            for i, (n, s_, e_) in enumerate(step):
                gap = 0.0 if prev is None else (s_ - prev) / 1e3
                print(f"| {i} | {(s_ - t0) / 1e3:.1f} | {(e_ - s_) / 1e3:.1f} | {gap:.1f} | `{n[:90]}` |")
                prev = e_
    table(rows, "all kernels (includes MIOpen find-mode warm-up)", top=15)


if __name__ == "__main__":
    main()
