#!/usr/bin/env python3
"""Budget of the graphed headline step from ONE rocprofv3 kernel trace (round-5 review, item 4):

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/sb -o sb -- python $R/bench.py --legs none \
        --no-cpu-baseline --no-fp32-leg --no-cold --roofline-iters 5
    python tools/step_budget.py /tmp/sb [ms_per_step of that run] > profiles/r06_step_budget.md

A step = the dispatches from sp3d::fetch_ring_kernel to sp3d::nms_merge_kernel.  Steps with the modal dispatch sequence
(no stamp kernels: the headline graph) are averaged position by position; every position is assigned to a tier by kernel
name.  Output: the sequence with average duration and average gap before each kernel, sums per tier, launch count, and
wall - sum(kernels) = what the dependent kernel boundaries cost."""
import csv
import glob
import os
import sys
from collections import Counter, OrderedDict

TIERS = [("unproject", "unprojection (+ re-tiling)"), ("pack_nhwc", "unprojection (+ re-tiling)"),
         ("zdft", "opening 7^3 conv, frequency domain"), ("cfft2d", "opening 7^3 conv, frequency domain"),
         ("freq_contract", "opening 7^3 conv, frequency domain"),
         ("conv3_split", "full resolution 3^3 (80x80x20)"), ("wino_fused16", "half resolution 3^3 (40x40x10)"),
         ("wino_fused_", "full resolution 3^3 (80x80x20)"),
         ("wino_input", "quarter resolution 3^3 (20x20x5)"), ("wino_output", "quarter resolution 3^3 (20x20x5)"),
         ("wino_gemm", "quarter resolution 3^3 (20x20x5)"),
         ("Cijk", "library GEMMs (quarter resolution products, 1^3 convs, up-convs)"),
         ("maxpool", "pooling / up-convolution / 1^3"), ("upsample", "pooling / up-convolution / 1^3"),
         ("channel_shift", "pooling / up-convolution / 1^3"), ("nms", "NMS + top-k"), ("fetch_ring", "camera-table fetch")]


def tier(name):
    for key, t in TIERS:
        if key in name:
            return t
    return "other (" + name.split("(")[0][:40] + ")"


def load(path):
    rows = []
    for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r[1])
    return rows


def main():
    rows = load(sys.argv[1])
    bench_ms = float(sys.argv[2]) if len(sys.argv) > 2 else None
    starts = [i for i, r in enumerate(rows) if "fetch_ring_kernel" in r[0]]
    steps = []
    for a in starts:
        for b in range(a + 1, min(a + 400, len(rows))):
            if "fetch_ring_kernel" in rows[b][0]:
                break
            if "nms_merge_kernel" in rows[b][0]:
                steps.append(rows[a:b + 1])
                break
    steps = [s for s in steps if not any("stamp_kernel" in r[0] for r in s)]
    seqs = Counter(tuple(r[0] for r in s) for s in steps)
    modal, _ = seqs.most_common(1)[0]
    same = [s for s in steps if tuple(r[0] for r in s) == modal]
    same = same[len(same) // 4:]                       # steady state: drop the first quarter (warm-up replays)
    n = len(modal)
    dur = [sum((s[i][2] - s[i][1]) for s in same) / len(same) / 1e3 for i in range(n)]
    gap = [0.0] + [sum((s[i][1] - s[i - 1][2]) for s in same) / len(same) / 1e3 for i in range(1, n)]
    wall = sum((s[-1][2] - s[0][1]) for s in same) / len(same) / 1e3
    # step-to-step distance (start of fetch to start of the next fetch) where the next step follows directly
    idx = {id(s): k for k, s in enumerate(steps)}
    d2d = []
    for k in range(len(steps) - 1):
        if tuple(r[0] for r in steps[k]) == modal and tuple(r[0] for r in steps[k + 1]) == modal:
            d = (steps[k + 1][0][1] - steps[k][0][1]) / 1e3
            if d < 3 * wall:
                d2d.append(d)
    d2d = d2d[len(d2d) // 4:]
    print(f"# Step budget of the graphed headline step (BASELINE configs[1], B=4): {len(same)} steady-state replays averaged\n")
    print(f"* dispatches per step: **{n}**")
    print(f"* sum of kernel durations: **{sum(dur):.1f} us**")
    print(f"* first kernel start -> last kernel end: **{wall:.1f} us**; gaps between consecutive kernels: **{sum(gap):.1f} us** "
          f"({sum(gap) / (n - 1):.2f} us per boundary)")
    if d2d:
        print(f"* step start -> next step start (back-to-back replays): **{sum(d2d) / len(d2d):.1f} us** "
              f"(= wall + {sum(d2d) / len(d2d) - wall:.1f} us between graph launches)")
    if bench_ms:
        print(f"* bench.py ms_per_step of the same run (no profiler overhead subtracted): {bench_ms * 1e3:.1f} us")
    print("\n## Per tier\n\n| tier | launches | kernel us | gaps before its kernels us | % of kernel time |")
    print("|---|---:|---:|---:|---:|")
    agg = OrderedDict()
    for i, name in enumerate(modal):
        a = agg.setdefault(tier(name), [0, 0.0, 0.0])
        a[0] += 1; a[1] += dur[i]; a[2] += gap[i]
    for t, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {t} | {a[0]} | {a[1]:.1f} | {a[2]:.1f} | {100 * a[1] / sum(dur):.1f} |")
    print(f"| **total** | **{n}** | **{sum(dur):.1f}** | **{sum(gap):.1f}** | 100 |")
    print("\n## In launch order\n\n| # | tier | kernel | avg us | gap before us |")
    print("|---:|---|---|---:|---:|")
    for i, name in enumerate(modal):
        print(f"| {i} | {tier(name)[:28]} | `{name[:100]}` | {dur[i]:.1f} | {gap[i]:.2f} |")


if __name__ == "__main__":
    main()
