#!/bin/bash
# timing + one rocprofv3 --pmc pass per variant (L1 accesses, L1 -> L2 read requests, wave-loads) -> gpurun_out/r05_l1_residency.{json,md}
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05_l1; mkdir -p $OUT
python $R/tools/l1_residency_sweep.py > $OUT/timing.json 2> $OUT/timing.err
cd /tmp && export TMPDIR=/tmp
i=0
python $R/tools/l1_residency_sweep.py --list | while read -r name; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum --output-format csv -d $OUT/v$i -o pmc -- \
      python $R/tools/l1_residency_sweep.py --only "$name" > $OUT/v$i.log 2>&1
  echo "$name" > $OUT/v$i.name
done
python - <<PY
import csv, glob, json, os
out = "$OUT"
timing = json.load(open(os.path.join(out, "timing.json")))
rows = []
for nf in sorted(glob.glob(os.path.join(out, "v*.name")), key=lambda p: int(os.path.basename(p)[1:-5])):
    name = open(nf).read().strip()
    d = nf[:-5]
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "unproject_brick_kernel" in row.get("Kernel_Name", ""):
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    t = timing.get(name, {})
    rec = {"variant": name, **t, **{k: round(v) for k, v in m.items()}}
    if m.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        rec["l1_hit_rate"] = round(1.0 - m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
    rows.append(rec)
json.dump(rows, open(os.path.join("$R", "gpurun_out", "r05_l1_residency.json"), "w"), indent=1)
with open(os.path.join("$R", "gpurun_out", "r05_l1_residency.md"), "w") as f:
    f.write("| variant | us | bit-identical | L1 accesses | L1->L2 read requests | L1 hit rate | wave-loads |\n|---|---:|---|---:|---:|---:|---:|\n")
    for r in rows:
        f.write("| %s | %s | %s | %s | %s | %s | %s |\n" % (r["variant"], r.get("us"), r.get("bit_identical"), r.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
                r.get("TCP_TCC_READ_REQ_sum"), r.get("l1_hit_rate"), r.get("TA_FLAT_READ_WAVEFRONTS_sum")))
print(open(os.path.join("$R", "gpurun_out", "r05_l1_residency.md")).read())
PY
