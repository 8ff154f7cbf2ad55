#!/bin/bash
# round 6: L2-fill traffic of the brick kernel under the two brick -> XCD maps, channels-last result, per workload:
#   variant 120            = the library default since round 6: one block of brick columns per XCD (octants at B = 1,
#                            quadrants at B = 2, halves at B = 4)
#   variant 120 | 1 << 22  = round 5's round-robin chunks of consecutive bricks (4194424)
#   bash tools/pmc_blocks.sh  ->  gpurun_out/r06_pmc_blocks.json
# separate rocprofv3 --pmc passes, no trace flags (collect_pmc.sh's rule); FETCH_SIZE x2 per the gfx950 note of the guide
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06_pmc_blocks; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for WL in stress_b1_v10 coarse_b1_v5 coarse_b4_v5; do
  for VAR in 120 4194424; do
    i=0
    for group in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
      i=$((i+1))
      timeout 300 rocprofv3 --pmc $group --output-format csv -d "$OUT/${WL}_$VAR/pass$i" -o pmc -- \
        python "$R/tools/run_kernel.py" --workload "$WL" --variant "$VAR" --iters 12 --cl > "$OUT/${WL}_${VAR}_pass$i.log" 2>&1
    done
    python "$R/tools/pmc_summary.py" "$OUT/${WL}_$VAR" > "$OUT/${WL}_$VAR.summary.json"
  done
done
python - "$OUT" <<'PY' > $R/gpurun_out/r06_pmc_blocks.json
import json, sys, glob, os
out = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.summary.json"))):
    s = json.load(open(f))
    k = [n for n in s if "unproject" in n][0]
    d = {c: v["mean"] for c, v in s[k].items()}
    name = os.path.basename(f).replace(".summary.json", "")
    out[name] = {"kernel": k[:60], "read_MB": round(2 * d["FETCH_SIZE"] * 1024 / 1e6, 1),
                 "read_MB_from_rdreq": round((d.get("TCC_EA0_RDREQ_128B_sum", 0) * 128 + d.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + d.get("TCC_EA0_RDREQ_32B_sum", 0) * 32) / 1e6, 1),
                 "read_MB_dram_flagged": round(d.get("TCC_EA0_RDREQ_DRAM_sum", 0) * 128 / 1e6, 1),
                 "write_MB": round(d["WRITE_SIZE"] * 1024 / 1e6, 1), "l2_hit": d.get("TCC_HIT_sum"), "l2_miss": d.get("TCC_MISS_sum"),
                 "gui_active_cycles": d.get("GRBM_GUI_ACTIVE")}
print(json.dumps(out, indent=1))
PY
cat $R/gpurun_out/r06_pmc_blocks.json
rm -rf $OUT/*/pass*
