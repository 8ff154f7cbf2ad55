#!/bin/bash
# SQ / LDS / L2-atomic counters of the backward scatter kernels (separate rocprofv3 --pmc passes, no trace flags).
#   bash tools/pmc_bwd.sh <outdir> [fine_p4_b2|coarse_b4]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; WL=${2:-fine_p4_b2}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  BWD_ONLY=$WL timeout 300 rocprofv3 --pmc $group --output-format csv -d "$OUT/pass$i" -o pmc -- \
      python "$R/tools/bench_bwd.py" > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($group) rc=$?"
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN
TCC_ATOMIC_sum TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum
TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum
GRBM_GUI_ACTIVE
GROUPS
python "$R/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
python3 - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "bwd" in k:
        print(k[:110])
        for c, x in sorted(v.items()):
            print("   %-28s %14.1f  (n=%d)" % (c, x["mean"], x["n"]))
PY
