#!/bin/bash
# SQ / LDS / TA counters of one unprojection kernel variant (4 rocprofv3 --pmc passes, no trace flags).
#   bash tools/pmc_sq_lds.sh <outdir> <workload> <variant> [--cl]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; WL=$2; VAR=$3; shift 3
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $group --output-format csv -d "$OUT/pass$i" -o pmc -- \
      python "$R/tools/run_kernel.py" --workload "$WL" --variant "$VAR" --iters 8 "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($group) rc=$?"
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM
TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
GRBM_GUI_ACTIVE
GROUPS
python "$R/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
