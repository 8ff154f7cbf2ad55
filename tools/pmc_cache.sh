#!/bin/bash
# Cache-path counters only (L1 accesses / L1->L2 requests / L2 hits, misses, fabric reads) for one kernel variant:
# 3 rocprofv3 passes, no trace flags combined with --pmc.
#   bash tools/pmc_cache.sh <outdir> <workload> <variant> [extra run_kernel.py args]
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
      python "$R/tools/run_kernel.py" --workload "$WL" --variant "$VAR" --iters 12 "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($group) rc=$?"
done <<'GROUPS'
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
GROUPS
python "$R/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
rm -rf "$OUT"/pass*/
