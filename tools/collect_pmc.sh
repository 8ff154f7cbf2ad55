#!/bin/bash
# Collect hardware counters for the unprojection kernel, one rocprofv3 run per counter group
# (no trace flags are combined with --pmc).  Usage on the GPU box:
#   bash tools/collect_pmc.sh <outdir> <workload> <variant> [extra run_kernel.py args]
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
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE GRBM_COUNT
TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
GROUPS
python "$R/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
