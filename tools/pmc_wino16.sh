#!/bin/bash
# instruction counts and matrix-pipe busy cycles of wino_fused16_kernel<64, MODE, 2, 2> (one rocprofv3 --pmc run per group, no
# trace flags): bash tools/pmc_wino16.sh <tag>  ->  gpurun_out/<tag>_pmc_wino16.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r05}
OUT=$R/gpurun_out/${TAG}_pmc_wino16; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $group --output-format csv -d "$OUT/pass$i" -o pmc -- python "$R/tools/run_wino16.py" --iters 12 > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($group) rc=$?"
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
GRBM_GUI_ACTIVE GRBM_COUNT
GROUPS
python "$R/tools/pmc_summary.py" "$OUT" > "$R/gpurun_out/${TAG}_pmc_wino16.json"
rm -rf $OUT/pass*
head -c 3000 "$R/gpurun_out/${TAG}_pmc_wino16.json"
