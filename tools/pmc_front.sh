#!/bin/bash
# counters of the opening conv's kernels (cfft2d_88, freq_contract, zdft_inv_cl) as tools/bench_fused_zdft.py launches them
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06_pmc_front; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $group --output-format csv -d "$OUT/pass$i" -o pmc -- python "$R/tools/bench_fused_zdft.py" > "$OUT/pass$i.log" 2>&1
done
python "$R/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
python - "$OUT/summary.json" <<'PY'
import json, sys
s = json.load(open(sys.argv[1]))
for k, v in s.items():
    if any(t in k for t in ("cfft2d", "freq_contract", "zdft")):
        print(k[:70], {c: round(x["mean"], 1) for c, x in v.items()})
PY
rm -rf $OUT/pass*
