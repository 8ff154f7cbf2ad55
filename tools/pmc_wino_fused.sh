KIND=${1:-fp32}; R=$PWD; OUT=$R/gpurun_out/pmc_wf; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $group --output-format csv -d $OUT/pass$i -o pmc -- python $R/tools/run_wino_fused.py 4 $KIND > $OUT/pass$i.log 2>&1
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY
GRBM_GUI_ACTIVE GRBM_COUNT
GROUPS
cd $R; python tools/pmc_summary.py $OUT > $OUT/summary.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d.items():
    if "wino_fused" in k or "conv3_split" in k:
        for c,x in sorted(v.items()): print("%-32s %16.1f"%(c,x["mean"]))
PY
