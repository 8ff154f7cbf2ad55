#!/bin/bash
# Round-3 profile set (GPU box): bench line (+ legs), kernel trace of the bench command, PMC traffic of the unprojection
# kernel the step runs, instruction-issue counters of the unprojection kernels on three grids, A/B of all variants, NMS.
#   bash tools/profile_round3.sh      -> gpurun_out/r03_*   (copy what is judged into profiles/)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r03
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. bench line + kernel trace of the same command
python $R/bench.py --steps 100 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
rm -rf $O/${TAG}_trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o bench -- \
    python $R/bench.py --steps 20 --no-cpu-baseline --no-fp32-leg --no-cold --legs none > $O/${TAG}_trace_bench.json 2> $O/${TAG}_trace.err
python $R/tools/rocpd_stats.py $O/${TAG}_trace > $O/${TAG}_bench_kernel_stats.md 2>> $O/${TAG}_trace.err
find $O/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
rm -rf $O/${TAG}_trace
cd $R
# 2. HBM-side traffic of the kernel the step runs (brick, channels-last), planar-result kernel, dense grids
bash tools/collect_pmc.sh $O/${TAG}_pmc_coarse_b4_cl coarse_b4_v5 -1 --cl > /dev/null 2>&1
python tools/pmc_traffic.py $O/${TAG}_pmc_coarse_b4_cl/summary.json B4_V5_J15_240x128_80x80x20 > $O/pmc_traffic.json
bash tools/collect_pmc.sh $O/${TAG}_pmc_coarse_b4 coarse_b4_v5 -1 > /dev/null 2>&1
python tools/pmc_traffic.py $O/${TAG}_pmc_coarse_b4/summary.json B4_V5_J15_240x128_80x80x20 > $O/pmc_traffic_planar.json
bash tools/collect_pmc.sh $O/${TAG}_pmc_stress_v10_cl stress_b1_v10 -1 --cl > /dev/null 2>&1
bash tools/collect_pmc.sh $O/${TAG}_pmc_fine64_cl fine_b10_v5 -1 --cl > /dev/null 2>&1
for d in coarse_b4 coarse_b4_cl stress_v10_cl fine64_cl; do rm -rf $O/${TAG}_pmc_$d/pass*/; done
# 3. instruction-issue counters (SQ_INSTS_* by class, LDS, TA) of the brick and wave-private patch kernels
for w in coarse_b4_v5 stress_b1_v10 fine_b10_v5; do
  bash tools/pmc_sq_lds.sh $O/${TAG}_issue_${w}_brick $w 120 --cl > /dev/null 2>&1
  bash tools/pmc_sq_lds.sh $O/${TAG}_issue_${w}_wpatch $w 128 --cl > /dev/null 2>&1
  rm -rf $O/${TAG}_issue_${w}_brick/pass*/ $O/${TAG}_issue_${w}_wpatch/pass*/
done
# 4. A/B of the unprojection kernels, heat-map footprint sweep, NMS kernels
python tools/ab_variants.py --rounds 5 --iters 100 --variants 24,56,120,376,128 > $O/${TAG}_ab_variants.json 2> /dev/null
python tools/exp_hmsize.py > $O/${TAG}_exp_hmsize.json 2> /dev/null
bash tools/kstats.sh "nms" tools/bench_nms.py > $O/${TAG}_nms_kernels.txt 2>&1
python tools/bench_bwd.py > $O/${TAG}_backward_kernels.json 2> /dev/null
ls $O | grep ${TAG}
