#!/bin/bash
# Round profile set (GPU box): kernel trace of the bench command + PMC passes of the kernels DESIGN.md quotes.
#   bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*   (copy what is judged into profiles/)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. bench line + kernel trace of the same command
python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
rm -rf $O/${TAG}_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o bench -- \
    python $R/bench.py --no-cpu-baseline --no-cold --no-fp32-leg > $O/${TAG}_trace_bench.json 2> $O/${TAG}_trace.err   # the bench command minus its CPU leg and its rotating-input leg: the kernel average is then the warm one bench.py reports
python $R/tools/rocpd_stats.py $O/${TAG}_trace > $O/${TAG}_bench_kernel_stats.md 2>> $O/${TAG}_trace.err
find $O/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
rm -rf $O/${TAG}_trace
# 2. PMC: unprojection kernels on the bench workload (planar default), stress and fine configs
cd $R
bash tools/collect_pmc.sh $O/${TAG}_pmc_coarse_b4 coarse_b4_v5 -1 > /dev/null 2>&1
bash tools/collect_pmc.sh $O/${TAG}_pmc_coarse_b4_cl coarse_b4_v5 -1 --cl > /dev/null 2>&1
python tools/pmc_traffic.py $O/${TAG}_pmc_coarse_b4/summary.json B4_V5_J15_240x128_80x80x20 > $O/pmc_traffic_planar.json
python tools/pmc_traffic.py $O/${TAG}_pmc_coarse_b4_cl/summary.json B4_V5_J15_240x128_80x80x20 > $O/pmc_traffic.json
bash tools/collect_pmc.sh $O/${TAG}_pmc_stress_v10 stress_b1_v10 -1 > /dev/null 2>&1
bash tools/collect_pmc.sh $O/${TAG}_pmc_stress_v10_cl stress_b1_v10 -1 --cl > /dev/null 2>&1
bash tools/collect_pmc.sh $O/${TAG}_pmc_fine64 fine_b10_v5 -1 > /dev/null 2>&1
bash tools/collect_pmc.sh $O/${TAG}_pmc_fine64_v24 fine_b10_v5 24 > /dev/null 2>&1
for d in coarse_b4 coarse_b4_cl stress_v10 stress_v10_cl fine64 fine64_v24; do rm -rf $O/${TAG}_pmc_$d/pass*/; done
# 3. PMC: fused Winograd kernel
for kind in fp32 split half direct; do
  bash tools/pmc_wino_fused.sh $kind > $O/${TAG}_pmc_wino_fused_$kind.txt 2>&1
  cp $O/pmc_wf/summary.json $O/${TAG}_pmc_wino_fused_$kind.json 2>/dev/null
  rm -rf $O/pmc_wf
done
ls $O | grep ${TAG}
