#!/bin/bash
# round 5 (review item 4, tail): PMC traffic of the kernels that had timing only - configs[3] (10 views, 160x160x40, fp32)
# and configs[4] (ten 64^3 cubes, 4 views, bf16 maps and cubes: unproject_brick_h_kernel) -> gpurun_out/r05_pmc_*.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash $R/tools/collect_pmc.sh $R/gpurun_out/r05_pmc_stress stress_b1_v10 -1 --cl > /dev/null 2>&1
python $R/tools/pmc_traffic.py $R/gpurun_out/r05_pmc_stress/summary.json B1_V10_J15_240x128_160x160x40 > $R/gpurun_out/r05_pmc_configs3_stress_v10.json
bash $R/tools/collect_pmc.sh $R/gpurun_out/r05_pmc_bf16 fine_b10_v4 -1 --cl --bf16 > /dev/null 2>&1
python $R/tools/pmc_traffic.py $R/gpurun_out/r05_pmc_bf16/summary.json B10_V4_J15_240x128_64x64x64_bf16 > $R/gpurun_out/r05_pmc_configs4_bf16_v4.json
head -12 $R/gpurun_out/r05_pmc_configs3_stress_v10.json $R/gpurun_out/r05_pmc_configs4_bf16_v4.json
rm -rf $R/gpurun_out/r05_pmc_stress/pass* $R/gpurun_out/r05_pmc_bf16/pass*
