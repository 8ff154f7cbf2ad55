cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/full_gpu_tests.log
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
python tools/exp_wino_split.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r03_conv3_split_ab.json
( for v in "" "--s3" "--s3 --var=sl2" "--s3 --var=sl8" "--s3 --var=nt" "--s3 --var=same"; do echo "## conv3_timeline.py $v"; python tools/conv3_timeline.py $v 2>/dev/null | grep -v amdgpu.ids | cut -c1-1200; done ) > gpurun_out/r03_conv3_timeline.txt
bash tools/trace_bench.sh > /dev/null 2>&1
tail -3 gpurun_out/full_gpu_tests.log; python -c "
import json; d=json.load(open('gpurun_out/r03_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_in_step'), d['roofline'].get('frac_behind_producer'), d['legs'].keys())"
head -12 gpurun_out/r03_bench_kernel_stats.md
