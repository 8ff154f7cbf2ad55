# kernel trace of the full-size train step (bench.py --legs train_step): where the 138 ms go (last 3 steps of the trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_train_trace
timeout 1500 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_train_trace -o train -- \
    python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-cold --legs train_step --train-steps 6 --train-warmup 3 \
    > $O/${TAG}_train_trace_bench.json 2> $O/${TAG}_train_trace.err
python $R/tools/trace_tail.py $O/${TAG}_train_trace --ms 420 --top 45 > $O/${TAG}_train_step_kernels.md
rm -rf $O/${TAG}_train_trace
cat $O/${TAG}_train_step_kernels.md | cut -c1-220
