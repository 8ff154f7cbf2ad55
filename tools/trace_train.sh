# kernel trace of the full-size train step (bench.py's train_step leg alone): where the 138 ms go (last 3 steps of the trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_train_trace
timeout 1500 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_train_trace -o train -- \
    python $R/tools/run_train_leg.py > $O/${TAG}_train_trace_leg.json 2> $O/${TAG}_train_trace.err
python $R/tools/trace_tail.py $O/${TAG}_train_trace --ms ${MS:-410} --top 60 > $O/${TAG}_train_step_kernels.md
rm -rf $O/${TAG}_train_trace
cat $O/${TAG}_train_trace_leg.json; cat $O/${TAG}_train_step_kernels.md | cut -c1-200
