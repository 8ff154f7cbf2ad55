# kernel trace of the pose stage (tools/bench_posenet.py 4 = 16 person cubes): steady-state tail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_pose_trace
SP3D_POSE_ONLY=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_pose_trace -o pose -- \
    python $R/tools/bench_posenet.py 4 > $O/${TAG}_pose_trace.json 2> $O/${TAG}_pose_trace.err
python $R/tools/trace_tail.py $O/${TAG}_pose_trace --ms ${2:-25} --top 40 > $O/${TAG}_pose_stage_kernels.md
rm -rf $O/${TAG}_pose_trace
tail -3 $O/${TAG}_pose_trace.json; cat $O/${TAG}_pose_stage_kernels.md | cut -c1-190
