R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o bench -- \
    python $R/bench.py --steps 20 --no-cpu-baseline --no-fp32-leg --no-cold --legs none > $O/${TAG}_trace_bench.json 2> $O/${TAG}_trace.err
python $R/tools/rocpd_stats.py $O/${TAG}_trace > $O/${TAG}_bench_kernel_stats.md 2>> $O/${TAG}_trace.err
find $O/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
rm -rf $O/${TAG}_trace
