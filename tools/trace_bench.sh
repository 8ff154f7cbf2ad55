R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=r02c
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-cold --no-fp32-leg > $O/${TAG}_trace_bench.json 2> $O/${TAG}_trace.err
python $R/tools/rocpd_stats.py $O/${TAG}_trace --sequence > $O/${TAG}_bench_kernel_stats.md
find $O/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
rm -rf $O/${TAG}_trace
grep -A90 "in launch order" $O/${TAG}_bench_kernel_stats.md | cut -c1-150
