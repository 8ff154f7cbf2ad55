#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 24; do
  rm -rf $R/gpurun_out/pf_$v
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pf_$v -o pmc -- python $R/tools/run_kernel.py --workload coarse_b4_v5 --variant $v --iters 12 > /dev/null 2>&1
  python - <<PY
import csv,glob
rows=[r for f in glob.glob("$R/gpurun_out/pf_$v/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "unproject" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
vals=[float(r["Counter_Value"]) for r in rows]
print("variant $v FETCH_SIZE KB mean", sum(vals)/max(1,len(vals)), "-> read MB", 2*sum(vals)/max(1,len(vals))*1024/1e6)
PY
done
