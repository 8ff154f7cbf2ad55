cd $GRAFT_REPO_ROOT
echo "== CU-partitioned: proc A on CUs 0-127, proc B on 128-255"
HSA_CU_MASK=0:0-127 python tools/diag_concurrency2.py --child 0 100 2>/dev/null | tail -1 &
HSA_CU_MASK=0:128-255 python tools/diag_concurrency2.py --child 1 100 2>/dev/null | tail -1
wait
echo "== both on CUs 0-127 (same half)"
HSA_CU_MASK=0:0-127 python tools/diag_concurrency2.py --child 0 100 2>/dev/null | tail -1 &
HSA_CU_MASK=0:0-127 python tools/diag_concurrency2.py --child 1 100 2>/dev/null | tail -1
wait
