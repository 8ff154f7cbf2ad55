# victim / aggressor matrix: (1) every stage of the plan checked for reproducibility next to a neighbour running the WHOLE plan;
# (2) the whole plan checked next to a neighbour running ONE stage
cd $GRAFT_REPO_ROOT
echo "== victims: each stage, neighbour = whole plan"
DIAG_AGGRESSOR=70 python tools/diag_concurrency3.py --child 9 1 > /dev/null 2>&1 &
sleep 12
python tools/diag_concurrency3.py --child 0 40 2>/dev/null | tail -1
wait
for st in front full32 half64 quarter128 up2 pool; do
  echo "== aggressor: $st only, victim = whole plan"
  DIAG_AGGRESSOR=32 DIAG_AGGRESSOR_STAGE=$st python tools/diag_concurrency3.py --child 9 1 > /dev/null 2>&1 &
  sleep 10
  python tools/diag_concurrency2.py --child 0 60 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['mismatching_iterations']['v2v_plan_fixed_input'], 'of', d['iters'], 'plan iterations differ')"
  wait
done
