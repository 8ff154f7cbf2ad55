# one copy of the plan check (tools/diag_concurrency2.py) next to a neighbour that leaves NaNs in LDS / registers
cd $GRAFT_REPO_ROOT
for mode in lds vgpr; do
  build_tools/poison_neighbour $mode 45 > /tmp/poison_$mode.txt &
  sleep 6
  echo "== neighbour poisons $mode"; python tools/diag_concurrency2.py --child 0 60 2>/dev/null | tail -1
  wait; cat /tmp/poison_$mode.txt
done
