# which kernel of the frequency-domain opening conv is disturbed by a neighbour running the half-resolution fused Winograd layer?
cd $GRAFT_REPO_ROOT
DIAG_AGGRESSOR=60 DIAG_AGGRESSOR_STAGE=half64 python tools/diag_concurrency3.py --child 9 1 > /dev/null 2>&1 &
sleep 12
python tools/diag_concurrency5.py --child 0 40 2>/dev/null | tail -1
wait
