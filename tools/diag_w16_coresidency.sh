# which part of wino_fused16_kernel disturbs a co-resident freq_contract_kernel?  measurement builds -DSP3D_W16_ABLATE=mask
# (1 no matrix instructions, 2 no weight loads, 4 no operand splits, 8 no LDS reads, 15 all of them) under diag_two_streams.py
cd $GRAFT_REPO_ROOT
echo "== shipped"; DIAG_ONLY_FIRST=1 python tools/diag_two_streams.py 12 2>/dev/null | tr -d '\n ' | cut -c1-400; echo
for m in 1 2 4 8 15; do
  echo "== ablate $m"; SP3D_LIB=selfpose3d_amd/ablate/libsp3d_w16_$m.so DIAG_ONLY_FIRST=1 python tools/diag_two_streams.py 12 2>/dev/null | tr -d '\n ' | cut -c1-400; echo
done
