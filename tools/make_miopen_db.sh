# Populate a MIOpen user find-db with the convolution shapes of bench.py (headline step, pose stage, train leg), so that
# later runs start from it instead of searching (bench.py copies selfpose3d_amd/miopen_db to a private temp dir and points
# MIOPEN_USER_DB_PATH there).   gpurun -- 'bash tools/make_miopen_db.sh'  then copy gpurun_out/miopen_db/* into the package
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/miopen_db; mkdir -p $O/miopen_db
export MIOPEN_USER_DB_PATH=$O/miopen_db
export SP3D_NO_SHIPPED_MIOPEN_DB=1
s=$(date +%s)
python $R/bench.py --steps 20 --no-cpu-baseline --legs pose_stage,planar_handover,train_step --train-steps 3 --train-warmup 2 2>/dev/null | cut -c1-120
e=$(date +%s); echo "populate wall $((e-s)) s"
ls -la $O/miopen_db; du -sh $O/miopen_db
