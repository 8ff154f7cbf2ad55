#!/bin/bash
# Aggressor-side experiments for the GPU-sharing finding (round 5): does another form of wino_fused16_kernel stop corrupting
# packed-fp32 neighbours?  Builds measurement libraries next to the shipped one and runs tools/shared_gpu_check.py on each
# (victims keep their packed arithmetic: the DEFAULT flavour of everything else).
#   scalar_acc   -DSP3D_W16_SCALAR_ACC    the kernel's own accumulator updates as v_fma_f32 instead of v_pk_fma_f32
#   prio0/prio3  -DSP3D_W16_SETPRIO=n     s_setprio n in the matrix-instruction waves
#   ablate_1     -DSP3D_W16_ABLATE=1      no matrix instructions at all (round 4's control: clean)
set -u
cd "$(dirname "$0")/.."
mkdir -p selfpose3d_amd/ablate gpurun_out
ITERS=${1:-8}
echo "== shipped"; python tools/shared_gpu_check.py --iters $ITERS --only "brick" 2>/dev/null | tail -1
for v in "scalar_acc:-DSP3D_W16_SCALAR_ACC" "prio0:-DSP3D_W16_SETPRIO=0" "prio3:-DSP3D_W16_SETPRIO=3" "ablate_1:-DSP3D_W16_ABLATE=1"; do
  name=${v%%:*}; flag=${v#*:}
  lib=selfpose3d_amd/ablate/libsp3d_w16_$name.so
  [ -f $lib ] || python -c "from selfpose3d_amd import build as b; b.build_variant('$lib', ['$flag'])" >/dev/null 2>&1
  echo "== $name ($flag)"; python tools/shared_gpu_check.py --iters $ITERS --only "brick" --lib $lib 2>/dev/null | tail -1
done
