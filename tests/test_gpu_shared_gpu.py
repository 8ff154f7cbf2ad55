"""Round-4 finding turned into code (profiles/r04_gpu_sharing_finding.md): while waves of wino_fused16_kernel execute
v_mfma_f32_16x16x32_bf16 on a CU, packed-fp32 VALU results (v_pk_fma / v_pk_mul / v_pk_add_f32) of OTHER waves on that CU
come out wrong - two streams of one process suffice.  One plan per GPU on one stream (the product's deployment) never
co-schedules them; for a GPU that IS shared the library has a second flavour without any packed-fp32 instruction,
libsp3d_nopk.so, loaded when SP3D_SHARED_GPU=1.  tools/shared_gpu_check.py runs the aggressor on stream A and the victims on
stream B and counts results that differ BITWISE from the single-stream ones."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(shared: bool, iters: int = 8):
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("SP3D_SHARED_GPU", None)
    if shared:
        env["SP3D_SHARED_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shared_gpu_check.py"), "--iters", str(iters)], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "shared_gpu_check_%s.json" % ("nopk" if shared else "default")), "w") as f:
        json.dump(rec, f, indent=1)
    return rec


def test_shared_gpu_flavour_is_immune_on_two_streams():
    """SP3D_SHARED_GPU=1 loads libsp3d_nopk.so; every victim - the unprojection brick kernel, the frequency-domain
    kernels, the whole root-net forward - is bit-identical to its single-stream result next to the aggressor"""
    rec = _check(shared=True)
    assert rec["library"].endswith("libsp3d_nopk.so"), rec
    assert len(rec["victims"]) == 4
    for name, (bad, worst) in rec["victims"].items():
        assert bad == 0 and worst == 0.0, (name, bad, worst)


@pytest.mark.xfail(strict=False, reason="the default flavour keeps its hand-written packed-fp32 unprojection arithmetic "
                                        "(6-9 % faster) and is wrong next to wino_fused16_kernel's matrix instructions on the "
                                        "same CU - by design it needs the GPU to itself (one plan, one stream); not strict: "
                                        "whether a given box / run shows the interaction is not ours to promise")
def test_default_flavour_on_two_streams():
    rec = _check(shared=False)
    assert rec["library"].endswith("libsp3d.so"), rec
    for name, (bad, worst) in rec["victims"].items():
        assert bad == 0, (name, bad, worst)
