#!/usr/bin/env python3
"""Sanitizer build of the HOST side of libsp3d.so (the reference has none; argument checks, geometry /
XCD-map set-up, plan caches, launch code are host C++ inside the .hip sources): the host half of every source compiled with
AddressSanitizer + UndefinedBehaviorSanitizer (device code unchanged), then the CPU tests that drive the C ABI without a GPU
(tests/test_host_cabi.py: every export, every error path) run against it under the sanitizer runtime.
    python tools/sanitize_host.py            # build + run; exit status = pytest's; log -> gpurun_out/sanitize_host.txt
(--gpu adds the GPU parity tests; on the round-6 test box the HIP runtime does not come up under the ASan runtime - the process ends
at the first device call without a report - so the recorded run is the CPU one: profiles/r06_sanitize_host_cpu.txt, 23 passed.)"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from selfpose3d_amd import build   # noqa: E402

out = os.path.join(ROOT, "build_tools", "libsp3d_asan.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-shared-libsan"]
build.build_variant(out, host_flags=san, link_flags=["-fsanitize=address,undefined", "-shared-libsan"])
rt = glob.glob(os.path.join(build.LLVM_BIN, "..", "lib", "clang", "*", "lib", "linux", "libclang_rt.asan-x86_64.so"))[0]
env = dict(os.environ, SP3D_LIB_PATH=out, LD_PRELOAD=os.path.realpath(rt), PYTHONPATH=ROOT,
           ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:verify_asan_link_order=0",
           UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
tests = ["tests/test_host_cabi.py", "-k", "not build and not nopk and not flavour and not finished and not packed and not selector"]
if "--gpu" in sys.argv:
    tests = ["tests/test_host_cabi.py", "tests/test_gpu_parity.py", "tests/test_gpu_fused_zdft.py", "-m", "gpu or not gpu",
             "-k", "not build and not nopk and not flavour and not finished and not packed and not selector"]
r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q"] + tests, cwd=ROOT, env=env, capture_output=True, text=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = r.stdout[-6000:] + "\n==== stderr ====\n" + r.stderr[-6000:]
open(os.path.join(ROOT, "gpurun_out", "sanitize_host.txt"), "w").write(log)
print(log[-3000:])
sys.exit(r.returncode)
