"""Build libsp3d.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m selfpose3d_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.  hipcc
cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["sp3d_unproject.hip", "sp3d_proposal.hip", "sp3d_epilogue.hip", "sp3d_synth.hip", "sp3d_fftconv.hip", "sp3d_winograd.hip", "sp3d_fft.hip"]
HEADERS = ["sp3d_device.h", "sp3d_tuning.h", os.path.join("..", "..", "include", "sp3d.h")]
LIB = os.path.join(HERE, "libsp3d.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]
LINK = ["-L/opt/rocm/lib", "-lhipfft"]      # torch loads its own libhipfft.so.0 first; the loader reuses that instance


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_variant(out: str, extra_flags=()) -> str:
    """measurement builds (e.g. -DSP3D_TIMELINE for tools/wave_timeline.py); never loaded by the package"""
    subprocess.check_call([HIPCC] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + LINK + ["-o", out])
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + LINK + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
