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
SOURCES = ["sp3d_unproject.hip", "sp3d_proposal.hip", "sp3d_epilogue.hip", "sp3d_synth.hip", "sp3d_fftconv.hip", "sp3d_winograd.hip", "sp3d_fft.hip", "sp3d_gbn.hip"]
HEADERS = [os.path.join("..", "pk_src1.py"), "sp3d_device.h", "sp3d_proj_pk.h", "sp3d_tuning.h", "sp3d_twiddles.h", os.path.join("..", "..", "include", "sp3d.h")]
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


# per-source extra flags.  sp3d_winograd.hip: no SLP vectorisation - next to matrix instructions the v_pk_add_f32 /
# v_pk_fma_f32 the vectoriser forms out of the operand transforms cost more than the scalar instructions they replace
# (half-resolution fused Winograd kernel 83.4 -> 78.5 us); the unprojection kernels, on the other hand, want it.
PER_SOURCE_FLAGS = {"sp3d_winograd.hip": ["-fno-slp-vectorize"],
                    # round 4: the packed instructions the vectoriser formed in these kernels (complex multiplies: swapped
                    # halves of source 1) came out wrong next to another plan's matrix instructions; without the vectoriser
                    # they are immune at the same speed (bench step 1.563 vs 1.569 ms).  Round 5 found the one affected
                    # instruction form and `_compile_one` rewrites it in every source (pk_src1.py), so this flag is no longer
                    # what protects them - it stays because it costs nothing.
                    "sp3d_fft.hip": ["-fno-slp-vectorize"], "sp3d_fftconv.hip": ["-fno-slp-vectorize"]}


LLVM_BIN = os.environ.get("SP3D_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

# The second flavour of the library, for a GPU that is SHARED between streams or processes (SP3D_SHARED_GPU=1, _lib.load):
# no packed-fp32 instruction anywhere (-DSP3D_NO_PK: the hand-written pairs as two plain VALU instructions;
# -fno-slp-vectorize: none formed by the compiler).  Same bits; profiles/r04_gpu_sharing_finding.md says why it exists.
NOPK_LIB = os.path.join(HERE, "libsp3d_nopk.so")
# (-target-feature -packed-fp32-ops: the backend itself stops selecting v_pk_{fma,mul,add}_f32, also for explicit float2 /
# float4 vector arithmetic; the host half of the compilation answers "not a recognized feature ... (ignoring)")
NOPK_FLAGS = ["-DSP3D_NO_PK", "-fno-slp-vectorize", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]




def _compile_one(src: str, objdir: str, cflags, verbose: bool, host_flags=()) -> str:
    """One source -> one host object with its gfx950 code object inside, in the steps `hipcc -c` runs internally, with ONE
    step added between the device compile and the assembler: `pk_src1.fix_asm` (no packed-fp32 instruction may take its low
    result from the high half of source 1 - see selfpose3d_amd/pk_src1.py for the measurement behind that rule)."""
    from . import pk_src1
    base = os.path.join(objdir, src.replace(".hip", ""))
    path = os.path.join(CSRC, src)
    cuid = "-cuid=" + src.replace(".", "_")                     # device and host halves must agree on it

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if os.environ.get("SP3D_PLAIN_HIPCC", "0") not in ("", "0"):
        # escape hatch for a toolchain whose internal steps differ from the ones spelled out below (another ROCm layout, a
        # changed bundler): ONE plain `hipcc -c`, no assembly rewrite - and therefore no packed-fp32 instructions at all
        # (NOPK_FLAGS), so that the form pk_src1 exists for cannot occur.  Same results, unprojection kernels 6-9 % slower.
        flags = cflags + [f for f in NOPK_FLAGS if f not in cflags]
        run([HIPCC] + flags + ["-Wno-unused-command-line-argument", "-c", path, "-o", base + ".o"])
        return base + ".o"
    run([HIPCC] + cflags + ["-Wno-unused-command-line-argument", "--cuda-device-only", "-S", cuid, path, "-o", base + ".raw.s"])
    try:
        with open(base + ".raw.s") as fh:
            text, n = pk_src1.fix_asm(fh.read())
    except ValueError as e:
        # an instruction the exchange cannot repair (low result from the HIGH halves of both multiplicands - a legitimate
        # form a compiler update may start to emit): this translation unit is compiled without packed-fp32 instructions
        # instead of failing the build (round-5 advice)
        import warnings
        warnings.warn(f"{src}: {e}; compiling this source without packed-fp32 instructions")
        cflags = cflags + [f for f in NOPK_FLAGS if f not in cflags]
        run([HIPCC] + cflags + ["-Wno-unused-command-line-argument", "--cuda-device-only", "-S", cuid, path, "-o", base + ".raw.s"])
        with open(base + ".raw.s") as fh:
            text, n = pk_src1.fix_asm(fh.read())
    if pk_src1.count_risky(text):
        raise RuntimeError(f"{src}: packed-fp32 instructions reading the high half of source 1 survive the rewrite")
    with open(base + ".s", "w") as fh:
        fh.write(text)
    os.remove(base + ".raw.s")
    if verbose:
        print(f"{src}: {n} packed-fp32 instructions rewritten (source 0 <-> source 1)")
    run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", base + ".s", "-o", base + ".dev.o"])
    run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", base + ".hsaco", base + ".dev.o"])
    run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + base + ".hsaco",
         "-output=" + base + ".hipfb"])
    run([HIPCC] + cflags + list(host_flags) + ["-Wno-unused-command-line-argument", "--cuda-host-only", cuid, "-Xclang",
                                               "-fcuda-include-gpubinary", "-Xclang", base + ".hipfb", "-c", path, "-o", base + ".o"])
    for ext in (".dev.o", ".hsaco", ".hipfb"):
        os.remove(base + ext)
    return base + ".o"


def _compile_objects(objdir: str, extra_flags=(), verbose: bool = False, host_flags=()):
    """every source through `_compile_one`, in parallel; returns the object paths"""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        futs = [pool.submit(_compile_one, src, objdir, cflags + list(extra_flags) + PER_SOURCE_FLAGS.get(src, []), verbose, host_flags)
                for src in SOURCES]
        return [f.result() for f in futs]


def _link(objs, out: str, verbose: bool = False, link_flags=()):
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + LINK + list(link_flags) + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


def build_variant(out: str, extra_flags=(), host_flags=(), link_flags=()) -> str:
    """measurement builds (e.g. -DSP3D_TIMELINE for tools/wave_timeline.py; host_flags = sanitizers for the HOST half only,
    tools/sanitize_host.py); never loaded by the package unless SP3D_LIB_PATH names it"""
    objdir = out + ".obj"
    _link(_compile_objects(objdir, extra_flags, host_flags=host_flags), out, link_flags=link_flags)
    return out


def _stale(lib: str) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        objs = _compile_objects(os.path.join(HERE, "build", "obj"), verbose=verbose)
        _link(objs, LIB + ".tmp", verbose)
        os.replace(LIB + ".tmp", LIB)
    if force or _stale(NOPK_LIB):
        objs = _compile_objects(os.path.join(HERE, "build", "obj_nopk"), NOPK_FLAGS, verbose=verbose)
        _link(objs, NOPK_LIB + ".tmp", verbose)
        os.replace(NOPK_LIB + ".tmp", NOPK_LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
