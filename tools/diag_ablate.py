#!/usr/bin/env python3
"""Ablation timing of the pipelined unprojection kernels (measurement only; never shipped).

One library per -DSP3D_ABLATE=<mask> (compile-time, so every variant is a cleanly optimised kernel): a part of the
kernel is REMOVED (result stores, tap loads, projection, FMAs) or the wave starts are staggered, and the rest is timed
with HIP events - shows which parts overlap and which add up.

    python tools/diag_ablate.py --build-only          # CPU box: cross-compile the libraries
    python tools/diag_ablate.py [--variants 24,56] > gpurun_out/diag.json
"""
import argparse, ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)

MASKS = {"full": 0, "no_store": 1, "no_loads": 2, "no_proj": 4, "stagger": 8, "no_fma": 16, "no_gather": 2 | 16,
         "only_proj": 2 | 1 | 16, "only_store": 2 | 4 | 16, "only_loads": 4 | 1 | 16, "loads_fma": 4 | 1}
LIBDIR = os.path.join(ROOT, "selfpose3d_amd", "ablate")


def lib_path(mask):
    return os.path.join(LIBDIR, f"libsp3d_ablate{mask}.so")


def build_all():
    from selfpose3d_amd import build as _build
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(_build.CSRC, "sp3d_unproject.hip")
    procs = []
    for mask in sorted(set(MASKS.values())):
        out = lib_path(mask)
        if os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(src):
            continue
        procs.append(subprocess.Popen([_build.HIPCC] + _build.FLAGS + [f"-DSP3D_ABLATE={mask}", src, "-o", out]))
    for p in procs:
        assert p.wait() == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--variants", default="24,56")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--workloads", default="coarse_b4,fine_b10,stress_b1")
    args = ap.parse_args()
    if args.build_only:
        build_all()
        return
    import numpy as np, torch
    from selfpose3d_amd import _lib, synthetic as syn
    from selfpose3d_amd.camera_pack import pack_cameras
    dev = torch.device("cuda:0")
    img, (w, h), J = (960, 512), (240, 128), 15
    wls = {"coarse_b4": (4, 5, syn.INITIAL_CUBE_SIZE, syn.SPACE_SIZE, False),
           "coarse_b1": (1, 5, syn.INITIAL_CUBE_SIZE, syn.SPACE_SIZE, False),
           "fine_b10": (10, 5, syn.FINE_CUBE_SIZE, syn.FINE_GRID_SIZE, True),
           "stress_b1": (1, 10, (160, 160, 40), syn.SPACE_SIZE, False)}
    P, I, V_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
    libs = {}
    for k, m in MASKS.items():
        l = ctypes.CDLL(lib_path(m))
        l.sp3d_unproject_fwd_variant.restype = I
        l.sp3d_unproject_fwd_variant.argtypes = [P, I, P, P, P, P, P, I, I, I, I, I, I, I, I, P, I, I, I, V_]
        assert l.sp3d_debug_set_diag(0) == m
        libs[k] = l

    def timed(fn, iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    out = {}
    for name in args.workloads.split(","):
        B, V, cube, gs, fine = wls[name]
        meta = syn.make_meta(B, V, img)
        cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
        if fine:
            rng = np.random.default_rng(0)
            c = np.stack([rng.uniform(-1500, 1500, B), rng.uniform(-2000, 1000, B), rng.uniform(700, 1100, B)], 1)
            centers = torch.from_numpy(c.astype(np.float32)).to(dev)
        else:
            centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
        valid = torch.ones(B, dtype=torch.uint8, device=dev)
        hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=7)]
        packed = _lib.pack_heatmaps(hms, jp=16)
        views = [packed[c] for c in range(V)]
        ptrs = _lib._ptr_array(views)
        gsz = _lib._f3(gs)
        X, Y, Z = cube
        stream = torch.cuda.current_stream().cuda_stream
        out[name] = {}
        for v in [int(x) for x in args.variants.split(",")]:
            for cl in (False, True):
                Jc = 16 if cl else J
                cubes = torch.empty((B, Jc, X, Y, Z), dtype=torch.float32, device=dev)
                row = {}
                for k, l in libs.items():
                    run = lambda: l.sp3d_unproject_fwd_variant(ptrs, 16, cam.data_ptr(), centers.data_ptr(), valid.data_ptr(),
                                                               cubes.data_ptr(), None, B, V, Jc, h, w, X, Y, Z, gsz, img[0],
                                                               img[1], v | (0x1000000 if cl else 0), stream)
                    assert run() == 0
                    timed(run, 10)
                    row[k] = round(min(timed(run, args.iters) for _ in range(3)), 2)
                out[name][f"v{v}{'_cl' if cl else ''}"] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
