#!/usr/bin/env python3
"""The GPU-sharing finding as ONE script (round 5; replaces the ~20 diag_concurrency* / diag_victims* / diag_cumask /
diag_poison / diag_stream_ids scripts of round 4, whose results are in profiles/r04_gpu_sharing_finding.md).

  --mode streams    (default) ONE process, two streams: the half-resolution fused Winograd layer (wino_fused16_kernel<64>,
                    v_mfma_f32_16x16x32_bf16) loops on stream A while a victim kernel runs on stream B; every victim
                    result is compared BITWISE with the result the same call gave with the GPU to itself.
  --mode processes  N copies of this script (--copies, default 2) run the root-net forward at the same time on cuda:0;
                    each compares every iteration with its own first.
The library flavour is the one `selfpose3d_amd._lib.load()` picks: libsp3d.so, or with SP3D_SHARED_GPU=1 libsp3d_nopk.so
(no packed-fp32 instruction); --lib PATH loads a measurement build instead.  Prints one JSON line; exit code 0 always (the
caller judges the counts): {"victims": {name: [mismatching iterations, max |diff|]}, "iterations": n, "library": path}.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workload(dev):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench.build_workload(4, 0, dev, "cl3d", "fft", True, False, False)


def streams(args):
    import torch
    from selfpose3d_amd import _lib
    dev = torch.device("cuda:0")
    cfg, meta, hms, model, golden = workload(dev)
    g = torch.Generator().manual_seed(5)
    cl = lambda *s: torch.rand(*s, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    with torch.no_grad():
        model(hms, meta)
        plan = model.v2v_net._plan
        h64 = cl(4, 64, 40, 40, 10)
        spec0 = torch.view_as_complex(torch.rand(4 * 16 * 15, 88, 88, 2, generator=g).to(dev)).contiguous()
        wz = torch.view_as_complex(torch.rand(16, 16, 15, 88, 88, 2, generator=g).to(dev)).contiguous()
        sp5 = spec0.view(4, 16, 15, 88, 88)
        pl = model.project_layer

        def unproj():
            c, _ = pl.get_voxel(hms, meta, model.grid_size, [model.grid_center], model.cube_size, want_grids=False,
                                pad_channels=True, channels_last=True)
            return c
        victims = {"unprojection (brick kernel)": unproj, "freq_contract": lambda: _lib.freq_contract(sp5, wz),
                   "cfft2d_88 fwd+inv": lambda: _lib.cfft2d_(_lib.cfft2d_(spec0.clone(), False, rows_in=80), True, rows_out=80),
                   "root-net forward (whole plan)": lambda: model(hms, meta)[0]}
        aggressor = lambda: plan._res(h64, "skip_res2")           # wino_fused16_kernel<64, ...> twice + epilogues
        sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        torch.cuda.synchronize()
        out = {}
        for vn, vfn in victims.items():
            if args.only and args.only not in vn:
                continue
            with torch.cuda.stream(sb):
                ref = vfn().clone()
            torch.cuda.synchronize()
            bad, worst = 0, 0.0
            for _ in range(args.iters):
                with torch.cuda.stream(sa):
                    for _ in range(10):
                        aggressor()
                with torch.cuda.stream(sb):
                    outs = [vfn().clone() for _ in range(4)]
                torch.cuda.synchronize()
                for o in outs:
                    if not torch.equal(o, ref):
                        bad += 1
                        worst = max(worst, float((o.float() - ref.float()).abs().max()))
                        break
            out[vn] = [bad, worst]
    print(json.dumps({"mode": "streams", "iterations": args.iters, "victims": out, "library": _lib.LIB_PATH,
                      "shared_gpu_env": os.environ.get("SP3D_SHARED_GPU")}))


def processes(args):
    if args.child:
        import torch
        from selfpose3d_amd import _lib
        dev = torch.device("cuda:0")
        cfg, meta, hms, model, golden = workload(dev)
        with torch.no_grad():
            ref = model(hms, meta)[0].clone()
            torch.cuda.synchronize()
            bad, worst = 0, 0.0
            for _ in range(args.iters):
                o = model(hms, meta)[0]
                torch.cuda.synchronize()
                if not torch.equal(o, ref):
                    bad += 1
                    worst = max(worst, float((o - ref).abs().max()))
        print(json.dumps({"bad": bad, "worst": worst, "library": _lib.LIB_PATH}))
        return
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "processes", "--child", "--iters", str(args.iters)]
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(args.copies)]
    recs = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        lines = [ln for ln in so.splitlines() if ln.startswith("{")]
        recs.append(json.loads(lines[-1]) if lines else {"error": se[-400:]})
    print(json.dumps({"mode": "processes", "copies": args.copies, "iterations": args.iters, "per_process": recs,
                      "shared_gpu_env": os.environ.get("SP3D_SHARED_GPU")}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["streams", "processes"], default="streams")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--copies", type=int, default=2)
    ap.add_argument("--only", default="")
    ap.add_argument("--lib", default="")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.lib:
        from selfpose3d_amd import _lib as L
        L.LIB_PATH = os.path.abspath(a.lib)
    (streams if a.mode == "streams" else processes)(a)
