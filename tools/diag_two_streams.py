#!/usr/bin/env python3
"""ONE process, two streams: the half-resolution fused Winograd layer (wino_fused16_kernel<64,...>) loops on stream A while a
victim runs on stream B and is compared with its own result obtained alone.  Victims: libsp3d kernels and LIBRARY kernels.
Also the reverse pairing: other aggressors (full-resolution direct convolution, quarter-resolution layer, fp32 fused Winograd)."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from selfpose3d_amd import _lib
if os.environ.get("SP3D_LIB"):                      # a measurement build (e.g. -DSP3D_W16_ABLATE=mask) instead of the shipped library
    _lib.LIB_PATH = os.path.abspath(os.environ["SP3D_LIB"])
ONLY_FIRST = bool(os.environ.get("DIAG_ONLY_FIRST"))
dev = torch.device("cuda:0")
cfg, meta, hms, model, golden = bench.build_workload(4, 0, dev, "cl3d", "fft", True, False, False)
g = torch.Generator().manual_seed(5)
cl = lambda *s: torch.rand(*s, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 25
with torch.no_grad():
    model(hms, meta); plan = model.v2v_net._plan
    x32, h32, h64, q128 = cl(4, 32, 80, 80, 20), cl(4, 32, 40, 40, 10), cl(4, 64, 40, 40, 10), cl(4, 128, 20, 20, 5)
    spec0 = torch.view_as_complex(torch.rand(4 * 16 * 15, 88, 88, 2, generator=g).to(dev)).contiguous()
    wz = torch.view_as_complex(torch.rand(16, 16, 15, 88, 88, 2, generator=g).to(dev)).contiguous()
    sp5 = spec0.view(4, 16, 15, 88, 88)
    a32 = torch.rand(8192, 1024, generator=g).to(dev); w32 = torch.rand(1024, 1024, generator=g).to(dev)
    big = torch.rand(64 * 1024 * 1024, generator=g).to(dev)
    shift = torch.rand(16, generator=g).to(dev); x16 = cl(4, 16, 80, 80, 20)
    pl = model.project_layer

    def unproj():
        c, _ = pl.get_voxel(hms, meta, model.grid_size, [model.grid_center], model.cube_size, want_grids=False,
                            pad_channels=True, channels_last=True)
        return c
    victims = {"unprojection (brick kernel, explicit v_pk)": unproj, "freq_contract": lambda: _lib.freq_contract(sp5, wz),
               "cfft2d_88 fwd+inv": lambda: _lib.cfft2d_(_lib.cfft2d_(spec0.clone(), False, rows_in=80), True, rows_out=80),
               "zdft_fwd": lambda: torch.view_as_real(_lib.zdft_fwd_cl(x16, 16, (88, 88, 28))),
               "library fp32 GEMM": lambda: a32 @ w32,
               "library elementwise (exp, mul, add)": lambda: torch.exp(big * 0.5) * 1.5 + big,
               "library complex mul": lambda: torch.view_as_real(spec0 * spec0),
               "library copy": lambda: big.clone()}
    aggressors = {"half-res fused Winograd (wino_fused16<64>)": lambda: plan._res(h64, "skip_res2"),
                  "half-res 32->64 (wino_fused16<32>)": lambda: plan._res(h32, "encoder_res1"),
                  "full-res direct conv (conv3_split)": lambda: plan._res(x32, "skip_res1"),
                  "quarter-res (wino_input + rocBLAS + wino_output)": lambda: plan._res(q128, "mid_res")}
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    out = {}
    for an, afn in aggressors.items():
        if ONLY_FIRST and an != next(iter(aggressors)):
            continue
        row = {}
        for vn, vfn in victims.items():
            if (an != next(iter(aggressors)) or ONLY_FIRST) and vn not in ("freq_contract", "library complex mul", "unprojection (brick kernel, explicit v_pk)"):
                continue
            with torch.cuda.stream(sb):
                ref = vfn().clone()
            torch.cuda.synchronize()
            bad, worst = 0, 0.0
            for it in range(iters):
                with torch.cuda.stream(sa):
                    for _ in range(10): afn()
                with torch.cuda.stream(sb):
                    outs = [vfn() for _ in range(4)]
                torch.cuda.synchronize()
                for o in outs:
                    if not torch.equal(o, ref):
                        bad += 1; worst = max(worst, float((o.float() - ref.float()).abs().max())); break
            row[vn] = [bad, worst]
        out[an] = row
    print(json.dumps({"iterations": iters, "mismatching_iterations_and_max_abs_diff": out}, indent=1))
