#!/usr/bin/env python3
"""diag_concurrency2.py found the V2V inference plan irreproducible when processes share a GPU: which layer?  Every stage of
the plan on FIXED inputs, N iterations, P processes.   python tools/diag_concurrency3.py [nproc] [iters]"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import importlib.util, numpy as np, torch
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    rank, iters = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda:0")
    cfg, meta, hms, model, golden = bench.build_workload(4, 0, dev, "cl3d", "fft", True, False, False)
    from selfpose3d_amd import _lib
    res = {}
    if os.environ.get("DIAG_AGGRESSOR"):           # neighbour role: the whole plan (or one stage of it), over and over
        import time
        only = os.environ.get("DIAG_AGGRESSOR_STAGE")
        with torch.no_grad():
            out = model(hms, meta)
            plan = model.v2v_net._plan
            g = torch.Generator().manual_seed(3)
            cl = lambda *sh: torch.rand(*sh, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
            x32, h32, h64, q64, q128 = cl(4, 32, 80, 80, 20), cl(4, 32, 40, 40, 10), cl(4, 64, 40, 40, 10), cl(4, 64, 20, 20, 5), cl(4, 128, 20, 20, 5)
            x16 = cl(4, 16, 80, 80, 20); w0, s0 = plan.t["front"]
            stage = {"front": lambda: plan._front_fft(x16, w0, s0), "full32": lambda: plan._res(x32, "skip_res1"), "half64": lambda: plan._res(h64, "skip_res2"),
                     "quarter128": lambda: plan._res(q128, "mid_res"), "up2": lambda: plan._up2x(q128, "decoder_upsample2", h64), "pool": lambda: plan._pool(x32)}
            t0 = time.time()
            while time.time() - t0 < float(os.environ["DIAG_AGGRESSOR"]):
                for _ in range(20):
                    if only: stage[only]()
                    else: model(hms, meta)
                torch.cuda.synchronize()
        print(json.dumps({"aggressor": only or "whole plan"}), flush=True)
        sys.exit(0)
    with torch.no_grad():
        model(hms, meta)                        # builds the plan
        plan = model.v2v_net._plan
        g = torch.Generator().manual_seed(3)

        def cl(*shape):
            return torch.rand(*shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)

        def loop(name, fn):
            first, bad, worst = None, 0, 0.0
            for it in range(iters):
                out = fn()
                torch.cuda.synchronize()
                if first is None:
                    first = out.clone()
                elif not torch.equal(out, first):
                    bad += 1
                    worst = max(worst, float((out - first).abs().max()))
            res[name] = [bad, worst]
        x16 = cl(4, 16, 80, 80, 20)
        w0, s0 = plan.t["front"]
        loop("front_fft", lambda: plan._front_fft(x16, w0, s0))
        x = plan._front_fft(x16, w0, s0).clone()
        loop("front_res(16->32,full)", lambda: plan._res(x, "front_res"))
        x32 = cl(4, 32, 80, 80, 20)
        loop("skip_res1(32,full)", lambda: plan._res(x32, "skip_res1"))
        loop("pool(full)", lambda: plan._pool(x32))
        h32 = cl(4, 32, 40, 40, 10)
        loop("encoder_res1(32->64,half)", lambda: plan._res(h32, "encoder_res1"))
        h64 = cl(4, 64, 40, 40, 10)
        loop("skip_res2(64,half)", lambda: plan._res(h64, "skip_res2"))
        q64 = cl(4, 64, 20, 20, 5)
        loop("encoder_res2(64->128,quarter)", lambda: plan._res(q64, "encoder_res2"))
        q128 = cl(4, 128, 20, 20, 5)
        loop("mid_res(128,quarter)", lambda: plan._res(q128, "mid_res"))
        loop("decoder_res2(128,quarter)", lambda: plan._res(q128, "decoder_res2"))
        loop("upsample2(quarter->half)", lambda: plan._up2x(q128, "decoder_upsample2", h64))
        loop("decoder_res1(64,half)", lambda: plan._res(h64, "decoder_res1"))
        o = model.v2v_net.output_layer
        wT, sT, wg = plan.t["decoder_upsample1"]
        loop("upsample1+head(half->full)", lambda: _lib.upsample2x_head_(h64, wg, sT, x32, o.weight, o.bias))
    print(json.dumps({"rank": rank, "iters": iters, "mismatching_iterations_and_max_diff": res}), flush=True)
    sys.exit(0)
nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), str(iters)], stdout=subprocess.PIPE, text=True) for r in range(nproc)]
for p in procs:
    out, _ = p.communicate()
    print([l for l in out.splitlines() if l.startswith("{")][-1:] or out[-800:])
