#!/usr/bin/env python3
"""Which stage of the V2V plan is the FIRST to differ in an iteration that goes wrong under GPU sharing?  The plan's stage
methods are wrapped to clone their outputs (no host sync in between); per iteration the first differing stage is tallied.
   python tools/diag_concurrency4.py [nproc] [iters]"""
import os, sys, json, subprocess, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import importlib.util, numpy as np, torch
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    rank, iters = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda:0")
    cfg, meta, hms, model, golden = bench.build_workload(4, 0, dev, "cl3d", "fft", True, False, False)
    from selfpose3d_amd import _lib
    with torch.no_grad():
        out0 = model(hms, meta)
        plan = model.v2v_net._plan
        cubes = [None]
        model.v2v_net.register_forward_pre_hook(lambda m, i: cubes.__setitem__(0, i[0].clone()))
        model(hms, meta)
        x_in = cubes[0]
        rec = []
        def wrap(obj, name, tag=None):
            orig = getattr(obj, name)
            def f(*a, **k):
                y = orig(*a, **k)
                label = (tag or name) + (":" + a[1] if len(a) > 1 and isinstance(a[1], str) else "")
                rec.append((label, y.clone()))
                return y
            setattr(obj, name, f)
        for n in ("_front_fft", "_res", "_pool", "_up2x", "_conv3", "_conv1"):
            wrap(plan, n)
        wrap(_lib, "upsample2x_head_", "head")
        first = None
        tally = collections.Counter()
        detail = {}
        for it in range(iters):
            rec.clear()
            y = model.v2v_net(x_in)
            torch.cuda.synchronize()
            cur = list(rec)
            if first is None:
                first = cur
                continue
            for idx, ((la, a), (lb, b)) in enumerate(zip(cur, first)):
                if not torch.equal(a, b):
                    key = f"{idx}:{la}"
                    tally[key] += 1
                    if key not in detail:
                        d = (a - b).abs()
                        nz = torch.nonzero(d.reshape(-1) > 0).flatten()
                        detail[key] = {"max": float(d.max()), "n_bad": int(nz.numel()), "numel": int(d.numel()),
                                       "first_bad_flat": int(nz[0]), "last_bad_flat": int(nz[-1]), "shape": list(a.shape), "stride": list(a.stride())}
                    break
        print(json.dumps({"rank": rank, "iters": iters, "stages": [l for l, _ in first], "first_differing_stage": dict(tally), "detail": detail}), flush=True)
    sys.exit(0)
nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), str(iters)], stdout=subprocess.PIPE, text=True) for r in range(nproc)]
for p in procs:
    out, _ = p.communicate()
    print([l for l in out.splitlines() if l.startswith("{")][-1:] or out[-1500:])
