#!/usr/bin/env python3
"""Two processes on ONE GPU (bench.py --share-gpu found it): is the root-net forward still deterministic, and if not, which
stage differs?  Each process runs the eager forward N times and compares every stage's output with its first iteration and,
for the final output, with the reference golden.   python tools/diag_concurrency.py [nproc] [iters] [bench flags...]"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import importlib.util, numpy as np, torch
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    rank, iters = int(sys.argv[2]), int(sys.argv[3])
    flags = sys.argv[4:]
    dev = torch.device("cuda:0")
    bench.use_shipped_miopen_db()
    cfg, meta, hms, model, golden = bench.build_workload(4, 0, dev, "ncdhw" if "--ncdhw" in flags else "cl3d", "direct" if "--direct" in flags else "fft",
                                                         "--no-winograd" not in flags, "--planar-input" in flags, False)
    stages = {}
    v2v = model.v2v_net
    v2v.register_forward_hook(lambda m, i, o: stages.update(cubes=i[0].detach().clone(), v2v=o.detach().clone()))
    from selfpose3d_amd.project_layer import clear_pack_cache
    first, bad = None, {}
    errs = []
    for it in range(iters):
        clear_pack_cache(); model.project_layer._cam_key = None
        with torch.no_grad():
            out = model(hms, meta)
        torch.cuda.synchronize()
        cur = dict(stages, root=out[0].clone(), gc=out[1].clone())
        if first is None:
            first = cur
        else:
            for k in cur:
                if not torch.equal(cur[k], first[k]):
                    d = float((cur[k].float() - first[k].float()).abs().max())
                    bad.setdefault(k, []).append((it, d))
        errs.append(bench.check_output(out, golden)["root_cubes_max_abs_err"])
    print(json.dumps({"rank": rank, "iters": iters, "golden_err_first": errs[0], "golden_err_max": max(errs),
                      "nondeterministic": {k: {"count": len(v), "max_diff": max(d for _, d in v), "first_iter": v[0][0]} for k, v in bad.items()}}), flush=True)
    sys.exit(0)
nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), str(iters)] + sys.argv[3:], stdout=subprocess.PIPE, text=True) for r in range(nproc)]
for p in procs:
    out, _ = p.communicate()
    print([l for l in out.splitlines() if l.startswith("{")][-1:] or out[-500:])
