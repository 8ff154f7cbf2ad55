#!/usr/bin/env python3
"""Control experiment for diag_concurrency.py: ONLY library kernels (elementwise, rocBLAS GEMM, copy) in N processes sharing
one GPU - are THEY bit-reproducible from iteration to iteration?   python tools/diag_concurrency_torch.py [nproc] [iters]"""
import os, sys, json, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    rank, iters = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    a = torch.randn(4096, 512, generator=g).to(dev); w = torch.randn(512, 512, generator=g).to(dev) * 0.05
    big = torch.randn(64 * 1024 * 1024 // 4, generator=g).to(dev)
    first, bad = None, {}
    for it in range(iters):
        x = a
        outs = {}
        for l in range(12):                      # a chain of dependent kernels, fresh output tensors every time
            x = torch.relu(x @ w) + 0.5 * x
            outs[f"l{l}"] = x
        y = (big * 1.5 + 2.0).view(4096, -1)[:, :512] + x          # a 64 MiB streaming kernel feeding a dependent one
        outs["stream"] = y
        torch.cuda.synchronize()
        if first is None:
            first = {k: v.clone() for k, v in outs.items()}
        else:
            for k, v in outs.items():
                if not torch.equal(v, first[k]):
                    bad.setdefault(k, []).append(it)
    print(json.dumps({"rank": rank, "iters": iters, "nondeterministic": {k: len(v) for k, v in bad.items()}}), flush=True)
    sys.exit(0)
nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), str(iters)], stdout=subprocess.PIPE, text=True) for r in range(nproc)]
for p in procs:
    out, _ = p.communicate()
    print([l for l in out.splitlines() if l.startswith("{")][-1:] or out[-500:])
