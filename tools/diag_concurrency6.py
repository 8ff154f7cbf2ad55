#!/usr/bin/env python3
"""One process runs the V2V plan (diag_concurrency2's check); its neighbour on the same GPU is (a) nothing, (b) a torch-only
GEMM loop, (c) another copy of the plan.   python tools/diag_concurrency6.py [iters]"""
import os, sys, json, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
iters = sys.argv[1] if len(sys.argv) > 1 else "120"
me = [sys.executable, os.path.join(ROOT, "tools", "diag_concurrency2.py"), "--child", "0", iters]
load = [sys.executable, "-c", "import torch,time\nd=torch.device('cuda:0')\na=torch.randn(4096,4096,device=d)\nt=time.time()\nwhile time.time()-t<%s:\n    b=a@a\n    torch.cuda.synchronize()\n" % "60"]
light = [sys.executable, "-c", "import torch,time\nd=torch.device('cuda:0')\na=torch.randn(256,256,device=d)\nt=time.time()\nwhile time.time()-t<%s:\n    b=torch.relu(a)+1\n    torch.cuda.synchronize()\n" % "60"]
for name, nb in (("alone", None), ("next to a torch GEMM loop", load), ("next to a light torch elementwise loop", light), ("next to another copy", me)):
    n = subprocess.Popen(nb, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) if nb else None
    if n is not None:
        time.sleep(8)
    p = subprocess.run(me, capture_output=True, text=True)
    if n is not None:
        n.kill(); n.wait()
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print(name, "->", line[-1] if line else p.stderr[-400:], flush=True)
