# (1) is a LIBRARY bf16 GEMM neighbour (rocBLAS / hipBLASLt bf16 matrix instructions) enough to disturb freq_contract / cfft2d_88?
# (2) same question inside ONE process: the fused Winograd layer on one stream, the victims on another
cd $GRAFT_REPO_ROOT
python - <<'PY' &
import torch, time
d = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=d, dtype=torch.bfloat16)
t = time.time()
while time.time() - t < 45:
    for _ in range(10): b = a @ a
    torch.cuda.synchronize()
PY
sleep 10
echo "== neighbour process: torch bf16 GEMM loop"; python tools/diag_concurrency5.py --child 0 30 2>/dev/null | tail -1
wait
echo "== ONE process, two streams: wino_fused16 (stream A) next to the victims (stream B)"
python - <<'PY'
import os, sys, json, threading, time, torch
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from selfpose3d_amd import _lib
dev = torch.device("cuda:0")
cfg, meta, hms, model, golden = bench.build_workload(4, 0, dev, "cl3d", "fft", True, False, False)
g = torch.Generator().manual_seed(5)
with torch.no_grad():
    model(hms, meta); plan = model.v2v_net._plan
    h64 = torch.rand(4, 64, 40, 40, 10, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    spec0 = torch.view_as_complex(torch.rand(4 * 16 * 15, 88, 88, 2, generator=g).to(dev)).contiguous()
    wz = torch.view_as_complex(torch.rand(16, 16, 15, 88, 88, 2, generator=g).to(dev)).contiguous()
    sp5 = spec0.view(4, 16, 15, 88, 88)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    res = {}
    for name, fn in (("freq_contract", lambda: _lib.freq_contract(sp5, wz)), ("cfft2d_88", lambda: _lib.cfft2d_(_lib.cfft2d_(spec0.clone(), False, rows_in=80), True, rows_out=80))):
        with torch.cuda.stream(sb):
            ref = fn().clone()
        torch.cuda.synchronize()
        bad = 0
        for it in range(40):
            with torch.cuda.stream(sa):
                for _ in range(12): plan._res(h64, "skip_res2")
            with torch.cuda.stream(sb):
                outs = [fn() for _ in range(6)]
            torch.cuda.synchronize()
            bad += any(not torch.equal(o, ref) for o in outs)
        res[name] = bad
    print(json.dumps({"one_process_two_streams_mismatching_iterations_of_40": res}))
PY
