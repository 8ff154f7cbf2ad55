#!/usr/bin/env python3
"""diag_concurrency.py, stage by stage: (a) unprojection with the camera table re-uploaded every iteration, (b) with the
cached table, (c) the V2V plan alone on fixed cubes, (d) NMS alone - each N iterations in P processes sharing one GPU,
every output compared bit for bit with the process's first.   python tools/diag_concurrency2.py [nproc] [iters]"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import importlib.util, numpy as np, torch
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    rank, iters = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda:0")
    if os.environ.get("DIAG_VA_SHIFT") and rank > 0:      # a different virtual-address layout in every process but the first
        _keep = [torch.empty(rank * 777 * 1024 * 1024 + 4096 * 13, dtype=torch.uint8, device=dev), torch.empty(3 * 1024 * 1024 + 512, dtype=torch.uint8, device=dev)]
    cfg, meta, hms, model, golden = bench.build_workload(4, 0, dev, "cl3d", "fft", True, False, False)
    from selfpose3d_amd import _lib
    from selfpose3d_amd.project_layer import clear_pack_cache
    if os.environ.get("DIAG_NO_DIRECT"):          # full-resolution 3x3x3 layers on the fused Winograd kernel (20 KB of LDS) instead
        model.v2v_net.direct_conv = False         # of the direct split convolution (152 KB of LDS per workgroup)
    pl = model.project_layer
    res = {}

    def loop(name, fn):
        first, bad = None, 0
        for it in range(iters):
            out = fn()
            torch.cuda.synchronize()
            if first is None:
                first = [o.clone() for o in out]
            elif not all(torch.equal(a, b) for a, b in zip(out, first)):
                bad += 1
        res[name] = bad
        return first

    def unproj(fresh_table):
        def f():
            if fresh_table:
                clear_pack_cache(); pl._cam_key = None
            c, _ = pl.get_voxel(hms, meta, model.grid_size, [model.grid_center], model.cube_size, want_grids=False,
                                pad_channels=True, channels_last=True)
            return [c]
        return f
    import contextlib
    side = torch.cuda.stream(torch.cuda.Stream(dev)) if os.environ.get("DIAG_SIDE_STREAM") else contextlib.nullcontext()
    res["stream"] = "side stream" if os.environ.get("DIAG_SIDE_STREAM") else "default stream (handle %d)" % torch.cuda.current_stream(dev).cuda_stream
    with torch.no_grad(), side:
        loop("unproject_fresh_camera_table", unproj(True))
        cubes = loop("unproject_cached_camera_table", unproj(False))[0]
        y = loop("v2v_plan_fixed_input", lambda: [model.v2v_net(cubes)])[0]
        loop("nms_fixed_input", lambda: list(_lib.nms_topk(y.squeeze(1).contiguous(), 10)[:2]))
        model.v2v_net.fused_inference = False
        loop("v2v_miopen_fixed_input", lambda: [model.v2v_net(cubes)])
    print(json.dumps({"rank": rank, "iters": iters, "mismatching_iterations": res}), flush=True)
    sys.exit(0)
nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), str(iters)], stdout=subprocess.PIPE, text=True) for r in range(nproc)]
for p in procs:
    out, _ = p.communicate()
    print([l for l in out.splitlines() if l.startswith("{")][-1:] or out[-500:])
