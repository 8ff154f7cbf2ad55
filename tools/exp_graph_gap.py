#!/usr/bin/env python3
"""How much of the headline step is NOT inside its graph?  The step's graph with a clock-stamp kernel as its first and its
last node (chip-wide 100 MHz counter): (last - first) of a replay vs the distance of consecutive replays' first stamps.
    python tools/exp_graph_gap.py > gpurun_out/r06_graph_gap.json"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from selfpose3d_amd import _lib
from selfpose3d_amd.camera_pack import pack_cameras

dev = torch.device("cuda:0")
cfg, meta, hms, model, golden = bench.build_workload(4, 0, dev)
lib = _lib.load()
lib.sp3d_debug_stamp.restype = C.c_int
lib.sp3d_debug_stamp.argtypes = [C.c_void_p, C.c_void_p]
with torch.no_grad():
    for _ in range(3):
        model(hms, meta)
torch.cuda.synchronize()
pl = model.project_layer
tab = torch.from_numpy(pack_cameras(meta, 4, pl.img_size, None)).to(dev)
N = 200
stamps = torch.zeros((N, 2), dtype=torch.int64, device=dev)
slot = torch.zeros(1, dtype=torch.int64, device=dev)
out = {}
for copies in (1, 2):
    graphs = []
    with pl.static_camera_table(tab):
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s), torch.no_grad():
            model(hms, meta)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize()
        scratch = torch.zeros(2, dtype=torch.int64, device=dev)
        for c in range(copies):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                _lib.check(lib.sp3d_debug_stamp(scratch.data_ptr(), _lib._stream(dev)), "stamp")
                model(hms, meta)
                _lib.check(lib.sp3d_debug_stamp(scratch[1:].data_ptr(), _lib._stream(dev)), "stamp")
                stamps_view = stamps          # copy the two stamps into row `slot` of the log, then advance the slot
                stamps.view(-1).index_copy_(0, (slot * 2 + torch.arange(2, device=dev)), scratch)
                slot.add_(1)
            graphs.append(g)
    slot.zero_(); stamps.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        graphs[i % copies].replay()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / N * 1e6
    t = stamps.cpu().numpy().astype(np.float64) / 100.0          # us
    inside = (t[:, 1] - t[:, 0])[20:]
    period = np.diff(t[:, 0])[20:]
    out[f"{copies}_executable(s)"] = {"us_inside_the_graph_first_to_last_stamp": round(float(np.median(inside)), 1),
                                      "us_between_first_stamps_of_consecutive_replays": round(float(np.median(period)), 1),
                                      "us_outside": round(float(np.median(period) - np.median(inside)), 1),
                                      "host_wall_us_per_replay": round(wall, 1)}
print(json.dumps(out, indent=1))
