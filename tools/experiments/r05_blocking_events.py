"""A/B: how the host waits for a slot of GraphedRootNet's pinned camera ring - spinning (HIP's default event wait) or
sleeping on the completion interrupt (torch.cuda.Event(blocking=True)).  Same captured step, same box, alternating windows.
Per variant: ms per step (max over 3 windows and min), CPU ms of the Python thread and of the whole process per step.
    python tools/experiments/r05_blocking_events.py > gpurun_out/r05_blocking_events.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from selfpose3d_amd import distributed as D  # noqa: E402
from selfpose3d_amd.graphs import GraphedRootNet  # noqa: E402
from selfpose3d_amd.project_layer import clear_pack_cache  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    bench.use_shipped_miopen_db()
    torch.backends.cudnn.benchmark = True
    cfg, meta, hms, model, golden = bench.build_workload(4, 0, dev)
    for _ in range(2):
        clear_pack_cache()
        model.project_layer._cam_key = None
        with torch.no_grad():
            model(hms, meta)
    torch.cuda.synchronize(dev)
    graphs = {}
    for name, blocking in (("spin", False), ("blocking", True)):
        GraphedRootNet.BLOCKING_EVENTS = blocking
        graphs[name] = GraphedRootNet(model, hms, meta)
        assert graphs[name].blocking_events == blocking
    rec = {name: {"ms_per_step": [], "host_ms_per_step": [], "process_cpu_ms_per_step": []} for name in graphs}
    steps = 300
    for rep in range(4):
        for name, g in graphs.items():
            el, cpu_t, cpu_p = D.timed_steps_host(lambda: g(), steps, dev)
            if rep == 0:
                continue                           # first window: settle
            r = rec[name]
            r["ms_per_step"].append(round(1e3 * el / steps, 4))
            r["host_ms_per_step"].append(round(1e3 * cpu_t / steps, 4))
            r["process_cpu_ms_per_step"].append(round(1e3 * cpu_p / steps, 4))
    a = graphs["spin"]()[0].clone()
    b = graphs["blocking"]()[0].clone()
    torch.cuda.synchronize(dev)
    rec["outputs_bit_identical"] = bool(torch.equal(a, b))
    rec["steps_per_window"] = steps
    rec["what"] = ("GraphedRootNet step (BASELINE configs[1], batch 4) with spinning vs blocking ring-slot events; three windows each, "
                   "alternating; host_ms = CPU time of the Python thread, process_cpu = all threads of the process")
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
