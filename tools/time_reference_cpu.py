#!/usr/bin/env python3
"""Time the REFERENCE's own Python on this host's CPU cores (build container only: needs /root/reference).

    python tools/time_reference_cpu.py            # -> profiles/cpu_reference.json

Imports /root/reference/lib through the survey's shims (tests/golden/make_goldens.py) and times, on the synthetic
scene of SURVEY.md §8(d), at torch threads in {1, all cores}:
    ProjectLayer.forward            (lib/models/project_layer.py:42-102)
    CuboidProposalNet.forward       (lib/models/cuboid_proposal_net.py:102-122: unprojection + V2V + NMS)
for BASELINE configs[0] (B=1, 384x288 -> 96x72 heat-maps) and configs[1] (B=4, 960x512 -> 240x128), 80x80x20 voxels.
Median of >=5 runs after 2 warm-ups.  "Speed" follows lib/core/function.py:140 (views x frames / s); frames/s beside it.
bench.py embeds the file as `cpu_reference` (static: the reference cannot travel to the GPU box)."""
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import make_goldens as mg
from selfpose3d_amd import synthetic as syn


def median_time(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    mg._install_shims()
    from models.project_layer import ProjectLayer
    from models.cuboid_proposal_net import CuboidProposalNet
    cores = os.cpu_count()
    model = ""
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    out = {"host": {"cpu": model, "logical_cpus": cores, "machine": platform.machine(), "torch": torch.__version__,
                    "note": "build container (the GPU box never has /root/reference); reference imported unmodified "
                            "via the SURVEY App. C shims"},
           "convention": "speed_views_x_frames_per_s = V*B/t (lib/core/function.py:140); frames_per_s = B/t",
           "configs": {}}
    V, J = 5, 15
    for name, B, img, hm in (("configs[0] B=1 384x288->96x72", 1, (384, 288), (96, 72)),
                             ("configs[1] B=4 960x512->240x128", 4, (960, 512), (240, 128))):
        cfg = mg.make_cfg(img, hm, syn.SPACE_SIZE, syn.SPACE_CENTER, syn.INITIAL_CUBE_SIZE, syn.FINE_GRID_SIZE, (64, 64, 64), J)
        layer = ProjectLayer(cfg)
        net = CuboidProposalNet(cfg)
        syn.fill_parameters_deterministic(net, seed=71, scale=0.05)
        net.eval()
        meta = syn.make_meta(B, V, img)
        hms = syn.random_heatmaps(B, V, J, hm[1], hm[0], seed=1000)
        rec = {}
        for threads in (1, cores):
            torch.set_num_threads(threads)
            with torch.no_grad():
                t_pl = median_time(lambda: layer(hms, meta, list(syn.SPACE_SIZE), [list(syn.SPACE_CENTER)],
                                                 list(syn.INITIAL_CUBE_SIZE)))
                t_net = median_time(lambda: net(hms, meta))
            rec[f"threads_{threads}"] = {
                "ProjectLayer.forward_ms": round(t_pl * 1e3, 2),
                "CuboidProposalNet.forward_ms": round(t_net * 1e3, 2),
                "frames_per_s": round(B / t_net, 3), "speed_views_x_frames_per_s": round(V * B / t_net, 3),
                "unprojection_share": round(t_pl / t_net, 3)}
            print(name, threads, rec[f"threads_{threads}"], flush=True)
        out["configs"][name] = rec
    with open(os.path.join(ROOT, "profiles", "cpu_reference.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
