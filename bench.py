#!/usr/bin/env python3
"""bench.py - throughput of the SelfPose3d root-localisation hot path on MI355X.

Workload (BASELINE.json configs[1], the config the metric is quoted on):
  Panoptic 5-view `CuboidProposalNet` forward-only, batch 4 per GPU: 5 x (4,15,128,240) heat-maps
  (960x512 network input, YAML-native) resident in HBM -> unprojection into the 80x80x20 grid
  (HIP kernels) -> V2V 3D conv stack (PyTorch-ROCm/MIOpen, fp32) -> fused NMS/top-k (HIP).
  One "step" = one such forward over one batch.  Multi-GPU = one process per GPU, frames sharded
  by rank (weak scaling), NO data-path collective (the path is embarrassingly parallel over frames).

Heat-maps are handed over the way this repo's backbone emits them (PoseResNet.forward_views: (B,15,h,w) views of one
channels-last (V,B,h,w,16) buffer), so the step has no re-tiling pass; `--planar-input` gives the reference's planar
(B,15,h,w) hand-over instead (one extra pack kernel per step).  Samples 0,1 of rank 0's batch are the inputs of the
reference golden tests/golden/rootnet_full.npz and the weights are that golden's: after the timed loop the step's own
output is checked against the reference's root cubes and proposals (`output_check`).

Prints ONE JSON line on rank 0 (contract in the task statement) with three extra objects:
  roofline      - the unprojection kernel: algorithmic bytes / HIP-event time vs 8 TB/s HBM peak (+ path time, cold inputs)
  cpu_baseline  - the CPU oracle (C port, OpenMP) + torch-CPU V2V on a bounded sample of the same workload, all cores and 1
  cpu_reference - the reference's own Python timed in the build container (static: profiles/cpu_reference.json)

N > 1: barriers and the max over ranks run on a gloo group in host memory, RCCL carries only the DDP gradient buckets of
legs.train_step (selfpose3d_amd/distributed.py: init_split), so the headline does not depend on RCCL; the legs after the headline
measurement run under --leg-deadline (600 s): if a collective never returns, rank 0 prints the line it has
(top level: legs_complete / scaling_valid = false) and every rank exits with status 3 - a hang can neither lose the headline nor pass
as a finished scaling run.  config.rccl_ranks_seen = the sum of an all-reduce of ones on the data-plane communicator (must equal N).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


EXIT_LEGS_INCOMPLETE = 3      # world > 1: the line was printed, but a leg hung, a leg with a collective failed or RCCL did not span every rank
COLLECTIVE_LEGS = ("data_plane", "train_step", "train_step_ssv")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="frames per GPU per step (configs[1]: 4)")
    ap.add_argument("--roofline-iters", type=int, default=300)
    ap.add_argument("--no-fp32-leg", action="store_true",
                    help="skip the extra leg that re-times the step with every 3^3 product on fp32 matrix instructions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-reps", type=int, default=20)
    ap.add_argument("--front-conv", choices=["fft", "direct"], default="fft",
                    help="7x7x7 opening conv of V2V: frequency domain (rocFFT + sp3d_freq_contract) or MIOpen direct")
    ap.add_argument("--no-winograd", action="store_true",
                    help="keep the 1/2- and 1/4-resolution 3x3x3 convs on MIOpen instead of Winograd F(2,3) (HIP transforms + rocBLAS)")
    ap.add_argument("--v2v-layout", choices=["ncdhw", "cl3d"], default="cl3d",
                    help="memory format of the V2V stack (fp32 either way)")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of as one HIP graph")
    ap.add_argument("--graph-copies", type=int, default=int(os.environ.get("SP3D_GRAPH_COPIES", "1")),
                    help="graph executables of the step replayed in turn (1: one executable, replays cannot overlap their launch)")
    ap.add_argument("--no-cold", action="store_true", help="skip timing the kernel on inputs rotating through >256 MiB")
    ap.add_argument("--planar-input", action="store_true",
                    help="hand the heat-maps over as the reference does, planar (B,J,h,w): adds the re-tiling pass to the step")
    ap.add_argument("--no-check", action="store_true", help="skip the reference-golden check of the step's output")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="leave library-GEMM selection to the rocBLAS/hipBLASLt heuristics (default: opt in to TunableOp "
                         "for the inference plan's GEMMs, V2VNet.tune_gemms(True); recorded in config.gemm_selection)")
    ap.add_argument("--leg-deadline", type=float, default=600.0,
                    help="world > 1 only: seconds the legs after the headline measurement may take before rank 0 prints the line "
                         "it has (legs_complete / scaling_valid false) and every rank exits with status 3 (a collective that never "
                         "returns must not lose the headline - and must not read as a finished scaling run either); 0 disables")
    ap.add_argument("--legs", default="auto",
                    help="extra legs next to the headline: comma list of pose_stage, train_step, planar_handover, unprojection_grids, unprojection_backward, or 'auto' "
                         "(all four: train_step = BASELINE configs[2], at every N) or 'none'")
    ap.add_argument("--share-gpu", action="store_true",
                    help="smoke mode for boxes with ONE GPU: every rank uses cuda:0 and the process group is gloo (the whole "
                         "multi-rank code path - sharding, DDP gradient all-reduce, max-over-ranks timing - without RCCL); the "
                         "JSON line says so in config.parallelism")
    ap.add_argument("--spread-repeats", type=int, default=3,
                    help="box_spread: the timed K steps + this many - 1 further repeats of K steps (min / max ms_per_step)")
    ap.add_argument("--train-steps", type=int, default=5)
    ap.add_argument("--train-warmup", type=int, default=3)
    ap.add_argument("--train-find", choices=["search", "immediate"], default="search",
                    help="MIOpen kernel selection for the train_step leg: search (cudnn.benchmark, one-off minutes) or immediate")
    return ap.parse_args()


def golden_inputs(cfg):
    """inputs / weights of tests/golden/rootnet_full.npz (reference CuboidProposalNet at this very size): B=2 heat-maps
    (sample 0 uniform x0.35, sample 1 Gaussian 'people') and the deterministic parameter fill"""
    from selfpose3d_amd import synthetic as syn
    g = np.load(os.path.join(ROOT, "tests", "golden", "rootnet_full.npz"))
    V, J = int(g["V"]), int(g["J"])
    w, h = [int(v) for v in g["hm"]]
    seed = int(g["hm_seed"])
    rnd = syn.random_heatmaps(2, V, J, h, w, seed=seed)
    ppl, _ = syn.people_heatmaps(2, V, J, h, w, [int(v) for v in g["img"]], seed=seed + 1)
    return g, [torch.stack([0.35 * rnd[v][0], ppl[v][1]]) for v in range(V)]


def build_workload(batch, rank, dev, v2v_layout="cl3d", front_conv="fft", winograd=True, planar_input=False,
                   tune_gemms=True):
    from selfpose3d_amd import _lib, synthetic as syn
    from selfpose3d_amd.v2v_net import V2VNet
    V2VNet.tune_gemms(bool(tune_gemms))              # explicit opt-in: the library itself leaves TunableOp alone
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.cuboid_proposal_net import CuboidProposalNet
    from selfpose3d_amd.project_layer import nhwc_heatmap_views

    cfg = load_config(None)                         # Panoptic 5-cam defaults == the reference YAML
    V, J = int(cfg.DATASET.CAMERA_NUM), int(cfg.NETWORK.NUM_JOINTS)
    w, h = cfg.NETWORK.HEATMAP_SIZE
    meta = syn.make_meta(batch, V, cfg.NETWORK.IMAGE_SIZE)          # CPU tensors, as a DataLoader emits
    hms = syn.random_heatmaps(batch, V, J, h, w, seed=1000 + rank)
    golden = None
    if rank == 0 and batch >= 2:                    # samples 0,1 = the reference golden's inputs (checked after the run)
        golden, gh = golden_inputs(cfg)
        hms = [torch.cat([gh[v], hms[v][2:]], 0) for v in range(V)]
    hms = [x.to(dev) for x in hms]
    if not planar_input:                            # as PoseResNet.forward_views emits them: views of (V,B,h,w,16)
        hms = nhwc_heatmap_views(_lib.pack_heatmaps(hms, jp=16), J)
    model = CuboidProposalNet(cfg)
    syn.fill_parameters_deterministic(model, seed=71, scale=0.05)   # non-degenerate weights (= the golden's)
    model.eval().to(dev)
    if v2v_layout == "cl3d":
        model.use_channels_last(True)
    model.v2v_net.fft_front = front_conv == "fft"
    model.v2v_net.winograd = bool(winograd)
    return cfg, meta, hms, model, golden


def check_output(out, golden, tol=5e-5):
    """the step's own output (rank 0, samples 0,1) against the reference CuboidProposalNet -> V2VNet -> nms golden"""
    from selfpose3d_amd import synthetic as syn
    root_cubes, grid_centers = out
    rc = root_cubes[:2].float().cpu().numpy()
    N = rc[0].size
    ref = golden["root_sub"]
    scale = float(np.abs(ref).max())
    err = float(np.abs(rc.reshape(2, N)[:, golden["sub_idx"]] - ref).max())
    sum_err = float(np.abs(rc.astype(np.float64).sum(axis=(1, 2, 3)) - golden["root_sum"]).max() / golden["root_abs_sum"].max())
    vals, idx = golden["nms_vals"], golden["nms_idx"]
    gc = grid_centers[:2].float().cpu()
    cs = torch.tensor(syn.INITIAL_CUBE_SIZE, dtype=torch.float32)
    gs, cen = torch.tensor(syn.SPACE_SIZE), torch.tensor(syn.SPACE_CENTER)
    checked = bad = 0
    for b in range(2):
        for k in range(vals.shape[1]):
            gap = min(abs(float(vals[b, k] - vals[b, k - 1])) if k else 9.0,
                      abs(float(vals[b, k] - vals[b, k + 1])) if k + 1 < vals.shape[1] else 9.0)
            if gap > 4 * tol * scale:
                loc = torch.from_numpy(idx[b, k]).float() / (cs - 1) * gs + cen - gs / 2.0
                checked += 1
                bad += 0 if torch.equal(gc[b, k, :3], loc) else 1
    ok = err <= tol * max(1.0, scale) and sum_err <= 1e-6 and bad == 0 and checked >= 10
    return {"ok": bool(ok), "reference": "tests/golden/rootnet_full.npz (reference CuboidProposalNet, CPU fp32)",
            "root_cubes_max_abs_err": err, "root_cubes_range": scale, "checksum_rel_err": sum_err,
            "proposal_indices_checked": checked, "proposal_indices_wrong": bad}


def use_shipped_miopen_db():
    """MIOpen searches every new convolution shape once (minutes for the train leg's ~100 shapes on an empty user db).
    selfpose3d_amd/miopen_db holds the find results of this file's shapes on gfx950 (tools/make_miopen_db.sh): a PRIVATE
    copy of it becomes this process's user db, so warm-ups start from known results (shapes it does not hold are searched as
    usual; a MIOpen of another version ignores the files).  Steady-state times are the same either way - it is the search's
    own result that is cached.  Off with SP3D_NO_SHIPPED_MIOPEN_DB=1 or when MIOPEN_USER_DB_PATH is already set."""
    import shutil
    import tempfile
    src = os.path.join(ROOT, "selfpose3d_amd", "miopen_db")
    if os.environ.get("MIOPEN_USER_DB_PATH") or os.environ.get("SP3D_NO_SHIPPED_MIOPEN_DB") or not os.path.isdir(src):
        return None
    dst = tempfile.mkdtemp(prefix="sp3d_miopen_")
    for f in os.listdir(src):
        if f.endswith(".txt"):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst


def event_time_ms(fn, iters, dev):
    """average duration of fn() launched back-to-back on torch's current stream (HIP events)"""
    for _ in range(10):
        fn()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / iters


def roofline_leg(cfg, meta, hms, model, iters, dev, planar_input, cold=True):
    """The dominant kernel of the unprojection as THIS step runs it (channels-last 16-channel cubes for the opening conv's
    z-DFT pass -> the 4x4x4-brick kernel; planar cubes -> the 64-consecutive-voxel pipelined kernel otherwise), timed
    alone with HIP events on the launch stream; next to it the
    path time (re-tiling pass included when the heat-maps arrive planar), the same kernel on inputs that rotate through
    8 x 39 MB (MALL-cold-ish), and the other unprojection kernels of the library on the same workload."""
    from selfpose3d_amd import _lib
    from selfpose3d_amd.project_layer import _packed_source
    B, J, h, w = hms[0].shape
    V = len(hms)
    pl = model.project_layer
    cam = pl.camera_table(meta, B, None, dev)
    centers, valid = pl.centers_valid([model.grid_center], B, dev)
    cube, gs, img = model.cube_size, model.grid_size, pl.img_size
    N = cube[0] * cube[1] * cube[2]
    src = _packed_source(hms, 16, torch.float32)
    planar = [x.contiguous() for x in hms]
    packed = src if src is not None else _lib.pack_heatmaps(planar, jp=16)
    views = [packed[c] for c in range(V)]

    # what the bench step launches: on this grid the opening conv's z pass reads channels-last 16-channel cubes, so the
    # step runs the 4x4x4-brick kernel (library default for channels-last results); a V2V that wants planar cubes gets
    # the 64-consecutive-voxel kernel
    def k_lin_planar():           # planar result, library default for this grid: 64 consecutive voxels per wave
        _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, J, h, w, cube, gs, img, False)

    def k_brick_cl():             # channels-last result: 4x4x4-brick kernel
        _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16, h, w, cube, gs, img, False,
                           channels_last=True)

    def k_planar():
        _lib.unproject_fwd(planar, _lib.LAYOUT_PLANAR, 0, cam, centers, valid, B, J, h, w, cube, gs, img, False)

    with torch.no_grad():
        fft_view = model.v2v_net.input_view(B, *cube, dev) if hasattr(model.v2v_net, "input_view") else None

    def k_strided():              # the same kernel writing straight into the FFT conv's zero-padded input (as in the step)
        _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, J, h, w, cube, gs, img, False, out=fft_view)

    scratch = torch.empty_like(packed)

    def k_pack():
        _lib.pack_heatmaps(planar, jp=16, out=scratch)

    with torch.no_grad():
        step_cl = bool(getattr(model.v2v_net, "wants_channels_last_cubes", lambda *a: False)(*cube))
        # round 6: on this grid the step runs the unprojection FUSED with the opening conv's z pass (no cubes in HBM at all)
        sz = getattr(model.v2v_net, "wants_zspectrum", lambda *a: None)(*cube, J) if step_cl else None
    step_zd = sz is not None

    def k_fused():                # the launch of the step: heat-maps in, 4x4-tiled z-spectrum of the cubes out
        _lib.unproject_fwd_zdft(views, 16, cam, centers, valid, B, J, h, w, cube, gs, img, sz)

    t_lin = event_time_ms(k_lin_planar, iters, dev)
    t_brick = event_time_ms(k_brick_cl, iters, dev)
    t_fused = t_zdft = None
    if step_zd:
        t_fused = event_time_ms(k_fused, iters, dev)
        cubes_cl = _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16, h, w, cube, gs, img, False,
                                      channels_last=True)[0]
        S = (88, 88, int(sz))
        t_zdft = event_time_ms(lambda: _lib.zdft_fwd_cl(cubes_cl, J, S), iters, dev)
        del cubes_cl
    t_k = t_fused if step_zd else (t_brick if step_cl else t_lin)
    t_planar = event_time_ms(k_planar, max(10, iters // 10), dev)
    t_pack = event_time_ms(k_pack, iters, dev)
    with torch.no_grad():
        t_strided = event_time_ms(k_strided, iters, dev) if fft_view is not None else None
    # algorithmic bytes per launch: read every heat-map element once, write cubes
    # once; `grids` is not requested by the root net so its 3N term is dropped.
    alg_bytes_8d = 4.0 * B * (V * J * h * w + J * N)
    # the fused launch's own algorithmic bytes: every heat-map element read once + its OUTPUT written
    # once - the z-spectrum, (SZ/2+1) complex bins per Z-real column and channel: 8*B*J*K*X*Y bytes instead of 4*B*J*N
    alg_bytes = (4.0 * B * V * J * h * w + 8.0 * B * J * (sz // 2 + 1) * cube[0] * cube[1]) if step_zd else alg_bytes_8d
    achieved = alg_bytes / (t_k * 1e-3) / 1e9
    path = t_k + (t_pack if planar_input else 0.0)
    out = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "kernel": ("sp3d::unproject_brick_kernel<16,false,float,float,true> (unprojection + z-DFT of the opening conv)" if step_zd else
                   "sp3d::unproject_brick_kernel<16,true,float,float>" if step_cl
                   else "sp3d::unproject_pipe_kernel<16,true,1,false,float,float>"), "kernel_us": round(t_k * 1e3, 2),
        "kernel_result": ("(B,15,15,20,20,16) complex z-spectrum of the cubes in 4x4 tiles, read by the opening conv's x,y pass; "
                          "the cubes themselves never reach HBM") if step_zd else
                         ("channels-last (B,80,80,20,16) cubes read by the opening conv's z-DFT pass" if step_cl
                          else "planar (B,15,80,80,20) cubes"),
        "algorithmic_bytes": int(alg_bytes),
        "algorithmic_bytes_what": ("fused launch: heat-maps once (4*B*V*J*h*w) + z-spectrum once (8*B*J*15*X*Y)" if step_zd else
                                   "SURVEY 8(d): heat-maps once + cubes once"),
        "path_us": round(path * 1e3, 2),
        "path": "unprojection kernel only: the heat-maps arrive as views of the backbone's channels-last buffer"
                if not planar_input else "re-tiling pass (pack_nhwc_kernel<16>) + unprojection kernel: planar hand-over",
        "path_frac": round(alg_bytes / (path * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "path_us_planar_handover": round((t_k + t_pack) * 1e3, 2),
        "kernel_us_strided_result": None if t_strided is None else round(t_strided * 1e3, 2),
        "other_kernels_us": {"pack_nhwc_kernel<16>": round(t_pack * 1e3, 2),
                             ("unproject_pipe_kernel<16,true,1> (planar result)" if step_cl else
                              "unproject_brick_kernel<16,true> (channels-last result)"):
                                 round((t_lin if step_cl else t_brick) * 1e3, 2),
                             "unproject_planar_kernel<16>": round(t_planar * 1e3, 2)},
        "timing": f"HIP events on the launch stream, {iters} back-to-back launches, inputs L2/MALL-warm",
    }
    if step_zd:
        two = t_brick + t_zdft
        out["unprojection_only"] = {
            "what": "SURVEY 8(d)'s accounting (heat-maps once + cubes once) for continuity with rounds 1-5: the un-fused kernel "
                    "(SP3D_FUSE_ZDFT=0 runs it in the step), and the fused launch charged with the same bytes",
            "algorithmic_bytes": int(alg_bytes_8d),
            "unfused_kernel": "sp3d::unproject_brick_kernel<16,true,float,float>", "unfused_kernel_us": round(t_brick * 1e3, 2),
            "unfused_frac": round(alg_bytes_8d / (t_brick * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "fused_frac_on_these_bytes": round(alg_bytes_8d / (t_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        out["fusion"] = {"replaces_us": {"unproject_brick_kernel<16,true>": round(t_brick * 1e3, 2),
                                         "zdft_fwd_cl_kernel<20,28,16>": round(t_zdft * 1e3, 2), "sum": round(two * 1e3, 2)},
                         "fused_us": round(t_k * 1e3, 2),
                         "bytes_the_two_kernels_move": int(alg_bytes_8d + 4.0 * B * 16 * N + 8.0 * B * J * (sz // 2 + 1) * 88 * 88),
                         "what": "stand-alone, warm; in the step: roofline.in_step_graph_stamps (fused) vs profiles/r05_bench_kernel_stats_final.md "
                                 "(34.2 + 22.4 us)"}
    # PMC record of the kernel the step runs (tools/profile_round.sh): the brick kernel's or the planar-result kernel's
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic_fused.json" if step_zd else ("pmc_traffic.json" if step_cl else "pmc_traffic_planar.json"))
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            if rec.get("workload") == f"B{B}_V{V}_J{J}_{w}x{h}_{cube[0]}x{cube[1]}x{cube[2]}" and \
                    ("brick" in rec.get("kernel", "")) == step_cl:
                out["traffic"] = rec.get("hbm_bytes_per_launch")
                out["traffic_source"] = rec.get("source")
        except Exception:
            pass
    # the kernel as it runs INSIDE the step: behind ~1.5 ms of V2V that has pushed the (static) heat-maps out of the L2s
    # and the Infinity Cache.  Emulated here by a 512 MiB fill between launches, the kernel alone between two events
    # per iteration (the rocprofv3 in-graph average of the same kernel is in profiles/, see frac_in_step_source)
    if cold:
        thrash = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)
        run_k = k_fused if step_zd else (k_brick_cl if step_cl else k_lin_planar)
        n_cc = max(20, iters // 6)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_cc)]
        for _ in range(3):
            thrash.fill_(1.0); run_k()
        for e0, e1 in evs:
            thrash.fill_(2.0)
            e0.record(); run_k(); e1.record()
        torch.cuda.synchronize(dev)
        t_in = float(np.median([e0.elapsed_time(e1) for e0, e1 in evs]))
        # ... and as it runs in PRODUCTION: the 2-D backbone's last layer has just written the heat-maps (here: a copy
        # of the same 36.9 MB into the buffer the kernel reads, behind the same 512 MiB fill) - they sit in the
        # Infinity Cache / L2 write path, which the static inputs of a benchmark loop do not
        staging = packed.clone()
        for e0, e1 in evs:
            thrash.fill_(3.0)
            packed.copy_(staging)
            e0.record(); run_k(); e1.record()
        torch.cuda.synchronize(dev)
        t_prod = float(np.median([e0.elapsed_time(e1) for e0, e1 in evs]))
        del thrash, staging
        out["kernel_us_behind_producer"] = round(t_prod * 1e3, 2)
        out["frac_behind_producer"] = round(alg_bytes / (t_prod * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["kernel_us_in_step"] = round(t_in * 1e3, 2)
        out["frac_in_step"] = round(alg_bytes / (t_in * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["frac_in_step_source"] = ("HIP events around the kernel alone, launched behind a 512 MiB fill (cold L2 / Infinity "
                                      f"Cache), median of {n_cc}; rocprofv3 in-graph average: profiles/r06_bench_kernel_stats.md")
    if cold:
        nset = 8
        sets = [_lib.pack_heatmaps([torch.rand_like(x) for x in planar], jp=16) for _ in range(nset)]
        state = {"i": 0}

        def k_cold():
            p = sets[state["i"] % nset]
            state["i"] += 1
            if step_zd:
                _lib.unproject_fwd_zdft([p[c] for c in range(V)], 16, cam, centers, valid, B, J, h, w, cube, gs, img, sz)
                return
            _lib.unproject_fwd([p[c] for c in range(V)], _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16 if step_cl else J,
                               h, w, cube, gs, img, False, channels_last=step_cl)
        t_cold = event_time_ms(k_cold, iters, dev)
        out["kernel_us_rotating_inputs"] = round(t_cold * 1e3, 2)
        out["frac_rotating_inputs"] = round(alg_bytes / (t_cold * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return out


def cpu_baseline_leg(cfg, meta, hms, model, reps):
    """CPU port of the same step on the host cores of this box: C oracle (OpenMP over voxels) for the unprojection
    and NMS + the V2V stack on torch-CPU, once on ALL cores (the headline `value`) and once pinned to 1 thread.
    Bounded sample: `reps` batches of the bench workload per setting.  This is the checker's code timed as a baseline -
    it is never on the product path."""
    from oracle import oracle
    from selfpose3d_amd.camera_pack import pack_cameras
    B = hms[0].shape[0]
    pl = model.project_layer
    cam = pack_cameras(meta, B, pl.img_size)
    centers = np.repeat(np.asarray([model.grid_center], np.float32), B, 0)
    valid = np.ones(B, np.uint8)
    hnp = [x.contiguous().cpu().numpy() for x in hms]
    import copy
    v2v = copy.deepcopy(model.v2v_net).cpu().eval().to(memory_format=torch.contiguous_format)
    nthr = torch.get_num_threads()
    host = os.cpu_count() or 1
    # torch-CPU convolutions stop scaling (and collapse: 0.2 samples/s on 256 threads vs 2.2 on one) far below the
    # 256 hardware threads of the GPU boxes: the multi-thread run uses at most 32
    cores = min(host, 32)
    runs = {}
    for threads, n in ((cores, reps), (1, max(1, reps // 3))):
        torch.set_num_threads(threads)
        oracle.set_threads(threads)
        t0 = time.perf_counter()
        for _ in range(n):
            cubes, _ = oracle.unproject_fwd(hnp, cam, centers, valid, model.grid_size, model.cube_size, pl.img_size,
                                            want_grids=False)
            with torch.no_grad():
                root = v2v(torch.from_numpy(cubes)).squeeze(1).numpy()
            oracle.nms_topk(np.ascontiguousarray(root), 10)
        dt = time.perf_counter() - t0
        runs[threads] = (n * B / dt, dt, n)
    torch.set_num_threads(nthr)
    oracle.set_threads(cores)
    return {"value": round(runs[cores][0], 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{runs[cores][2]} batches of {B} frames of the bench workload on {cores} threads ({runs[cores][1]:.1f} s) "
                      f"and {runs[1][2]} on 1 thread ({runs[1][1]:.1f} s): oracle/sp3d_oracle.c unprojection + NMS "
                      f"(C, OpenMP over voxels) and torch-CPU V2V",
            "value_1_thread": round(runs[1][0], 4), "host_cpus": host}


def pose_stage_leg(dev, proposals_per_frame=4):
    """BASELINE configs[2]/[4] inference side: PoseRegressionNet.forward_batched on B*K person proposals of the
    bench rig - one indexed unprojection launch into 64^3 cubes, V2V in chunks of 8 cubes, fused soft-argmax"""
    from selfpose3d_amd import _lib, synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.pose_regression_net import PoseRegressionNet
    from selfpose3d_amd.project_layer import nhwc_heatmap_views
    B, K = 4, proposals_per_frame
    cfg = load_config(None)
    V, J = int(cfg.DATASET.CAMERA_NUM), int(cfg.NETWORK.NUM_JOINTS)
    w, h = cfg.NETWORK.HEATMAP_SIZE
    meta = syn.make_meta(B, V, cfg.NETWORK.IMAGE_SIZE)
    hms = nhwc_heatmap_views(_lib.pack_heatmaps([x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=5)], jp=16), J)
    net = PoseRegressionNet(cfg)
    syn.fill_parameters_deterministic(net, seed=73, scale=0.05)
    net.eval().to(dev).use_channels_last(True)
    g = torch.Generator().manual_seed(3)
    gc = torch.zeros(B, K, 5)
    gc[..., 0] = (torch.rand(B, K, generator=g) - 0.5) * 4000
    gc[..., 1] = (torch.rand(B, K, generator=g) - 0.5) * 4000
    gc[..., 2] = 800 + torch.rand(B, K, generator=g) * 400
    gc = gc.to(dev)

    def run():
        with torch.no_grad():
            return net.forward_batched(hms, meta, gc)
    for _ in range(3):
        run()
    torch.cuda.synchronize(dev)
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / n * 1e3
    rec = {"what": "PoseRegressionNet.forward_batched, 5 views 240x128 -> 64^3 cubes, eager launches",
           "frames": B, "proposals": B * K, "ms_per_batch": round(ms, 3), "ms_per_person": round(ms / (B * K), 4),
           "persons_per_s": round(B * K / ms * 1e3, 1)}
    del net, hms
    try:
        rec["output_check"] = pose_stage_check(dev)
    except Exception as e:
        rec["output_check"] = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    return rec


def pose_stage_check(dev, v2v_tol=8e-6, joint_tol_mm=0.6):
    """the same code path (forward_batched: indexed unprojection launch, chunked inference-plan V2V at 64^3, fused
    soft-argmax) on the inputs of the reference golden tests/golden/posenet_full.npz - reference PoseRegressionNet at this
    very size (5 views, 240x128, J = 15, 64^3 cubes; 3 valid proposals + 1 invalid) - against the reference's V2V output
    (relative to its range) and joints; bounds = the GPU tests' (tests/test_gpu_reference_pins_r4.py, ~3x measured)"""
    from tests import golden_io as gio
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.pose_regression_net import PoseRegressionNet
    g = gio.load("posenet_full")
    J = int(g["J"])
    cfg = load_config(None, NETWORK__IMAGE_SIZE=[int(v) for v in g["img"]], NETWORK__HEATMAP_SIZE=[int(v) for v in g["hm"]],
                      NETWORK__NUM_JOINTS=J, PICT_STRUCT__CUBE_SIZE=[int(v) for v in g["fine_cube"]])
    hms, meta, gc = gio.posenet_full_inputs()
    net = PoseRegressionNet(cfg)
    syn.fill_parameters_deterministic(net, seed=int(g["pose_seed"]), scale=float(g["param_scale"]))
    net.eval().to(dev).use_channels_last(True)
    ys = []
    net.v2v_net.register_forward_hook(lambda m, i, o: ys.append(o))
    with torch.no_grad():
        pred = net.forward_batched([h.to(dev) for h in hms], meta, gc.to(dev), max_cubes_per_call=8)
    pairs = torch.nonzero(gc[:, :, 3] >= 0).numpy()                   # launch order of the valid (sample, slot) pairs
    y = ys[0][:len(pairs)]
    N = y[0, 0].numel()
    sub = torch.from_numpy(g["sub_idx"]).to(dev)
    ref = np.transpose(g["preds"], (1, 0, 2, 3))
    v2v_err = joint_err = 0.0
    for i, (b, k) in enumerate(pairs):
        row = int((g["grid_centers"][:b, k, 3] >= 0).sum())
        rng = float(max(abs(g[f"v2v_min_{k}"].min()), abs(g[f"v2v_max_{k}"].max())))
        got = y[i].reshape(J, N)[:, sub].float().cpu().numpy()
        v2v_err = max(v2v_err, float(np.abs(got - g[f"v2v_sub_{k}"][row]).max()) / rng)
        joint_err = max(joint_err, float(np.abs(pred[b, k].cpu().numpy() - ref[b, k]).max()))
    zeros_ok = int(torch.count_nonzero(pred.cpu()[gc[:, :, 3] < 0])) == 0
    return {"ok": bool(v2v_err <= v2v_tol and joint_err <= joint_tol_mm and zeros_ok),
            "reference": "tests/golden/posenet_full.npz (reference PoseRegressionNet, CPU fp32, 64^3 cubes, J=15, 240x128)",
            "v2v_output_max_err_rel_range": v2v_err, "joints_max_abs_err_mm": joint_err, "proposals_checked": int(len(pairs)),
            "invalid_proposal_is_zero": zeros_ok, "bounds": {"v2v_rel": v2v_tol, "joints_mm": joint_tol_mm}}


def unprojection_grids_leg(dev, iters=100):
    """The unprojection kernel alone on the grids of BASELINE configs[3] (10 views -> 160x160x40, B=1), configs[2] (ten
    64^3 person cubes, 5 views) and configs[4] (3- and 4-view rigs, 64^3 cubes, bf16 heat-maps and cubes): HIP-event time
    and fraction of the 8 TB/s HBM roofline at SURVEY 8(d)'s algorithmic bytes (2 bytes per element for bf16 tensors)."""
    from selfpose3d_amd import _lib, synthetic as syn
    from selfpose3d_amd.camera_pack import pack_cameras
    img, (w, h), J = (960, 512), (240, 128), 15
    cases = {"configs3_b1_v10_160x160x40": dict(B=1, V=10, cube=(160, 160, 40), gs=syn.SPACE_SIZE, fine=False, bf16=False),
             "configs2_ten_64cubes_v5": dict(B=10, V=5, cube=syn.FINE_CUBE_SIZE, gs=syn.FINE_GRID_SIZE, fine=True, bf16=False),
             "configs4_ten_64cubes_v4_bf16": dict(B=10, V=4, cube=syn.FINE_CUBE_SIZE, gs=syn.FINE_GRID_SIZE, fine=True, bf16=True),
             "configs4_ten_64cubes_v3_bf16": dict(B=10, V=3, cube=syn.FINE_CUBE_SIZE, gs=syn.FINE_GRID_SIZE, fine=True, bf16=True)}
    out = {}
    for name, c in cases.items():
        B, V, cube, gs = c["B"], c["V"], c["cube"], c["gs"]
        N = cube[0] * cube[1] * cube[2]
        meta = syn.make_meta(B, V, img)
        cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
        if c["fine"]:
            rng = np.random.default_rng(0)
            ctr = np.stack([rng.uniform(-1500, 1500, B), rng.uniform(-2000, 1000, B), rng.uniform(700, 1100, B)], 1)
            centers = torch.from_numpy(ctr.astype(np.float32)).to(dev)
        else:
            centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
        valid = torch.ones(B, dtype=torch.uint8, device=dev)
        hms = [x.to(dev) for x in syn.random_heatmaps(B, V, J, h, w, seed=7)]
        if c["bf16"]:
            packed = _lib.pack_heatmaps(hms, jp=16, out_dtype=torch.bfloat16)
            views = [packed[i] for i in range(V)]
            fn = lambda: _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16, h, w, cube, gs, img, False,
                                            channels_last=True, out_dtype=torch.bfloat16)
            fn_planar = lambda: _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, J, h, w, cube, gs, img,
                                                   False, out_dtype=torch.bfloat16)
            alg = 2.0 * B * (V * J * h * w + J * N)
            what = ("bf16 heat-maps and cubes, fp32 arithmetic, channels-last result (as the fp32 rows); two lanes per 32-byte "
                    "pixel (unproject_brick_h_kernel)")
        else:
            packed = _lib.pack_heatmaps(hms, jp=16)
            views = [packed[i] for i in range(V)]
            fn = lambda: _lib.unproject_fwd(views, _lib.LAYOUT_NHWC, 16, cam, centers, valid, B, 16, h, w, cube, gs, img, False,
                                            channels_last=True)
            alg = 4.0 * B * (V * J * h * w + J * N)
            what = "fp32, channels-last result (what the V2V stack reads)"
        # the clocks take ~20 ms of load to settle after an idle gap (66.7 -> 57 us over the first 300 launches of the
        # 160x160x40 kernel): 200 untimed launches, then the median of three timed hundreds
        event_time_ms(fn, 200, dev)
        t = float(np.median([event_time_ms(fn, iters, dev) for _ in range(3)]))
        out[name] = {"kernel_us": round(t * 1e3, 2), "algorithmic_bytes": int(alg),
                     "achieved_GBps": round(alg / (t * 1e-3) / 1e9, 1), "frac": round(alg / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "what": what}
        if c["bf16"]:
            event_time_ms(fn_planar, 100, dev)
            out[name]["kernel_us_planar_result"] = round(float(np.median([event_time_ms(fn_planar, iters, dev) for _ in range(3)])) * 1e3, 2)
        del packed, views, hms
    try:
        out["output_check"] = unprojection_grids_check(dev)
    except Exception as e:
        out["output_check"] = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    return out


ATOMIC_SEGMENT_RATE_G = 20.7      # memory atomics: G (instruction, 64-byte segment) pairs per second, profiles/r04_backward_kernels.md


def unprojection_backward_leg(dev, iters=30):
    """The backward scatter of the unprojection (autograd of lib/models/project_layer.py:93-99 w.r.t. the heat-maps) on the
    grids of the train step (BASELINE configs[2]): root grid 80x80x20 of B = 4 frames (per-tap scatter) and four 64^3 person
    cubes of 2 frames (block merge in LDS), fp32 atomics and the deterministic 64-bit fixed-point form.  HIP events around
    the binding's call (zero-fill of the gradient buffer included).  Two yardsticks: SURVEY 8(d)'s algorithmic bytes against
    8 TB/s, and the memory-atomic request rate measured on this chip (a scatter's own roofline: one slot per instruction and
    64-byte segment).  Check: the two scatter kernels agree bit for bit in the deterministic form on the person cubes, and
    the fp32 form is within 2e-6 of it."""
    from selfpose3d_amd import _lib, synthetic as syn
    from selfpose3d_amd.camera_pack import pack_cameras
    img, (w, h), J = (960, 512), (240, 128), 15
    out = {}
    for name, B, cube, gs, fine in (("root_grid_b4", 4, syn.INITIAL_CUBE_SIZE, syn.SPACE_SIZE, False),
                                    ("four_64cubes_b2", 2, syn.FINE_CUBE_SIZE, syn.FINE_GRID_SIZE, True)):
        V = 5
        N = cube[0] * cube[1] * cube[2]
        meta = syn.make_meta(B, V, img)
        cam = torch.from_numpy(pack_cameras(meta, B, img)).to(dev)
        hms = [x.to(dev) for x in syn.people_heatmaps(B, V, J, h, w, img, seed=3)[0]]
        if fine:
            P = 4
            rng = np.random.default_rng(0)
            c = np.stack([rng.uniform(-1500, 1500, P), rng.uniform(-2000, 1000, P), rng.uniform(700, 1100, P)], 1).astype(np.float32)
            centers = torch.from_numpy(c).to(dev)
            sample_of = torch.tensor([0, 0, 1, 1], dtype=torch.int32, device=dev)
        else:
            P = B
            centers = torch.tensor([syn.SPACE_CENTER] * B, dtype=torch.float32, device=dev)
            sample_of = None
        valid = torch.ones(P, dtype=torch.uint8, device=dev)
        g = torch.randn(P, J, *cube, device=dev)
        packed = _lib.pack_heatmaps(hms, jp=16)
        mask = torch.empty((P, N), dtype=torch.int16, device=dev)
        _lib.unproject_fwd([packed[i] for i in range(V)], _lib.LAYOUT_NHWC, 16, cam, centers, valid, P, J, h, w, cube, gs, img,
                           False, sample_of=sample_of, pass_mask=mask)

        def run(det=False, scatter=_lib.SCATTER_AUTO):
            return _lib.unproject_bwd_packed(cam, centers, valid, g, mask, B, V, J, 16, h, w, cube, gs, img, sample_of=sample_of,
                                             deterministic=det, return_packed=True, scatter=scatter)
        t = float(np.median([event_time_ms(run, iters, dev) for _ in range(3)]))
        td = float(np.median([event_time_ms(lambda: run(True), iters, dev) for _ in range(3)]))
        alg = 4.0 * (P * J * N + 2 * V * B * J * h * w)           # gradient cubes read + gradient maps read-modify-write
        rec = {"us": round(t * 1e3, 1), "deterministic_us": round(td * 1e3, 1), "algorithmic_bytes": int(alg),
               "frac_hbm": round(alg / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "kernel": "unproject_bwd3_kernel (8x8x4 voxel blocks merged in LDS, 64-bit fixed point)" if fine else
                         "unproject_bwd2_kernel (one memory atomic per tap and 64-byte pixel)"}
        out[name] = rec
        if fine:
            d2 = run(True, _lib.SCATTER_PER_TAP)
            t2 = float(np.median([event_time_ms(lambda: run(False, _lib.SCATTER_PER_TAP), iters, dev) for _ in range(3)]))
            d3, f3 = run(True), run(False)
            scale = float(d3.abs().max())
            rec["per_tap_kernel_us"] = round(t2 * 1e3, 1)
            rec["output_check"] = {"ok": bool(torch.equal(d2, d3)) and float((f3 - d3).abs().max()) <= 2e-6 * scale,
                                   "deterministic_merge_equals_per_tap_bitwise": bool(torch.equal(d2, d3)),
                                   "fp32_vs_fixed_point_max_rel": float((f3 - d3).abs().max()) / max(scale, 1e-30)}
    out["request_rate_model"] = {"G_segment_adds_per_s": ATOMIC_SEGMENT_RATE_G,
                                 "what": ("memory atomics retire at ~20.7 G (instruction, 64-byte segment) pairs per second on MI355X "
                                          "whatever type, scope or pattern (tools/global_atomic_bench.hip); the per-tap kernel on the root "
                                          "grid runs at that rate (5.7 M segments), the merge kernel leaves 3.06 M segments for the person "
                                          "cubes instead of ~19 M (TCC_ATOMIC, profiles/r04_pmc_bwd_fine.json)")}
    return out


def unprojection_grids_check(dev, tol=1e-6):
    """the kernels this leg times, on the inputs of the reference goldens of the same grids (reference ProjectLayer.get_voxel,
    tests/golden/make_goldens.py): configs[3] = unproj_stress_v10 (10 views -> 160x160x40), configs[2]/[4] cube =
    unproj_fine_full_240x128 (5 views -> 64^3), channels-last fp32 result; and the bf16-storage kernel against the same golden
    within bf16 rounding of the stored maps and cubes (2^-8 relative twice)"""
    from tests import golden_io as gio
    from selfpose3d_amd import _lib
    res = {}
    ok = True
    for name in ("unproj_stress_v10", "unproj_fine_full_240x128"):
        case = gio.Case(name)
        w, h = case.hm
        hms = [x.to(dev) for x in case.hms]
        cam = torch.from_numpy(case.cam).to(dev)
        centers = torch.from_numpy(case.centers).to(dev)
        valid = torch.from_numpy(case.valid).to(dev)
        exp_c, _, idx = case.expected()
        packed = _lib.pack_heatmaps(hms, jp=16)
        cl, _ = _lib.unproject_fwd([packed[c] for c in range(case.V)], _lib.LAYOUT_NHWC, 16, cam, centers, valid, case.B, 16,
                                   h, w, case.cube, case.grid_size, case.img, False, channels_last=True)
        got = cl[:, :case.J].reshape(case.B, case.J, case.N).float().cpu().numpy()
        err = float(np.abs((got[:, :, idx] if idx is not None else got) - exp_c).max())
        p16 = _lib.pack_heatmaps(hms, jp=16, out_dtype=torch.bfloat16)
        c16, _ = _lib.unproject_fwd([p16[c] for c in range(case.V)], _lib.LAYOUT_NHWC, 16, cam, centers, valid, case.B, case.J,
                                    h, w, case.cube, case.grid_size, case.img, False, out_dtype=torch.bfloat16)
        got16 = c16.reshape(case.B, case.J, case.N).float().cpu().numpy()
        err16 = float(np.abs((got16[:, :, idx] if idx is not None else got16) - exp_c).max())
        res[name] = {"fp32_max_abs_err": err, "bf16_storage_max_abs_err": err16}
        ok = ok and err <= tol and err16 <= 1.0 / 64.0            # values in [0,1]: two bf16 roundings of <= 2^-8 each, 4 taps
    return {"ok": bool(ok), "reference": "tests/golden/unproj_stress_v10.npz, unproj_fine_full_240x128.npz (reference ProjectLayer)",
            "cases": res, "bounds": {"fp32": tol, "bf16_storage": 1.0 / 64.0}}


def train_step_leg(args, rank, world, dev):
    """BASELINE configs[2]: full train step (ResNet-50 backbone on 5 x 960x512 views, root net 80x80x20 + its loss, pose
    net on 64^3 cubes, Adam), batch 2 per GPU, one process per GPU, DDP gradient all-reduce over RCCL when world > 1.
    Frames are synthetic and built ONCE per rank, resident on the device (no loader inside the timed region); the root net
    is randomly initialised, so the frame's ground-truth roots stand in for its proposals (a wrapper around root_net.forward,
    below - the metric's name says so) and the pose net runs once per person, as it does for a trained model."""
    from torch.utils.data import default_collate
    from selfpose3d_amd import distributed as D
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.multi_person_posenet import get_multi_person_pose_net
    from selfpose3d_amd.synthetic_dataset import SyntheticPanoptic
    cfg = load_config(os.path.join(ROOT, "configs", "panoptic_synthetic_960x512_cam5.yaml"))
    Bt = int(cfg.TRAIN.BATCH_SIZE)
    # MIOpen's immediate mode (benchmark off) picks kernels for the 3-D backward convolutions that are ~20x slower than the
    # ones its search finds (2.2 s vs ~0.1 s per step, measured): the leg pays the one-off search in its warm-up steps
    # (minutes on a box with an empty MIOpen user db; --train-find immediate skips it and says so in the record)
    torch.backends.cudnn.benchmark = args.train_find == "search"
    torch.manual_seed(D.rank_seed(0, rank))
    model = get_multi_person_pose_net(cfg, is_train=True).to(dev)
    model.use_channels_last(True)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=float(cfg.TRAIN.LR))
    find_unused = D.needs_find_unused(cfg)
    ds = SyntheticPanoptic(cfg, num_frames=Bt, seed=D.rank_seed(1, rank) % 100000, max_people=3)
    inputs, t2d, w2d, t3d, meta, _ = default_collate([ds[i] for i in range(Bt)])
    inputs = [x.to(dev).contiguous(memory_format=torch.channels_last) for x in inputs]
    t2d, w2d, t3d0 = [x.to(dev) for x in t2d], [x.to(dev) for x in w2d], t3d[0].to(dev)
    ddp = D.wrap_ddp(model, dev, find_unused=find_unused)

    def gt_proposals(grid_centers, m):
        gc = torch.zeros_like(grid_centers)
        gc[:, :, 3] = -1.0
        roots, nper = m[0]["roots_3d"].float().to(gc.device), m[0]["num_person"]
        for i in range(gc.shape[0]):
            n = int(nper[i])
            gc[i, :n, :3] = roots[i, :n]
            gc[i, :n, 3] = torch.arange(n, device=gc.device, dtype=torch.float32)
            gc[i, :n, 4] = 1.0
        return gc
    # the substitution lives HERE, not in the model: the root net still runs (its cubes feed loss_3d and its backward), only
    # the proposal table it hands on is replaced
    root_forward = model.root_net.forward

    def root_forward_gt(all_heatmaps, m, *a, **k):
        root_cubes, gc = root_forward(all_heatmaps, m, *a, **k)
        return root_cubes, gt_proposals(gc, m)
    model.root_net.forward = root_forward_gt
    ddp.train()
    state = {}

    def step():
        _, _, _, l2d, l3d, lcord = ddp(views=inputs, meta=meta, targets_2d=t2d, weights_2d=w2d, targets_3d=t3d0)
        loss = l2d.mean() + l3d.mean() + lcord.mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        state["loss"] = loss
        return loss
    el, _ = D.timed_steps(step, args.train_steps, args.train_warmup, dev)
    host = None
    if "host_contention" in args.legs_list and world == 1:
        try:
            host = D.host_contention(step, args.train_steps, dev)
        except Exception as e:
            host = {"error": f"{type(e).__name__}: {e}"}
    nbytes = int(sum(p.numel() for p in params) * 4)
    comm = None
    if world > 1:
        # how much of the gradient all-reduce does the step hide?  (a) the buckets reduced with nothing else running,
        # (b) the same step with the all-reduce switched off (DDP.no_sync: local gradients; LAST use of this model - the
        # ranks' parameters drift apart from here).  exposed = step - (b); overlap = 1 - exposed / (a).
        try:
            alone = D.allreduce_alone_ms(nbytes, dev)

            def step_local():
                with ddp.no_sync():
                    return step()
            el_l, _ = D.timed_steps(step_local, args.train_steps, 1, dev)
            ms, ms_l = 1e3 * el / args.train_steps, 1e3 * el_l / args.train_steps
            exposed = max(0.0, ms - ms_l)
            comm = {"allreduce_ms": round(alone, 3), "ms_per_step_without_allreduce": round(ms_l, 2),
                    "exposed_allreduce_ms": round(exposed, 3),
                    "overlap": round(min(1.0, max(0.0, 1.0 - exposed / alone)), 3) if alone > 0 else None,
                    "bus_GBps_alone": round(2 * (world - 1) / world * nbytes / (alone * 1e-3) / 1e9, 2),
                    "what": "allreduce_ms: all gradient buckets (32 MB each) reduced back to back with no compute running; "
                            "ms_per_step_without_allreduce: the same step under DDP.no_sync(); exposed = difference of the two steps"}
        except Exception as e:
            comm = {"error": f"{type(e).__name__}: {e}"}
    V = len(inputs)
    persons = int(sum(int(n) for n in meta[0]["num_person"]))
    slots_in_use = int(max(int(n) for n in meta[0]["num_person"]))       # the reference: one pose-net call per slot in use
    pose_calls = 1 if type(model).batch_slots_in_training else slots_in_use  # round 5: all slots in one pass (grouped BatchNorm)
    return {"metric": "multi-view frames/sec, full train step (BASELINE configs[2]), proposals = ground-truth roots",
            "value": round(D.job_throughput(Bt, args.train_steps, el, world), 3), "unit": "frames/s", "n_gpus": world,
            "ms_per_step": round(1e3 * el / args.train_steps, 2), "steps": args.train_steps, "warmup": args.train_warmup,
            "batch_per_gpu": Bt, "views": V, "scaling": "weak", "dtype": "f32",
            "collective": ((("DDP gradient all-reduce over gloo, all ranks on cuda:0 (--share-gpu smoke)" if args.share_gpu else
                             "DDP gradient all-reduce over RCCL (backend nccl)") + ", bucket_cap 32 MB, overlapped with backward")
                           if world > 1 else "none (single process)"),
            "host_contention": host,
            "allreduce_bytes_per_step": nbytes if world > 1 else 0, "gradient_bytes": nbytes,
            "allreduce_ms": None if comm is None else comm.get("allreduce_ms"), "allreduce_overlap": comm,
            "find_unused_parameters": bool(find_unused), "miopen_selection": args.train_find, "pose_net_calls_per_step": pose_calls, "candidate_slots_in_use": slots_in_use, "person_cubes_per_step": persons,
            "backbone_pass": "all views in one channels_last pass, per-view BatchNorm statistics (grouped kernels)",
            "loss_last": float(state["loss"].detach()), "data": "synthetic frames built once per rank, resident on the device",
            "config": "configs/panoptic_synthetic_960x512_cam5.yaml (ResNet-50, 80x80x20 root grid, 64^3 pose cubes)"}


def train_step_ssv_leg(args, rank, world, dev):
    """Opt-in leg (--legs ...,train_step_ssv): the SELF-SUPERVISED pose-net stage at full size - what SelfPose3d itself trains
    (reference configs/panoptic_ssl/resnet50/cam5_posenet.yaml; lib/models/multi_person_posenet_ssv.py:197-501): three view
    sets of 5 x 960x512 through the ResNet-50 backbone, two through the ResNet-18 attention net, frozen root net on set 3,
    pose net on both augmented sets for every proposal, re-projection + rendering losses, Adam.  Batch 2 per GPU
    (BASELINE configs[2]).  As in the supervised leg the randomly initialised root net cannot propose people, so the frame's
    ground-truth roots stand in for its proposals (wrapper around root_net.forward, here in bench.py)."""
    from torch.utils.data import default_collate
    from selfpose3d_amd import distributed as D
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.models import get_multi_person_pose_net
    from selfpose3d_amd.synthetic_dataset import SyntheticPanopticSSV
    cfg = load_config(os.path.join(ROOT, "configs", "cam5_posenet.yaml"), TRAIN__BATCH_SIZE=2)
    Bt = int(cfg.TRAIN.BATCH_SIZE)
    torch.backends.cudnn.benchmark = args.train_find == "search"
    torch.manual_seed(D.rank_seed(0, rank))
    model = get_multi_person_pose_net(cfg, is_train=True).to(dev)
    model.use_channels_last(True)
    for p in model.root_net.parameters():                        # FREEZE_ROOTNET (tools/train_3d.py select_trainable)
        p.requires_grad_(False)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=float(cfg.TRAIN.LR))
    ds = SyntheticPanopticSSV(cfg, num_frames=Bt, seed=D.rank_seed(1, rank) % 100000, max_people=3)
    batch = default_collate([ds[i] for i in range(Bt)])
    (in1, t1, w1, d1, m1, _, in2, t2, w2, d2, m2, _, in3, t3, w3, d3, m3, _) = batch
    cl = lambda v: [x.to(dev).contiguous(memory_format=torch.channels_last) for x in v]
    in1, in2, in3 = cl(in1), cl(in2), cl(in3)
    dv = lambda v: [x.to(dev) for x in v]
    t1, t2, t3, w1, w2, w3 = dv(t1), dv(t2), dv(t3), dv(w1), dv(w2), dv(w3)
    root_forward = model.root_net.forward

    def root_forward_gt(all_heatmaps, m, *a, **k):
        out = root_forward(all_heatmaps, m, *a, **k)
        gc = torch.zeros_like(out[3])
        gc[:, :, 3] = -1.0
        roots, nper = m[0]["roots_3d"].float().to(gc.device), m[0]["num_person"]
        for i in range(gc.shape[0]):
            n = int(nper[i])
            gc[i, :n, :3] = roots[i, :n]
            gc[i, :n, 3] = torch.arange(n, device=gc.device, dtype=torch.float32)
            gc[i, :n, 4] = 1.0
        return out[0], out[1], out[2], gc
    model.root_net.forward = root_forward_gt
    ddp = D.wrap_ddp(model, dev, find_unused=D.needs_find_unused(cfg))
    ddp.train()
    model.root_net.eval()
    state = {}

    def step():
        _, _, _, losses = ddp(views1=in1, meta1=m1, targets_2d1=t1, weights_2d1=w1, targets_3d1=d1[0],
                              views2=in2, meta2=m2, targets_2d2=t2, weights_2d2=w2, targets_3d2=d2[0],
                              views3=in3, meta3=m3, targets_2d3=t3, weights_2d3=w3, targets_3d3=d3[0], epoch=int(cfg.TRAIN.L1_EPOCH))
        loss = sum(v.mean() for v in losses.values() if v.requires_grad)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        state["loss"], state["keys"] = loss, sorted(losses)
    el, _ = D.timed_steps(step, args.train_steps, args.train_warmup, dev)
    persons = int(sum(int(n) for n in m3[0]["num_person"]))
    return {"metric": "multi-view frames/sec, full SELF-SUPERVISED train step (pose-net stage), proposals = ground-truth roots",
            "value": round(D.job_throughput(Bt, args.train_steps, el, world), 3), "unit": "frames/s", "n_gpus": world,
            "ms_per_step": round(1e3 * el / args.train_steps, 2), "steps": args.train_steps, "warmup": args.train_warmup,
            "batch_per_gpu": Bt, "views": len(in1), "view_sets": 3, "person_cubes_per_step": 2 * persons, "pose_net_calls_per_step": 1,
            "loss_terms": state["keys"], "loss_last": float(state["loss"].detach()), "dtype": "f32",
            "config": "configs/cam5_posenet.yaml (reference cam5_posenet.yaml hyper-parameters), TRAIN.BATCH_SIZE = 2",
            "data": "synthetic three-set frames built once per rank, resident on the device"}


def cpu_reference_record():
    """the reference's own Python on CPU, timed in the build container by tools/time_reference_cpu.py (the reference
    cannot travel to the GPU box): static record, host described inside"""
    f = os.path.join(ROOT, "profiles", "cpu_reference.json")
    if not os.path.exists(f):
        return None
    rec = json.load(open(f))
    c1 = rec["configs"].get("configs[1] B=4 960x512->240x128", {})
    allc = c1.get(f"threads_{rec['host']['logical_cpus']}", {})
    return {"value": allc.get("frames_per_s"), "unit": "samples/s", "kind": "reference", "cores": rec["host"]["logical_cpus"],
            "value_1_thread": c1.get("threads_1", {}).get("frames_per_s"),
            "sample": "reference CuboidProposalNet.forward (lib/models/cuboid_proposal_net.py:102-122) on configs[1] inputs, "
                      "median of 5 after 2 warm-ups, measured in the build container, NOT on this box",
            "host": rec["host"], "detail": c1, "source": "profiles/cpu_reference.json (tools/time_reference_cpu.py)"}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: re-exec under torch.distributed.run, one rank per
    GPU, rendezvous on 127.0.0.1 - the command line the task statement gives for N > 1.  Never returns.  (Round-3 review:
    a plain `python bench.py --gpus 8` ran ONE rank and printed n_gpus = 1.)"""
    import socket
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not args.share_gpu and n < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: {n} GPU(s) visible on this box (one rank per GPU; "
                         f"--share-gpu runs every rank on cuda:0 over gloo as a smoke test)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: exec %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    if args.share_gpu:
        # ranks that share a GPU load the library flavour without packed-fp32 instructions (libsp3d_nopk.so): the default
        # flavour's unprojection arithmetic is wrong next to another plan's matrix instructions on the same CU
        # (profiles/r05_shared_gpu.md).  Set before the first kernel call; inherited by the ranks self_launch starts.
        os.environ["SP3D_SHARED_GPU"] = "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the line's n_gpus must be the number of ranks "
                         f"that ran (launch with --nproc-per-node {args.gpus}, or plainly and let bench.py launch itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the unprojection path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    from selfpose3d_amd import distributed as D
    if world > 1:
        # control plane (barriers, max over ranks) on gloo, data plane (DDP gradient buckets of the train leg) on RCCL
        # (backend "nccl" == RCCL on ROCm); ranks that share one GPU cannot form an RCCL communicator: gloo for both
        D.init_split("gloo" if args.share_gpu else "nccl")

    miopen_db = use_shipped_miopen_db()        # before the first convolution of the process (a private directory per rank)
    tun = getattr(torch.cuda, "tunable", None)
    if tun is not None:                         # ranks must not race for one tunableop_results<ordinal>.csv in the cwd
        try:
            tun.write_file_on_exit(False)
        except Exception:
            pass
    torch.backends.cudnn.benchmark = True
    cfg, meta, hms, model, golden = build_workload(args.batch, rank, dev, args.v2v_layout, args.front_conv,
                                                   not args.no_winograd, args.planar_input, not args.no_gemm_tuning)
    legs = args.legs.split(",") if args.legs not in ("auto", "none") else \
        ([] if args.legs == "none" else ["pose_stage", "planar_handover", "unprojection_grids", "unprojection_backward", "train_step",
                                         "host_contention"])
    args.legs_list = legs

    from selfpose3d_amd.project_layer import clear_pack_cache

    def eager_step():
        # every step is a FRESH batch: drop the per-batch caches (re-tiled heat-maps, camera table) so the
        # timed region contains the host camera pack + upload, the pack kernel (planar hand-over) and the unprojection
        clear_pack_cache()
        model.project_layer._cam_key = None
        with torch.no_grad():
            return model(hms, meta)

    step, mode = eager_step, "eager"
    if not args.no_graph:
        try:
            for _ in range(2):
                eager_step()                       # MIOpen algorithm search must happen outside capture
            torch.cuda.synchronize(dev)
            from selfpose3d_amd.graphs import GraphedRootNet
            graphed = GraphedRootNet(model, hms, meta, copies=args.graph_copies)
            step, mode = (lambda: graphed()), "hipgraph"
        except Exception as e:                     # capture not possible: stay on the eager HIP path
            print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            step, mode = eager_step, "eager"

    # W untimed warm-up steps, then EXACTLY K steps between two barrier+synchronise points, slowest rank counts
    elapsed, out = D.timed_steps(step, args.steps, args.warmup, dev)
    # box_spread: the same K steps again (same barrier rule), so that the line carries its own noise figure - 20 steps of
    # 1.6 ms differ by 1-3 % from repeat to repeat and box to box, and a leg that does MORE work must not read faster than
    # the headline without that being visible (round-3 review).  `value` stays the first K steps, as the contract says.
    spread = [1e3 * elapsed / args.steps]
    for _ in range(max(0, args.spread_repeats - 1)):
        el_r, _ = D.timed_steps(step, args.steps, 0, dev)
        spread.append(1e3 * el_r / args.steps)

    # ---- the line's headline part exists BEFORE any leg runs: a leg that never returns cannot take it along -----------
    result = None
    if rank == 0:
        B = args.batch
        V, J = len(hms), hms[0].shape[1]
        value = D.job_throughput(B, args.steps, elapsed, world)
        handover = "planar (B,J,h,w) -> pack(HIP) -> " if args.planar_input else "channels-last views of the backbone's buffer -> "
        result = {
            "metric": "multi-view samples/sec (5-view Panoptic, 80x80x20 voxel)",
            "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "panoptic_5view_cuboid_proposal_net_fwd_b4 (BASELINE configs[1])",
                       "batch_per_gpu": B, "global_batch": B * world, "views": V, "joints": J,
                       "heatmap": [int(hms[0].shape[3]), int(hms[0].shape[2])], "image": list(cfg.NETWORK.IMAGE_SIZE),
                       "voxels": list(model.cube_size), "parallelism": f"frames sharded over {world} rank(s), no collective",
                       "process_groups": ("none (single process)" if world == 1 else
                                          "control plane (barriers of the timed regions, max over ranks) on gloo; data plane = DDP "
                                          "gradient buckets of legs.train_step only, on " + ("gloo" if args.share_gpu else "RCCL")),
                       "step": "heat-maps(HBM, " + handover + "unproject(HIP) -> V2V(fp32 in/out, fp32 accumulation: 7^3 opening conv " +
                               ("in the frequency domain (HIP z-DFT + 88x88 plane transforms + contraction)"
                                if args.front_conv == "fft" else "MIOpen direct") +
                               (", 3^3 convs as HIP kernels on the bf16 matrix pipe with exact 3-piece operand splits (direct conv at "
                                "full resolution, fused Winograd F(2,3) at half resolution), Winograd transforms + rocBLAS at quarter "
                                "resolution" if not args.no_winograd else "") +
                               ", other convs GEMM/MIOpen) -> NMS/top-k(HIP)",
                       "conv_arithmetic": "fp32 tensors; 3^3 products = 6 exact bf16 x bf16 partial products per fp32 multiply "
                                          "(hi/mid/lo pieces, dropped terms < 2^-24 relative), fp32 accumulation; tested bound: "
                                          "max error vs a float64 convolution <= 1.5x that of the fp32-MFMA kernel on the same "
                                          "inputs (tests/test_gpu_parity.py); measured 2.3e-6 (split Winograd) / 9.1e-6 (direct split) "
                                          "vs 2.4e-6 (fp32 MFMA) / 9.3e-6 (MIOpen direct fp32) on outputs of magnitude 8",
                       "heatmap_handover": "planar" if args.planar_input else "nhwc16_views",
                       "gemm_selection": ("library heuristics" if args.no_gemm_tuning else
                                          "PyTorch TunableOp for the plan's rocBLAS/hipBLASLt GEMMs, opted in by bench.py "
                                          "(V2VNet.tune_gemms(True)); process-wide flags restored after every forward"),
                       "miopen_user_db": ("private copy of selfpose3d_amd/miopen_db (find results of this file's convolution "
                                          "shapes: shortens warm-ups, same kernels as a fresh search)" if miopen_db else
                                          "the process's own (MIOPEN_USER_DB_PATH / default)"),
                       "weights": "deterministic N(0,0.05) fill (tests/golden/rootnet_full.npz)",
                       "v2v_layout": args.v2v_layout, "front_conv": args.front_conv, "winograd": not args.no_winograd, "launch": mode,
                       "graph_executables": (args.graph_copies if mode == "hipgraph" else 0)},
            "views_x_frames_per_s": round(value * V, 3),
            "value_window": (f"first of {len(spread)} windows of {args.steps} steps; by ms_per_step it ranks "
                             f"{1 + sorted(spread).index(spread[0])} of {len(spread)} (1 = fastest): legs measured later in the run may read "
                             f"faster than `value` by up to the box_spread"),
            "box_spread": {"ms_per_step_min": round(min(spread), 4), "ms_per_step_max": round(max(spread), 4),
                           "repeats": len(spread), "steps_each": args.steps,
                           "what": "the timed K steps (= value) and further repeats of K steps on the same box, same rule"},
        }
        if args.share_gpu:
            result["share_gpu_note"] = ("ranks that share a GPU run libsp3d_nopk.so (SP3D_SHARED_GPU=1: no packed-fp32 instructions), "
                                        "which is immune to the matrix-instruction / packed-fp32 interaction of "
                                        "profiles/r04_gpu_sharing_finding.md; the output check is ENFORCED (profiles/r05_shared_gpu.md)")
            result["config"]["parallelism"] = (f"SMOKE: {world} ranks share cuda:0, process group gloo (no RCCL): exercises the "
                                               f"multi-rank code path, not a scaling number")
            from selfpose3d_amd import _lib as _l
            result["config"]["library"] = os.path.basename(_l.LIB_PATH)
    # ---- extra legs that every rank takes part in (same timing rule); rank 0 adds them to the one JSON line -----------
    extra = {}
    # world > 1: everything below may sit in a collective that never returns (the DDP leg is the first place a multi-GPU
    # RCCL communicator of this repo meets hardware).  Every rank arms the same deadline; when it expires rank 0 prints the
    # line it has (headline measured above + the legs finished so far + what happened) and all ranks leave with status EXIT_LEGS_INCOMPLETE.
    import threading
    line_lock, line_state = threading.Lock(), {"printed": False, "incomplete": None, "rccl_ranks_seen": None}

    def print_line_once(note=None):
        with line_lock:
            if note is not None and line_state["incomplete"] is None:
                line_state["incomplete"] = note
            if rank != 0 or line_state["printed"]:
                return
            if note is not None:
                extra["deadline"] = note
            result.setdefault("legs", extra)
            # top level, where a driver looks: did every leg return, and may this line be one point of a scaling curve?
            # (false: the deadline fired, the final barrier failed, a leg recorded an error, or the data-plane communicator
            # did not span every rank; the process exits with EXIT_LEGS_INCOMPLETE when scaling_valid is false)
            failed = [k for k, v in list(extra.items()) if isinstance(v, dict) and "error" in v]
            result["legs_complete"] = line_state["incomplete"] is None and not failed
            seen = line_state["rccl_ranks_seen"]
            # scaling_valid: nothing hung, the communicator spanned every rank and the legs that USE it returned; an error record
            # of a leg without a collective (rank 0's kernel legs) makes the run incomplete, not the scaling point invalid
            collective_failed = [k for k in failed if k in COLLECTIVE_LEGS]
            result["scaling_valid"] = bool(line_state["incomplete"] is None and not collective_failed and
                                           (world == 1 or seen == world))
            if failed:
                result["legs_failed"] = failed
            for attempt in range(5):                 # the deadline thread may serialise while the main thread adds a leg
                try:
                    text = json.dumps(result)
                    break
                except RuntimeError:
                    if attempt == 4:
                        raise
            print(text, flush=True)
            line_state["printed"] = True
    deadline = D.Deadline(args.leg_deadline if world > 1 else 0.0,
                          lambda: print_line_once(f"legs not finished {args.leg_deadline:.0f} s after the headline measurement: "
                                                  f"line printed by the deadline thread, every rank exits with status "
                                                  f"{EXIT_LEGS_INCOMPLETE} (--leg-deadline)"), exit_code=EXIT_LEGS_INCOMPLETE)
    deadline.__enter__()
    if world > 1:
        # first collective of the data-plane communicator (RCCL unless --share-gpu), under the deadline: one all-reduce of
        # ones.  The line then PROVES how many ranks RCCL spanned; a communicator that cannot be built is an error entry
        # and an incomplete run, not a hang inside the train leg.
        import time as _time
        t_c = _time.perf_counter()
        try:
            seen = D.data_ranks_seen(dev)
            line_state["rccl_ranks_seen"] = seen
            if rank == 0:
                result["config"]["rccl_ranks_seen"] = seen
                result["config"]["data_plane"] = {"backend": "gloo (--share-gpu)" if args.share_gpu else "nccl (RCCL)",
                                                  "ranks_seen": seen, "first_collective_s": round(_time.perf_counter() - t_c, 3)}
        except Exception as e:
            extra["data_plane"] = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                result["config"]["rccl_ranks_seen"] = 0
    if "host_contention" in legs and world == 1:
        # host-side readiness for 8 ranks per node, measured on this 1-GPU box: the headline step (one graph replay + the
        # camera-table pack per step) and, in the train leg, the ~3 500-launch train step, re-timed with this process
        # confined to 1/8 of the host's cores and with the other 7/8 busy
        try:
            extra["host_contention"] = {"headline_step": D.host_contention(step, args.steps, dev),
                                        "what": "this process pinned to 1/8 of the host cores with one compute thread, then "
                                                "the same with 7 neighbour ranks' worth of spinning processes (2 each, pinned "
                                                "to the other shares); host_ms_per_step = CPU time of the Python thread per step; the train "
                                                "step under the same conditions: legs.train_step.host_contention"}
        except Exception as e:
            extra["host_contention"] = {"error": f"{type(e).__name__}: {e}"}
    if "planar_handover" in legs and not args.planar_input:
        # the reference hands the heat-maps over planar, (B,15,h,w) per view: the same step + the re-tiling pass
        try:
            planar = [x.contiguous() for x in hms]

            def planar_step():
                clear_pack_cache()
                model.project_layer._cam_key = None
                with torch.no_grad():
                    return model(planar, meta)
            pstep = planar_step
            if mode == "hipgraph":
                for _ in range(2):
                    planar_step()
                torch.cuda.synchronize(dev)
                from selfpose3d_amd.graphs import GraphedRootNet
                g_pl = GraphedRootNet(model, planar, meta)
                pstep = (lambda: g_pl())
            n_pl = max(10, min(50, args.steps))
            el_pl, _ = D.timed_steps(pstep, n_pl, 5, dev)
            extra["planar_handover"] = {"value": round(D.job_throughput(args.batch, n_pl, el_pl, world), 3), "unit": "samples/s",
                                        "ms_per_step": round(1e3 * el_pl / n_pl, 4), "steps": n_pl,
                                        "what": "same step with the reference's planar (B,15,h,w) hand-over: + pack_nhwc_kernel<16>"}
        except Exception as e:
            extra["planar_handover"] = {"error": f"{type(e).__name__}: {e}"}
    if "train_step" in legs:
        try:
            extra["train_step"] = train_step_leg(args, rank, world, dev)
        except Exception as e:                          # the headline must not die with a leg (all ranks fail alike)
            extra["train_step"] = {"error": f"{type(e).__name__}: {e}"}

    if "train_step_ssv" in legs:                                  # opt-in
        try:
            extra["train_step_ssv"] = train_step_ssv_leg(args, rank, world, dev)
        except Exception as e:
            extra["train_step_ssv"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        B = args.batch
        if golden is not None and not args.no_check:
            result["output_check"] = check_output(out, golden)
        if not args.no_fp32_leg and not args.no_winograd and world == 1:
            # the same step with the 3^3 products on v_mfma_f32_32x32x2_f32 / rocBLAS fp32 only (no bf16 operand splits):
            # what the split-product kernels buy, and a number that involves no bf16 instruction at all
            try:
                model.v2v_net.wino_split = False
                model.v2v_net.invalidate_plan()
                for _ in range(2):
                    eager_step()
                torch.cuda.synchronize(dev)
                alt_step = eager_step
                if mode == "hipgraph":
                    from selfpose3d_amd.graphs import GraphedRootNet
                    g2 = GraphedRootNet(model, hms, meta)
                    alt_step = (lambda: g2())
                n2 = max(10, min(50, args.steps))
                el2, out2 = D.timed_steps(alt_step, n2, 5, dev)
                result["fp32_matrix_instructions_only"] = {
                    "value": round(D.job_throughput(B, n2, el2, 1), 3), "unit": "samples/s", "ms_per_step": round(1e3 * el2 / n2, 4),
                    "steps": n2, "what": "same step, 3^3 convs as fused Winograd on v_mfma_f32_32x32x2_f32 + rocBLAS fp32 GEMMs "
                                         "(V2VNet.wino_split = False)",
                    "root_cubes_max_abs_diff_vs_headline": float((out2[0].float() - out[0].float()).abs().max())}
            except Exception as e:
                result["fp32_matrix_instructions_only"] = {"error": f"{type(e).__name__}: {e}"}
            finally:
                model.v2v_net.wino_split = True
                model.v2v_net.invalidate_plan()
        result["roofline"] = roofline_leg(cfg, meta, hms, model, args.roofline_iters, dev, args.planar_input,
                                          cold=not args.no_cold)
        if mode == "hipgraph":
            # the unprojection's time INSIDE the replayed step, from the step's own graph: a second capture of the same
            # step with two external timing-event nodes around ProjectLayer.get_voxel (the headline graph carries none)
            try:
                from selfpose3d_amd.graphs import GraphedRootNet
                g_t = GraphedRootNet(model, hms, meta, time_unprojection=True)
                ts, gaps = [], []
                for i in range(60):
                    g_t()
                    if i >= 10:
                        a, b = g_t.unprojection_us()
                        ts.append(a)
                        gaps.append(b)
                t_raw, t_gap = float(np.median(ts)), float(np.median(gaps))
                t_med = (t_raw - t_gap) * 1e-3                      # ms
                alg = float(result["roofline"]["algorithmic_bytes"]) if "algorithmic_bytes" in result["roofline"] else None
                rec = {"kernel_us": round(t_med * 1e3, 2), "stamp_to_stamp_us": round(t_raw, 2), "adjacent_stamps_us": round(t_gap, 2),
                       "min_us": round(min(ts) - t_gap, 2), "max_us": round(max(ts) - t_gap, 2), "replays": len(ts),
                       "source": "one-thread kernels writing the chip-wide 100 MHz clock before and after ProjectLayer.get_voxel "
                                 "INSIDE the replayed HIP graph of the step (a second capture of the same step; the headline graph "
                                 "carries none); kernel_us = stamp-to-stamp minus the distance of two adjacent stamps"}
                if alg:
                    rec["frac"] = round(alg / (t_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                result["roofline"]["in_step_graph_stamps"] = rec
                del g_t
            except Exception as e:
                result["roofline"]["in_step_graph_stamps"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_leg(cfg, meta, hms, model, args.cpu_baseline_reps)
        else:
            result["cpu_baseline"] = None
        result["cpu_reference"] = cpu_reference_record()
        if "pose_stage" in legs:
            try:
                extra["pose_stage"] = pose_stage_leg(dev)
            except Exception as e:
                extra["pose_stage"] = {"error": f"{type(e).__name__}: {e}"}
        if "unprojection_grids" in legs:
            try:
                extra["unprojection_grids"] = unprojection_grids_leg(dev)
            except Exception as e:
                extra["unprojection_grids"] = {"error": f"{type(e).__name__}: {e}"}
        if "unprojection_backward" in legs:
            try:
                extra["unprojection_backward"] = unprojection_backward_leg(dev)
            except Exception as e:
                extra["unprojection_backward"] = {"error": f"{type(e).__name__}: {e}"}
        result["legs"] = extra
    if world > 1:
        # the final barrier comes BEFORE the line: whether every rank got here is part of what the line says
        try:
            dist.barrier()
        except Exception as e:      # a peer has left (its deadline fired a moment earlier, or it died): the line must still get out
            print_line_once(f"final barrier failed ({type(e).__name__}): a peer rank left before this one")
    print_line_once()
    deadline.__exit__(None, None, None)
    D.shutdown()
    if rank == 0 and "output_check" in result and not result["output_check"]["ok"]:
        raise SystemExit(f"bench.py: the step's output does not match the reference golden: {result['output_check']}")
    if world > 1:
        # every rank agrees on the status without another collective: the deadline / barrier note is local knowledge, a leg's
        # error record is the same exception on every rank (all ranks run the legs alike)
        bad = line_state["incomplete"] is not None or line_state["rccl_ranks_seen"] != world or \
            any(isinstance(extra.get(k), dict) and "error" in extra[k] for k in COLLECTIVE_LEGS)
        if bad:
            raise SystemExit(EXIT_LEGS_INCOMPLETE)


if __name__ == "__main__":
    main()
