"""Round-3 GPU pins against outputs of the reference's own Python (tests/golden/make_goldens_r3.py):
  * the root net at the reference's OTHER shipped grid, 48x48x12 (prn32_cpn48x48x12_960x512_cam5.yaml): the generic
    inference-plan path (hipFFT plans instead of the z-DFT / 88x88 plane kernels, other Winograd / direct-conv shapes);
  * MultiPersonPoseNet.forward in TRAIN mode (lib/models/multi_person_posenet.py:36-102): the three losses and the
    gradient of backbone.final_layer.weight, with root-net proposals and with ground-truth proposals, scatter and
    deterministic backward;
  * MultiPersonPoseNetSSV.forward in TRAIN mode, pose-net stage (lib/models/multi_person_posenet_ssv.py:197-501):
    every loss term and three gradients; and the drop-in entry point tools/train_3d.py on an SSV YAML.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import golden_io as gio
from tests.test_gpu_reference_pins_r2 import _check_root

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("mode,golden", [("plan_cl", "rootnet_48"), ("plan_graph", "rootnet_48"), ("eager_plain", "rootnet_48"),
                                         ("plan_cl", "rootnet_160"), ("plan_graph", "rootnet_160")])
def test_rootnet_other_grids_vs_reference(dev, mode, golden):
    """48x48x12 (the reference's other shipped YAML) and BASELINE configs[3] (10 views, 160x160x40): CuboidProposalNet
    end to end against the reference's own run, through the inference plan's GENERIC path at these shapes"""
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.cuboid_proposal_net import CuboidProposalNet
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", golden + ".npz")):
        pytest.skip(golden + ".npz not generated")
    g = gio.load(golden)
    img, hm, V, J = [int(v) for v in g["img"]], [int(v) for v in g["hm"]], int(g["V"]), int(g["J"])
    cube = [int(v) for v in g["cube"]]
    seed = int(g["hm_seed"])
    rnd = syn.random_heatmaps(2, V, J, hm[1], hm[0], seed=seed)
    ppl, _ = syn.people_heatmaps(2, V, J, hm[1], hm[0], img, seed=seed + 1)
    hms = [torch.stack([0.35 * rnd[v][0], ppl[v][1]]) for v in range(V)]
    sums = np.array([float(h.double().sum()) for h in hms])
    assert np.allclose(sums, g["hm_sum"], rtol=0, atol=1e-6 * float(np.abs(sums).max()))
    meta = syn.make_meta(2, V, img)
    cfg = load_config(None, MULTI_PERSON__INITIAL_CUBE_SIZE=cube, PICT_STRUCT__CUBE_SIZE=[32, 32, 32])
    net = CuboidProposalNet(cfg)
    syn.fill_parameters_deterministic(net, seed=int(g["param_seed"]), scale=float(g["param_scale"]))
    net.eval().to(dev)
    hms = [h.to(dev) for h in hms]
    if mode == "eager_plain":
        net.v2v_net.fft_front = False
        net.v2v_net.winograd = False
    else:
        net.use_channels_last(True)
    with torch.no_grad():
        root_cubes, grid_centers = net(hms, meta)
        if mode == "plan_graph":
            from selfpose3d_amd.graphs import GraphedRootNet
            torch.cuda.synchronize()
            graphed = GraphedRootNet(net, hms, meta)
            root_cubes, grid_centers = graphed()
            torch.cuda.synchronize()
    _check_root(root_cubes, grid_centers, g, cube=cube, min_checked=6)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


# ---------------------------------------------------------------------------------------------------------------------
# Training pins.  What the library does underneath (round 5, read off MIOPEN_LOG_LEVEL=5 on the GPU box,
# tools/probe/miopen_log_probe.py): with cudnn.benchmark = False PyTorch still calls miopenFindConvolution*Algorithm, and
# MIOpen's default find mode DYNAMIC_HYBRID answers a find-db miss by TIMING the applicable kernels ("EvaluateInvokers ...
# Selected: ...") and keeping the fastest - per process, so the winner of a near-tie depends on the box and on what ran
# before.  Another backward algorithm = another summation order, and on these ill-conditioned gradients that moved the SSV
# backbone gradient between 1.0e-3 and 1.22e-2 of the reference's (rounds 3-4; the empty user db / binary cache of round 4
# did not pin it, they only removed other processes' leftovers).  So the two train-step pins run in a CHILD process
#   * with MIOPEN_FIND_MODE=FAST: a find-db miss is answered from the library's heuristics, nothing is timed - the same
#     kernels on every box of one image (SP3D_PINS_FIND_MODE overrides, e.g. DYNAMIC_HYBRID to see the spread);
#   * with a fresh, empty user db and kernel-binary cache;
#   * with MIOPEN_LOG_LEVEL=5 into a file, from which the child's record gets a "library" entry: MIOpen version, find mode,
#     db paths and the solver chosen per direction (FW / BWD / BWrW) with counts - so a surprise explains itself;
# and the bounds are judged against the reference's float64 reruns, wide enough for ANY kernel choice (see below).
# ---------------------------------------------------------------------------------------------------------------------
def _library_record(log_path, env):
    """what the MIOpen log of the child says: version, find mode, solver per direction"""
    import collections
    import re
    rec = {"find_mode_env": env.get("MIOPEN_FIND_MODE"), "user_db": env.get("MIOPEN_USER_DB_PATH"),
           "cache_dir": env.get("MIOPEN_CUSTOM_CACHE_DIR"), "solvers": {}}
    chosen = collections.defaultdict(collections.Counter)
    try:
        with open(log_path, errors="replace") as f:
            for ln in f:
                m = re.search(r"\b(FW|BWD|BWrW) Chosen Algorithm: (\S+)", ln)
                if m:
                    chosen[m.group(1)][m.group(2)] += 1
                    continue
                m = re.search(r"MIOPEN_FIND_MODE = (\S+)", ln)
                if m:
                    rec["find_mode"] = m.group(1)
                m = re.search(r"MIOpen version (\S+)", ln)
                if m:
                    rec["miopen"] = m.group(1)
                m = re.search(r"Raw device name: (\S+)", ln)
                if m:
                    rec["device"] = m.group(1)
    except OSError as e:
        rec["log_error"] = str(e)
    rec["solvers"] = {d: dict(c.most_common()) for d, c in chosen.items()}
    return rec


def _child(args, tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT, MIOPEN_USER_DB_PATH=str(tmp_path / "miopen_user_db"),
               MIOPEN_CUSTOM_CACHE_DIR=str(tmp_path / "miopen_cache"), MIOPEN_LOG_LEVEL="5",
               MIOPEN_FIND_MODE=os.environ.get("SP3D_PINS_FIND_MODE", "FAST"))
    os.makedirs(env["MIOPEN_USER_DB_PATH"], exist_ok=True)
    os.makedirs(env["MIOPEN_CUSTOM_CACHE_DIR"], exist_ok=True)
    log_path = str(tmp_path / "miopen_log.txt")
    with open(log_path, "w") as log:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(a) for a in args], cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=log, text=True, timeout=900)
    if r.returncode != 0:
        tail = [ln for ln in open(log_path, errors="replace") if not ln.startswith("MIOpen(HIP)")][-60:]
        raise AssertionError(r.stdout[-3000:] + "".join(tail))
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    rec["library"] = _library_record(log_path, env)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "training_pins_measured.json")
    try:
        cur = json.load(open(path))
    except Exception:
        cur = {}
    cur["_".join(str(a) for a in args[1:])] = rec
    json.dump(cur, open(path, "w"), indent=1)
    return rec


def _wide(x):
    """every floating tensor of a (nested) batch structure -> float64"""
    if torch.is_tensor(x):
        return x.double() if x.is_floating_point() else x
    if isinstance(x, dict):
        return {k: _wide(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_wide(v) for v in x)
    return x


def _train_step_child(tag, deterministic, batched=True, f64=False):
    """losses and gradients of one supervised train step against the reference's fp32 and float64 runs -> dict
    (batched: 0 = per-camera loop, 1 = one NCHW pass, 2 = one channels_last pass through the grouped BatchNorm kernels - the
    form tools/train_3d.py and the bench's train leg run)"""
    from selfpose3d_amd.multi_person_posenet import get_multi_person_pose_net
    from selfpose3d_amd import pose_resnet
    channels_last = int(batched) == 2
    pose_resnet.PoseResNet.batch_views_in_training = bool(batched)
    dev = torch.device("cuda:0")
    g = gio.load("train_step")
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    cfg = gio.train_cfg(USE_GT=(tag == "gt"))
    model = get_multi_person_pose_net(cfg, is_train=True)
    gio.he_fill(model, seed=int(g["param_seed"]))
    model.to(dev).train()
    if channels_last:
        model.use_channels_last(True)
    for pl in (model.root_net.project_layer, model.pose_net.project_layer):
        pl.deterministic_backward = deterministic
    inputs, t2d, w2d, t3d, meta, _ = gio.train_batch(cfg, B=2, seed=int(g["data_seed"]))
    if f64:            # a float64 model around the fp32 HIP unprojection: the library's convolution kernels drop out of the error
        model.double()
        inputs, t2d, w2d, t3d = _wide(inputs), _wide(t2d), _wide(w2d), _wide(t3d)
    inputs = [x.to(dev) for x in inputs]
    pred, hms, gc, l2d, l3d, lcord = model(views=inputs, meta=meta, targets_2d=t2d, weights_2d=w2d, targets_3d=t3d[0])
    rec = {"hm_sum_rel": float(np.abs(np.array([float(h.double().sum()) for h in hms]) / g[f"{tag}_hm_sum"] - 1.0).max())}
    for name, val in (("loss_2d", l2d), ("loss_3d", l3d), ("loss_cord", lcord)):
        ref = float(g[f"{tag}_{name}"])
        rec[name + "_rel"] = abs(float(val) - ref) / max(abs(ref), 1e-6)
    rec["valid_equal"] = bool(np.array_equal((gc[:, :, 3] >= 0).cpu().numpy(), g[f"{tag}_grid_centers"][:, :, 3] >= 0))

    def truth(got, name):           # error against the reference's float64 gradient, and the reference's own fp32 error
        ref32, ref64 = g[name], g[name + "_f64"]
        e_ref, e_got = _rel(ref32, ref64), _rel(got, ref64)
        rec[name] = {"err_vs_f64": e_got, "ref_fp32_err_vs_f64": e_ref, "ratio": e_got / max(e_ref, 1e-30)}

    fl = model.backbone.final_layer.weight
    for nm, term in (("2d", l2d), ("3d", l3d), ("cord", lcord)):
        if f"{tag}_grad_final_{nm}_f64" in g and term.requires_grad:
            gt_, = torch.autograd.grad(term.mean(), fl, retain_graph=True, allow_unused=True)
            got = np.zeros(tuple(fl.shape), np.float32) if gt_ is None else gt_.cpu().numpy()
            truth(got, f"{tag}_grad_final_{nm}")
    if tag == "net":
        ol = model.root_net.v2v_net.output_layer.weight
        fc = model.root_net.v2v_net.front_layers[0].block[0].weight
        ga, gb = torch.autograd.grad(l3d.mean(), (ol, fc), retain_graph=True)
        rec["net_grad_root_out_rel"] = _rel(ga.cpu().numpy(), g["net_grad_root_out"])
        truth(gb.cpu().numpy()[:, :15], "net_grad_root_front")
    (l2d.mean() + l3d.mean() + lcord.mean()).backward()
    truth(model.backbone.final_layer.weight.grad.cpu().numpy(), f"{tag}_grad_final")
    gp = model.pose_net.v2v_net.output_layer.weight.grad
    gp = np.zeros_like(g[f"{tag}_grad_pose_out"]) if gp is None else gp.cpu().numpy()
    if np.abs(g[f"{tag}_grad_pose_out"]).max() > 0:
        truth(gp, f"{tag}_grad_pose_out")
    else:
        rec["pose_out_grad_absmax"] = float(np.abs(gp).max())
    return rec


# error against float64 as a multiple of the reference's own fp32 error against float64 (its gradient is 1.4 % (3D term) /
# 0.17 % (pose term) away from its float64 rerun: fp32 backward through train-mode BatchNorm over ~20 conv layers is
# ill-conditioned); bounds = ~3x the largest value measured with the pinned kernel selection, see above.  A wrong gradient
# (a missing term, a sign) is off by O(1) = 70x .. 600x.
# Both backbone passes are pinned: "batched" (this repo's default, 6 % faster train step: one pass over all views with
# per-view BatchNorm statistics, pose_resnet.ViewBatchNorm2d) and "loop" (one call per camera, as the reference) - the same
# arithmetic, other convolution kernels.  Even with the selection pinned the ill-conditioned terms move from run to run
# (atomic split-K weight-gradient kernels, fp32 atomics of the scatter): measured over 8 child runs on 3 boxes, batched / loop:
# 3-D term 0.91-1.06 / 1.06, first root-V2V layer 0.82-1.40 / 0.75, pose term 4.9-8.1 / 5.2, pose head 2.5-3.4 / 2.3-2.7.
# Round 5: the bounds follow ONE rule, the SSV pin's: absolute bound = min(CAP, max(k x reference's own fp32 error, FLOOR)),
# k chosen so the absolute bound is >= 4x the largest value ever measured (the round-4 driver box showed the library's
# kernel choice moving an ill-conditioned gradient 7.5x between boxes of one pool) and CAP = 0.1 keeps every bound >= 10x
# below an O(1) error.  Absolute bounds: 3-D term / total 0.080 (seen 0.0154), first root-V2V layer 0.046 (0.0054),
# pose term / total 0.068 (0.0137), pose head 0.042 (0.0057); the well-conditioned 2-D term sits at the floor (seen 4e-7).
GRAD_RATIO_BOUND = {"net_grad_final_2d": 3.0, "net_grad_final_3d": 5.5, "net_grad_final": 5.5, "net_grad_root_front": 12.0,
                    "gt_grad_final_2d": 3.0, "gt_grad_final_cord": 40.0, "gt_grad_final": 40.0, "gt_grad_pose_out": 25.0}
GRAD_FLOOR = 1e-4          # where the problem is well conditioned both errors are ~1e-6: a ratio means nothing below this
GRAD_CAP = 0.1


@pytest.mark.miopen_sensitive
@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("tag,views", [("net", "batched"), ("gt", "batched"), ("net", "loop"), ("gt", "loop"),
                                       ("net", "batched_cl"), ("gt", "batched_cl")])
def test_supervised_train_step_vs_reference(dev, tag, views, deterministic, tmp_path):
    """losses <= 1e-4 relative; gradients against the reference's fp32 AND float64 runs, with the scatter and the
    deterministic unprojection backward, with the backbone's views batched (default) and looped"""
    if views != "batched" and deterministic:
        pytest.skip("the deterministic scatter is pinned with the default backbone pass")
    rec = _child(["--train-step-child", tag, int(deterministic), {"loop": 0, "batched": 1, "batched_cl": 2}[views]], tmp_path)
    print(json.dumps(rec))
    assert rec["hm_sum_rel"] <= 1e-4 and rec["valid_equal"]
    for name in ("loss_2d", "loss_3d", "loss_cord"):
        assert rec[name + "_rel"] <= 1e-4, (name, rec)
    if tag == "net":
        assert rec["net_grad_root_out_rel"] <= 1e-4               # needs only the forward pass to be right
    seen = 0
    for key, v in rec.items():
        if isinstance(v, dict) and "err_vs_f64" in v:
            bound = GRAD_RATIO_BOUND[key]
            assert v["err_vs_f64"] <= min(GRAD_CAP, max(bound * v["ref_fp32_err_vs_f64"], GRAD_FLOOR)), \
                (key, v, bound, rec.get("library"))
            seen += 1
    assert seen >= 3
    if "pose_out_grad_absmax" in rec:
        assert rec["pose_out_grad_absmax"] == 0.0           # pose net not reached: zero-anchored, exactly zero gradient


def _ssv_step_child(batched=True, f64=False):
    from selfpose3d_amd.models import get_multi_person_pose_net
    from selfpose3d_amd import pose_resnet
    pose_resnet.PoseResNet.batch_views_in_training = bool(batched)
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    g = gio.load("ssv_train_step")
    cfg = gio.train_cfg(ssv=True)
    model = get_multi_person_pose_net(cfg, is_train=True)
    gio.he_fill(model, seed=int(g["param_seed"]))
    model.to(dev).train()
    model.root_net.eval()                                # engine.train_3d_ssv / lib/core/function.py:46-48
    b = gio.train_batch(cfg, B=2, seed=int(g["data_seed"]), ssv=True)
    if f64:
        model.double()
        b = _wide(b)
    (in1, t1, w1, d1, m1, _, in2, t2, w2, d2, m2, _, in3, t3, w3, d3, m3, _) = b
    in1, in2, in3 = ([x.to(dev) for x in v] for v in (in1, in2, in3))
    pred, hm3, gc, losses = model(views1=in1, meta1=m1, targets_2d1=t1, weights_2d1=w1, targets_3d1=d1[0],
                                  views2=in2, meta2=m2, targets_2d2=t2, weights_2d2=w2, targets_3d2=d2[0],
                                  views3=in3, meta3=m3, targets_2d3=t3, weights_2d3=w3, targets_3d3=d3[0],
                                  epoch=int(g["epoch"]))
    rec = {"keys_equal": sorted(losses) == [str(k) for k in g["keys"]],
           "valid_equal": bool(np.array_equal((gc[:, :, 3] >= 0).cpu().numpy(), g["grid_centers"][:, :, 3] >= 0)), "loss_rel": {}}
    for k, v in losses.items():
        ref = float(g["loss_" + k])
        rec["loss_rel"][k] = abs(float(v.mean()) - ref) / max(abs(ref), 1e-6)
    ok = g["grid_centers"][:, :, 3] >= 0
    rec["joints_mm"] = float(np.abs(pred.cpu().numpy()[ok][..., :3] - g["pred"][ok][..., :3]).max())
    sum(v.mean() for v in losses.values() if v.requires_grad).backward()
    for nm, got in (("grad_final", model.backbone.final_layer.weight.grad),
                    ("grad_pose_out", model.pose_net.v2v_net.output_layer.weight.grad),
                    ("grad_attn_final", model.attn.backbone.final_layer.weight.grad)):
        got, ref32, ref64 = got.cpu().numpy(), g[nm], g[nm + "_f64"]
        e_ref, e_got = _rel(ref32, ref64), _rel(got, ref64)
        rec[nm] = {"err_vs_f64": e_got, "ref_fp32_err_vs_f64": e_ref, "ratio": e_got / max(e_ref, 1e-30),
                   "err_vs_ref_fp32": _rel(got, ref32)}
    return rec


# SSV step: joints in mm on 2000 mm cubes of 62.5 mm voxels (the soft-argmax of a random-weight pose net amplifies the
# library convolutions' rounding; a wrong cube or joint is off by >= a voxel).
# Gradients are judged like the supervised step's: error against the reference's FLOAT64 rerun of the same step
# (tests/golden/make_goldens_r3.py _ssv_float64_rerun) as a multiple of the reference's own fp32 error against it.  On CPU
# the reference's fp32 gradients are 1.4e-3 (backbone) / 3.0e-4 (pose head) / 7.1e-5 (attention head) from float64; this
# repo's, through MIOpen's backward kernels, were seen between 1.0e-3 and 1.22e-2 / 4.9e-4 and 2.6e-3 / 2.9e-5 and 7.4e-5
# from the reference's fp32 values - the largest on a box of the round-4 driver run, by which convolution kernels the
# library chose there.  k = 40 puts the bounds at 5.7e-2 / 1.2e-2 / 2.9e-3: the worst value ever seen passes with 4.7x to
# spare, and a wrong gradient (a missing term, a sign: off by O(1)) fails by >= 17x.  The bound is capped at 0.1 so that
# stays true whatever the yardstick file holds.
SSV_JOINTS_MM = 4.0                                                                  # measured 0.49-1.55 over 8 runs
SSV_GRAD_K, SSV_GRAD_FLOOR, SSV_GRAD_CAP = 40.0, 1e-4, 0.1


@pytest.mark.miopen_sensitive
@pytest.mark.parametrize("views", ["batched", "loop"])
def test_ssv_train_step_vs_reference(dev, views, tmp_path):
    rec = _child(["--ssv-step-child", int(views == "batched")], tmp_path)
    print(json.dumps(rec))
    assert rec["keys_equal"] and rec["valid_equal"]
    for k, e in rec["loss_rel"].items():
        assert e <= 2e-4, (k, e)
    assert rec["joints_mm"] <= SSV_JOINTS_MM, rec
    for nm in ("grad_final", "grad_pose_out", "grad_attn_final"):
        v = rec[nm]
        bound = min(max(SSV_GRAD_K * v["ref_fp32_err_vs_f64"], SSV_GRAD_FLOOR), SSV_GRAD_CAP)
        assert v["err_vs_f64"] <= bound, (nm, v, bound, rec.get("library"))


# Round 6 (advisor: the fp32 bounds above cannot see a gradient term that is a few percent off): the SAME steps with the model
# in float64 - convolutions, BatchNorm (grouped kernels: float64 statistics), soft-argmax, losses - around the fp32 HIP
# unprojection (forward, pass mask, scatter) and the fp32 rendering kernels.  The library's convolution choice no longer matters;
# what is left is the fp32 rounding of this repo's own kernels, amplified by the step's conditioning (the reference's all-fp32
# run is 1.4e-2 from its float64 run on the 3-D term).  Measured (profiles/r06_training_pins_f64.json; batched and looped backbone
# passes agree to 5 digits - nothing library-dependent is left): supervised 3.7e-8 (2-D term), 1.37e-3 (3-D term), 1.19e-3 (first
# root-V2V layer), 1.77e-3 / 2.2e-3 (pose terms, ground-truth proposals); SSV 9.6e-4 / 3.0e-4 / 5.6e-5; losses <= 3.4e-6.
# One bound for all: 5e-3 - a gradient term that is 1 % off fails, where the fp32 bounds above stop at 4-8 %.
F64_BOUND = 5e-3


@pytest.mark.parametrize("tag,views", [("net", "batched_cl"), ("gt", "batched_cl"), ("net", "loop")])
def test_supervised_train_step_float64_model_vs_reference_float64(dev, tag, views, tmp_path):
    rec = _child(["--train-step-child", tag, 1, {"loop": 0, "batched": 1, "batched_cl": 2}[views], "f64"], tmp_path)
    print(json.dumps(rec))
    assert rec["valid_equal"]
    for name in ("loss_2d", "loss_3d", "loss_cord"):
        assert rec[name + "_rel"] <= 1e-5, (name, rec)
    seen = 0
    for key, v in rec.items():
        if isinstance(v, dict) and "err_vs_f64" in v:
            assert v["err_vs_f64"] <= F64_BOUND, (key, v)
            seen += 1
    assert seen >= 3


@pytest.mark.parametrize("views", ["batched", "loop"])
def test_ssv_train_step_float64_model_vs_reference_float64(dev, views, tmp_path):
    rec = _child(["--ssv-step-child", int(views == "batched"), "f64"], tmp_path)
    print(json.dumps(rec))
    assert rec["keys_equal"] and rec["valid_equal"]
    for nm in ("grad_final", "grad_pose_out", "grad_attn_final"):
        assert rec[nm]["err_vs_f64"] <= F64_BOUND, (nm, rec[nm])


def _train_full_child(f64=False):
    """one supervised train step at BASELINE configs[2]'s sizes (ResNet-50, 5 x 960x512, 80x80x20 + 64^3, batch 2, proposals from
    ground truth) against tests/golden/train_step_full.npz (reference fp32 run + its float64 rerun)"""
    from selfpose3d_amd.multi_person_posenet import get_multi_person_pose_net
    dev = torch.device("cuda:0")
    g = gio.load("train_step_full")
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    cfg = gio.train_full_cfg(USE_GT=True)
    model = get_multi_person_pose_net(cfg, is_train=True)
    gio.he_fill(model, seed=int(g["param_seed"]))
    model.to(dev).train()
    if not f64:
        model.use_channels_last(True)                        # the form tools/train_3d.py and the bench's train leg run
    inputs, t2d, w2d, t3d, meta, _ = gio.train_batch(cfg, B=2, seed=int(g["data_seed"]))
    if f64:                                                  # (NCHW backbone: the grouped kernels take float64 up to 256 channels)
        model.double()
        inputs, t2d, w2d, t3d = _wide(inputs), _wide(t2d), _wide(w2d), _wide(t3d)
    inputs = [x.to(dev) for x in inputs]
    pred, hms, gc, l2d, l3d, lcord = model(views=inputs, meta=meta, targets_2d=t2d, weights_2d=w2d, targets_3d=t3d[0])
    (l2d.mean() + l3d.mean() + lcord.mean()).backward()
    rec = {"valid_equal": bool(np.array_equal((gc[:, :, 3] >= 0).cpu().numpy(), g["grid_centers"][:, :, 3] >= 0)),
           "hm_sum_rel": float(np.abs(np.array([float(h.double().sum()) for h in hms]) / g["hm_sum"] - 1.0).max())}
    for name, val in (("loss_2d", l2d), ("loss_3d", l3d), ("loss_cord", lcord)):
        ref = float(g[name + ("_f64" if f64 else "")])
        rec[name + "_rel"] = abs(float(val) - ref) / max(abs(ref), 1e-6)
    ok = g["grid_centers"][:, :, 3] >= 0
    rec["joints_mm"] = float(np.abs(pred.detach().cpu().numpy()[ok][..., :3] - g["pred"][ok][..., :3]).max())
    for nm, got in (("grad_final", model.backbone.final_layer.weight.grad),
                    ("grad_pose_out", model.pose_net.v2v_net.output_layer.weight.grad),
                    ("grad_conv1_sub", model.backbone.conv1.weight.grad.reshape(-1)[::7])):
        got, ref32, ref64 = got.cpu().numpy(), g[nm], g[nm + "_f64"]
        rec[nm] = {"err_vs_f64": _rel(got, ref64), "ref_fp32_err_vs_f64": _rel(ref32, ref64), "err_vs_ref_fp32": _rel(got, ref32)}
    return rec


FULL_GRAD_K, FULL_F64_BOUND = 40.0, 2e-2


@pytest.mark.miopen_sensitive
@pytest.mark.parametrize("f64", [False, True])
def test_train_step_at_full_size_vs_reference(dev, f64, tmp_path):
    """round-5 review weak #6: the train-step pins were at 128x96 / 24x24x8 / 16^3 only.  fp32: the rule of the small pins
    (k x the reference's own fp32-vs-float64 error, capped at 0.1); float64 model around the fp32 kernels: FULL_F64_BOUND"""
    rec = _child(["--train-full-child", int(f64)], tmp_path)
    print(json.dumps(rec))
    assert rec["valid_equal"] and rec["hm_sum_rel"] <= 1e-4
    for name in ("loss_2d", "loss_3d", "loss_cord"):
        assert rec[name + "_rel"] <= (1e-4 if f64 else 2e-4), (name, rec)       # measured: 1.8e-5 (float64 model) / 1.9e-6
    # measured (profiles/r06_training_pins_f64.json): fp32 2.1e-2 / 9.5e-3 / 3.3e-2 where the reference's own fp32 run is
    # 2.7e-2 / 1.5e-2 / 4.7e-2 from its float64 run; float64 model 8.6e-3 / 3.4e-3 / 9.9e-3
    for nm in ("grad_final", "grad_pose_out", "grad_conv1_sub"):
        v = rec[nm]
        bound = FULL_F64_BOUND if f64 else min(GRAD_CAP, max(FULL_GRAD_K * v["ref_fp32_err_vs_f64"], GRAD_FLOOR))
        assert v["err_vs_f64"] <= bound, (nm, v, bound, rec.get("library"))


@pytest.mark.miopen_sensitive
def test_train_entry_point_runs_an_ssv_yaml(dev, tmp_path):
    """drop-in check: tools/train_3d.py dispatches MODEL multi_person_posenet_ssv / WITH_SSV to the
    self-supervised loop and model, two iterations + validation on synthetic three-set frames"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_3d.py"), "--cfg",
                        os.path.join(ROOT, "configs", "synthetic_small_ssv.yaml"), "--frames", "2", "--max-iters", "2"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "loss_pose3d_ssv" in r.stderr + r.stdout


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    if sys.argv[1] == "--train-step-child":
        print(json.dumps(_train_step_child(sys.argv[2], bool(int(sys.argv[3])), int(sys.argv[4]),
                                           len(sys.argv) > 5 and sys.argv[5] == "f64")))
    elif sys.argv[1] == "--train-full-child":
        print(json.dumps(_train_full_child(bool(int(sys.argv[2])))))
    elif sys.argv[1] == "--ssv-step-child":
        print(json.dumps(_ssv_step_child(bool(int(sys.argv[2])), len(sys.argv) > 3 and sys.argv[3] == "f64")))
