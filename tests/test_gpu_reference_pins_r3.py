"""Round-3 GPU pins against outputs of the reference's own Python (tests/golden/make_goldens_r3.py):
  * the root net at the reference's OTHER shipped grid, 48x48x12 (prn32_cpn48x48x12_960x512_cam5.yaml): the generic
    inference-plan path (hipFFT plans instead of the z-DFT / 88x88 plane kernels, other Winograd / direct-conv shapes);
  * MultiPersonPoseNet.forward in TRAIN mode (lib/models/multi_person_posenet.py:36-102): the three losses and the
    gradient of backbone.final_layer.weight, with root-net proposals and with ground-truth proposals, scatter and
    deterministic backward;
  * MultiPersonPoseNetSSV.forward in TRAIN mode, pose-net stage (lib/models/multi_person_posenet_ssv.py:197-501):
    every loss term and three gradients; and the drop-in entry point tools/train_3d.py on an SSV YAML.
"""
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


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("tag", ["net", "gt"])
def test_supervised_train_step_vs_reference(dev, tag, deterministic, monkeypatch):
    """losses <= 1e-4 relative; gradients against the reference's fp32 AND float64 runs (see below), with the scatter and
    the deterministic unprojection backward"""
    from selfpose3d_amd.multi_person_posenet import get_multi_person_pose_net
    g = gio.load("train_step")
    # MIOpen picks its backward kernels by timing them (find mode) - a different algorithm, with a different summation
    # order, from one process to the next: on this ill-conditioned gradient that alone moved the error against float64
    # between 3x and 18x the reference's.  Immediate mode + deterministic kernels: the same algorithms every run.
    monkeypatch.setattr(torch.backends.cudnn, "benchmark", False)
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    cfg = gio.train_cfg(USE_GT=(tag == "gt"))
    model = get_multi_person_pose_net(cfg, is_train=True)
    gio.he_fill(model, seed=int(g["param_seed"]))
    model.to(dev).train()
    for pl in (model.root_net.project_layer, model.pose_net.project_layer):
        pl.deterministic_backward = deterministic
    inputs, t2d, w2d, t3d, meta, _ = gio.train_batch(cfg, B=2, seed=int(g["data_seed"]))
    inputs = [x.to(dev) for x in inputs]
    pred, hms, gc, l2d, l3d, lcord = model(views=inputs, meta=meta, targets_2d=t2d, weights_2d=w2d, targets_3d=t3d[0])
    hm_sum = np.array([float(h.double().sum()) for h in hms])
    assert np.allclose(hm_sum, g[f"{tag}_hm_sum"], rtol=1e-4)
    for name, val in (("loss_2d", l2d), ("loss_3d", l3d), ("loss_cord", lcord)):
        ref = float(g[f"{tag}_{name}"])
        assert abs(float(val) - ref) <= 1e-4 * max(abs(ref), 1e-6), (name, float(val), ref)
    assert np.array_equal((gc[:, :, 3] >= 0).cpu().numpy(), g[f"{tag}_grid_centers"][:, :, 3] >= 0)
    # Gradients.  fp32 backward passes through this net (train-mode BatchNorm over ~20 conv layers) are ill-conditioned:
    # the REFERENCE's own fp32 gradient is 1.4 % (3D term) / 0.17 % (pose term) away from its float64 rerun, which the
    # golden stores as the yardstick.  Pin: this repo's gradient must be about as close to the float64 one as the
    # reference's fp32 gradient is (measured on MI355X with the fixed kernel selection above: 1.06x its error on the 3D
    # term, 5.24x on the pose term, where MIOpen's backward kernels and the fp32 soft-argmax add their own rounding -
    # 0.9 % of the gradient's magnitude; with find mode - and in immediate mode too, once another test of the same run has
    # left entries in MIOpen's user db - it moved between 3x and 18x from process to process; bound: 40x = 7 % of the
    # gradient's magnitude - a wrong gradient (a missing term, a sign) is off by O(1) = 600x), and where the problem is
    # well conditioned (2D term, last V2V layer) match the reference to 1e-4.
    def close_to_truth(got, name, floor=1e-4):
        ref32, ref64 = g[name], g[name + "_f64"]
        e_ref, e_got = _rel(ref32, ref64), _rel(got, ref64)
        print(f"[{tag}] {name}: error vs float64 {e_got:.3e} (reference fp32: {e_ref:.3e}, ratio {e_got / max(e_ref, 1e-30):.2f})")
        assert e_got <= max(40.0 * e_ref, floor), (name, e_got, e_ref)

    fl = model.backbone.final_layer.weight
    for nm, term in (("2d", l2d), ("3d", l3d), ("cord", lcord)):
        if f"{tag}_grad_final_{nm}_f64" in g and term.requires_grad:
            gt_, = torch.autograd.grad(term.mean(), fl, retain_graph=True, allow_unused=True)
            got = np.zeros(tuple(fl.shape), np.float32) if gt_ is None else gt_.cpu().numpy()
            close_to_truth(got, f"{tag}_grad_final_{nm}")
    if tag == "net":
        ol = model.root_net.v2v_net.output_layer.weight
        fc = model.root_net.v2v_net.front_layers[0].block[0].weight
        ga, gb = torch.autograd.grad(l3d.mean(), (ol, fc), retain_graph=True)
        assert _rel(ga.cpu().numpy(), g["net_grad_root_out"]) <= 1e-4          # needs only the forward pass to be right
        close_to_truth(gb.cpu().numpy()[:, :15], "net_grad_root_front")
    (l2d.mean() + l3d.mean() + lcord.mean()).backward()
    close_to_truth(model.backbone.final_layer.weight.grad.cpu().numpy(), f"{tag}_grad_final")
    gp = model.pose_net.v2v_net.output_layer.weight.grad
    gp = np.zeros_like(g[f"{tag}_grad_pose_out"]) if gp is None else gp.cpu().numpy()
    if np.abs(g[f"{tag}_grad_pose_out"]).max() > 0:
        # (a fixed 2e-3 against the reference's fp32 gradient held on most boxes and failed at 2.5e-3 on one: MIOpen's
        # kernel choice moves it; the float64 yardstick is the honest bound here too)
        close_to_truth(gp, f"{tag}_grad_pose_out")
    else:
        assert np.abs(gp).max() == 0.0                  # pose net not reached: zero-anchored, exactly zero gradient


def test_ssv_train_step_vs_reference(dev, monkeypatch):
    from selfpose3d_amd.models import get_multi_person_pose_net
    monkeypatch.setattr(torch.backends.cudnn, "benchmark", False)       # same MIOpen kernels every run (see above)
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    g = gio.load("ssv_train_step")
    cfg = gio.train_cfg(ssv=True)
    model = get_multi_person_pose_net(cfg, is_train=True)
    gio.he_fill(model, seed=int(g["param_seed"]))
    model.to(dev).train()
    model.root_net.eval()                                # engine.train_3d_ssv / lib/core/function.py:46-48
    b = gio.train_batch(cfg, B=2, seed=int(g["data_seed"]), ssv=True)
    (in1, t1, w1, d1, m1, _, in2, t2, w2, d2, m2, _, in3, t3, w3, d3, m3, _) = b
    in1, in2, in3 = ([x.to(dev) for x in v] for v in (in1, in2, in3))
    pred, hm3, gc, losses = model(views1=in1, meta1=m1, targets_2d1=t1, weights_2d1=w1, targets_3d1=d1[0],
                                  views2=in2, meta2=m2, targets_2d2=t2, weights_2d2=w2, targets_3d2=d2[0],
                                  views3=in3, meta3=m3, targets_2d3=t3, weights_2d3=w3, targets_3d3=d3[0],
                                  epoch=int(g["epoch"]))
    assert sorted(losses) == [str(k) for k in g["keys"]]
    assert np.array_equal((gc[:, :, 3] >= 0).cpu().numpy(), g["grid_centers"][:, :, 3] >= 0)
    for k, v in losses.items():
        ref = float(g["loss_" + k])
        assert abs(float(v.mean()) - ref) <= 2e-4 * max(abs(ref), 1e-6), (k, float(v.mean()), ref)
    # predicted joints of the second augmented pass (mm)
    ok = g["grid_centers"][:, :, 3] >= 0
    # mm, on 2000 mm cubes of 62.5 mm voxels: the soft-argmax of a random-weight pose net amplifies the library convolutions'
    # rounding (measured 0.2-1.6 mm depending on MIOpen's kernel choice); a wrong cube or joint is off by >= a voxel
    assert np.abs(pred.cpu().numpy()[ok][..., :3] - g["pred"][ok][..., :3]).max() <= 5.0
    sum(v.mean() for v in losses.values() if v.requires_grad).backward()
    for nm, got in (("grad_final", model.backbone.final_layer.weight.grad),
                    ("grad_pose_out", model.pose_net.v2v_net.output_layer.weight.grad),
                    ("grad_attn_final", model.attn.backbone.final_layer.weight.grad)):
        e = _rel(got.cpu().numpy(), g[nm])
        print(f"[ssv] {nm}: {e:.3e} from the reference's fp32 gradient")
        assert e <= 2e-2, (nm, e)       # measured 7e-5 .. 1.3e-3; the library's kernel choice moves it (see above); wrong = O(1)


def test_train_entry_point_runs_an_ssv_yaml(dev, tmp_path):
    """drop-in check: tools/train_3d.py dispatches MODEL multi_person_posenet_ssv / WITH_SSV to the
    self-supervised loop and model, two iterations + validation on synthetic three-set frames"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_3d.py"), "--cfg",
                        os.path.join(ROOT, "configs", "synthetic_small_ssv.yaml"), "--frames", "2", "--max-iters", "2"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "loss_pose3d_ssv" in r.stderr + r.stdout
