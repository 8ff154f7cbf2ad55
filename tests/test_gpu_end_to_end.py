"""GPU end-to-end: full model (backbone -> root net -> pose net) forward, one optimiser step through the
HIP backward kernel, and the train / validate CLI entry points on synthetic frames."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "synthetic_small.yaml")


def _batch(cfg, n=2):
    from selfpose3d_amd.synthetic_dataset import SyntheticPanoptic
    ds = SyntheticPanoptic(cfg, num_frames=n, seed=3)
    return next(iter(torch.utils.data.DataLoader(ds, batch_size=n)))


def test_model_forward_and_training_step():
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.multi_person_posenet import get_multi_person_pose_net
    dev = torch.device("cuda:0")
    cfg = load_config(CFG)
    model = get_multi_person_pose_net(cfg, is_train=True).to(dev)
    inputs, t2d, w2d, t3d, meta, ihm = _batch(cfg)
    # inference from given heat-maps (views=None path of the reference): proposals near the GT roots
    model.eval()
    with torch.no_grad():
        pred, hms, gc, l2, l3, lc = model(meta=meta, input_heatmaps=[h.to(dev) for h in ihm])
    assert pred.shape == (2, 4, 15, 5) and gc.shape == (2, 4, 5)
    # training step with the backbone trainable: gradients reach it through sp3d_unproject_bwd
    model.train()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    out = model(views=[v.to(dev) for v in inputs], meta=meta, targets_2d=t2d, weights_2d=w2d, targets_3d=t3d[0])
    loss = out[3] + out[4] + out[5]
    assert torch.isfinite(loss)
    opt.zero_grad()
    loss.backward()
    g = model.backbone.final_layer.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    opt.step()


def test_cli_entry_points(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_3d.py"), "--cfg", CFG, "--frames", "4",
                        "--max-iters", "2"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = os.path.join(str(tmp_path), "output", "synthetic", "multi_person_posenet", "synthetic_small")
    assert os.path.isfile(os.path.join(out, "checkpoint.pth.tar")) and os.path.isfile(os.path.join(out, "final_state.pth.tar"))
    ck = torch.load(os.path.join(out, "checkpoint.pth.tar"), map_location="cpu")
    assert set(ck) >= {"epoch", "state_dict", "precision", "optimizer"}
    assert any(k.startswith("root_net.v2v_net.") for k in ck["state_dict"]) and any(k.startswith("backbone.") for k in ck["state_dict"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "validate_3d.py"), "--cfg", CFG, "--frames", "4",
                        "--test-file", os.path.join(out, "final_state.pth.tar")], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "root recall" in r.stderr
    # scalars under the reference's names (lib/core/function.py:329-334), through tensorboardX / torch's writer when present,
    # else selfpose3d_amd.engine.ScalarLog
    logs = os.listdir(os.path.join(out, "log"))
    assert logs, "no scalar log written"
    if "scalars.jsonl" in logs:
        import json
        tags = {json.loads(ln)["tag"] for ln in open(os.path.join(out, "log", "scalars.jsonl"))}
        assert {"train_loss", "train_loss_3d", "train_loss_cord"} <= tags
    # validate without a checkpoint: an error as in the reference (tools/validate_3d.py:91-92), unless --random-init is given
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "validate_3d.py"), "--cfg", CFG, "--frames", "2",
                        "--test-file", os.path.join(out, "no_such_file.pth.tar")], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "Check the model file for testing" in r.stderr


def test_stage_handoff_through_the_cli(tmp_path):
    """root-net stage run -> its model_epoch_1.pth.tar named by INIT_ROOTNET (and the same file, a whole-model file, by
    PRETRAINED_BACKBONE + PRETRAINED_BACKBONE_PSEUDOGT) in the next stage's YAML -> tools/train_3d.py loads both and says so;
    a YAML that names a missing file fails before training (reference tools/train_3d.py:150-180)"""
    import yaml
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = yaml.safe_load(open(CFG))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_3d.py"), "--cfg", CFG, "--frames", "2",
                        "--max-iters", "1"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = os.path.join(str(tmp_path), "output", "synthetic", "multi_person_posenet", "synthetic_small")
    stage1 = os.path.join(out, "model_epoch_1.pth.tar")
    assert os.path.isfile(stage1)
    nxt = dict(base)
    nxt["NETWORK"] = dict(base["NETWORK"], INIT_ROOTNET=stage1, PRETRAINED_BACKBONE=stage1, PRETRAINED_BACKBONE_PSEUDOGT=True,
                          FREEZE_ROOTNET=True)
    y2 = os.path.join(str(tmp_path), "stage2.yaml")
    yaml.safe_dump(nxt, open(y2, "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_3d.py"), "--cfg", y2, "--frames", "2",
                        "--max-iters", "1"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "initialised from NETWORK.PRETRAINED_BACKBONE" in r.stderr and "initialised from NETWORK.INIT_ROOTNET" in r.stderr
    sd1 = torch.load(stage1, map_location="cpu")
    sd2 = torch.load(os.path.join(str(tmp_path), "output", "synthetic", "multi_person_posenet", "stage2", "final_state.pth.tar"),
                     map_location="cpu")
    root = [k for k in sd1 if k.startswith("root_net.") and sd1[k].is_floating_point() and "running" not in k]
    assert root and all(torch.equal(sd1[k], sd2[k]) for k in root)          # loaded AND frozen: untouched by the step
    nxt["NETWORK"]["INIT_ROOTNET"] = os.path.join(str(tmp_path), "missing.pth.tar")
    y3 = os.path.join(str(tmp_path), "stage3.yaml")
    yaml.safe_dump(nxt, open(y3, "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_3d.py"), "--cfg", y3, "--frames", "2",
                        "--max-iters", "1"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "missing.pth.tar" in r.stderr


def test_rootnet_soft_forward_and_synthetic_training_branch():
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.cuboid_proposal_net import CuboidProposalNet
    from selfpose3d_amd.cuboid_proposal_net_soft import CuboidProposalNetSoft
    dev = torch.device("cuda:0")
    cfg = load_config(None, NETWORK__IMAGE_SIZE=[384, 288], NETWORK__HEATMAP_SIZE=[96, 72],
                      NETWORK__ROOTNET_ROOTHM=True, NETWORK__ROOTNET_TRAIN_SYNTH=True,
                      MULTI_PERSON__INITIAL_CUBE_SIZE=[24, 24, 8])
    B, V, J = 2, 5, 15
    meta = syn.make_meta(B, V, (384, 288))
    hms, _ = syn.people_heatmaps(B, V, J, 72, 96, (384, 288), seed=11)
    hms = [h.to(dev) for h in hms]
    soft = CuboidProposalNetSoft(cfg)
    syn.fill_parameters_deterministic(soft, seed=5, scale=0.05)
    soft.to(dev).eval()
    plain = CuboidProposalNet(cfg)
    plain.load_state_dict(soft.state_dict())
    plain.to(dev).eval()
    with torch.no_grad():
        rc_s, a, b, gc_s = soft(hms, meta)
        rc_p, gc_p = plain(hms, meta)
    assert a is None and b is None and torch.equal(rc_s, rc_p) and torch.equal(gc_s, gc_p)
    soft.train()
    rc, syn_cubes, target, gc = soft(hms, meta)
    assert syn_cubes.shape == target.shape == (B, 24, 24, 8)
    loss = torch.nn.functional.mse_loss(syn_cubes, target)
    loss.backward()
    assert soft.v2v_net.output_layer.weight.grad is not None


def test_hip_graph_replay_matches_eager_and_tracks_new_inputs():
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.cuboid_proposal_net import CuboidProposalNet
    from selfpose3d_amd.graphs import GraphedRootNet
    dev = torch.device("cuda:0")
    cfg = load_config(None, NETWORK__IMAGE_SIZE=[384, 288], NETWORK__HEATMAP_SIZE=[96, 72],
                      MULTI_PERSON__INITIAL_CUBE_SIZE=[24, 24, 8])
    B, V, J = 2, 5, 15
    meta = syn.make_meta(B, V, (384, 288))
    hms_a = [h.to(dev) for h in syn.people_heatmaps(B, V, J, 72, 96, (384, 288), seed=1)[0]]
    hms_b = [h.to(dev) for h in syn.people_heatmaps(B, V, J, 72, 96, (384, 288), seed=2)[0]]
    net = CuboidProposalNet(cfg)
    syn.fill_parameters_deterministic(net, seed=4, scale=0.05)
    net.eval().to(dev).use_channels_last(True)
    with torch.no_grad():
        ref_a = [t.clone() for t in net(hms_a, meta)]
        ref_b = [t.clone() for t in net(hms_b, meta)]
        meta2 = syn.make_meta(B, V, (384, 288), rotations=[5.0, -5.0], scale_mults=[1.1, 0.9], ssv_style=True)
        ref_c = [t.clone() for t in net(hms_b, meta2)]
    static = [h.clone() for h in hms_a]
    g = GraphedRootNet(net, static, meta)
    out = g()
    assert torch.equal(out[0], ref_a[0]) and torch.equal(out[1], ref_a[1])
    for s, h in zip(static, hms_b):                 # new frame written into the static input buffers
        s.copy_(h)
    out = g()
    assert torch.equal(out[0], ref_b[0]) and torch.equal(out[1], ref_b[1])
    # new calibration (different crop scale) reaches the kernels through the pinned camera table
    out2 = g(meta2)
    assert torch.equal(out2[0], ref_c[0]) and torch.equal(out2[1], ref_c[1]) and not torch.equal(out2[0], ref_b[0])


@pytest.mark.gpu
def test_synthetic_root_branch_kernels_match_reference_golden_and_torch_path():
    """sp3d_gaussian_target_3d / sp3d_render_root_heatmaps (§8 f3/f4) against the reference golden (fixed roots,
    no noise) and, for a B=3 random batch with per-sample crops, against the vectorised torch path on the CPU."""
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.cuboid_proposal_net_soft import CuboidProposalNetSoft
    from tests import golden_io as gio
    g = gio.load("rootnet_soft_synth")
    cfg = load_config(None, NETWORK__ROOTNET_ROOTHM=True, NETWORK__ROOTNET_TRAIN_SYNTH=True)
    cpu = CuboidProposalNetSoft(cfg)
    gpu = CuboidProposalNetSoft(cfg).to("cuda")
    cpu.noise_std = gpu.noise_std = 0.0
    u, uz, zn, lo, hi = g["u"], float(g["uz"]), g["zn"], g["lo"], g["hi"]
    x = (hi[0] - lo[0]) * torch.from_numpy(u[..., 0:1]) + lo[0]
    y = (hi[1] - lo[1]) * torch.from_numpy(u[..., 1:2]) + lo[1]
    z = ((hi[2] - lo[2]) * torch.full((1, 1, 1), uz) + lo[2]).expand(1, int(g["R"]), 1) + torch.from_numpy(zn) * 50
    roots = torch.cat([x, y, z], -1).float()
    target = gpu.target_cubes(roots.cuda()).cpu()
    assert float((target - torch.from_numpy(g["target"])).abs().max()) <= 1e-6
    meta = syn.make_meta(1, 5, (960, 512), ssv_style=True)
    meta[0]["trans"] = torch.from_numpy(g["trans"])
    hms = gpu.render_root_heatmaps(roots.cuda(), meta)
    for v in range(5):
        assert hms[v].shape == (1, 1, 128, 240)
        assert float((hms[v][0].cpu() - torch.from_numpy(g["hms"][v][0])).abs().max()) <= 2e-5
    # B=3, rotated / rescaled crops, no meta['trans'] (derived from centre/scale/rotation)
    meta3 = syn.make_meta(3, 5, (960, 512), rotations=[0.0, 20.0, -35.0], scale_mults=[1.0, 0.8, 1.2])
    r3 = cpu.sample_roots(3, "cpu", torch.Generator().manual_seed(4))
    t_c, t_g = cpu.target_cubes(r3), gpu.target_cubes(r3.cuda()).cpu()
    assert float((t_c - t_g).abs().max()) <= 1e-6
    h_c, h_g = cpu.render_root_heatmaps(r3, meta3), gpu.render_root_heatmaps(r3.cuda(), meta3)
    for a, b in zip(h_c, h_g):
        assert float((a - b.cpu()).abs().max()) <= 2e-5
    assert float(t_g.max()) > 0.5 and float(h_g[0].max()) > 0.0


@pytest.mark.gpu
def test_differentiable_joint_rendering_matches_reference_formula():
    """sp3d_render_joints_fwd/bwd against the reference's expression (multi_person_posenet_ssv.py:416-423:
    exp(-((xx - x)/3)^2/2 - ((yy - y)/3)^2/2), summed over people, clipped) evaluated by torch in float64, values and
    gradients w.r.t. the projected joints; ragged people counts per (view, sample)."""
    from selfpose3d_amd import _lib
    dev = torch.device("cuda:0")
    N, P, J, h, w = 6, 4, 15, 32, 48
    g = torch.Generator(device="cpu").manual_seed(17)
    kps = torch.stack([torch.rand((N, P, J), generator=g) * (w + 8) - 4, torch.rand((N, P, J), generator=g) * (h + 8) - 4], -1)
    kps[0, 1] = kps[0, 0] + 0.5                                    # overlapping people: the clip becomes active
    count = torch.tensor([4, 2, 0, 1, 3, 4], dtype=torch.int32)
    wgt = torch.randn((N, J, h, w), generator=g)
    k_gpu = kps.to(dev).requires_grad_(True)
    out = _lib.render_joint_heatmaps(k_gpu, count, h, w, 3.0)
    (out * wgt.to(dev)).sum().backward()
    k64 = kps.double().requires_grad_(True)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing="ij")
    ref = []
    for n in range(N):
        kk = k64[n, :int(count[n])]                                # (p, J, 2)
        x, y = kk[..., 0, None, None], kk[..., 1, None, None]
        hm = torch.exp(-(((xx - x) / 3.0) ** 2) / 2 - (((yy - y) / 3.0) ** 2) / 2)      # (p, J, h, w)
        ref.append(torch.clip(hm.sum(0), min=0.0, max=1.0) if count[n] > 0 else torch.zeros(J, h, w, dtype=torch.float64))
    ref = torch.stack(ref, 0)
    (ref * wgt.double()).sum().backward()
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 2e-6
    gk = k_gpu.grad.cpu().double()
    scale = max(1.0, float(k64.grad.abs().max()))
    assert float((gk - k64.grad).abs().max()) <= 2e-5 * scale
    assert float(gk[2].abs().max()) == 0.0 and float(gk[1, 2:].abs().max()) == 0.0     # people beyond the count get no gradient
    assert float(out.detach().max()) == 1.0                                             # the clip was exercised


@pytest.mark.gpu
def test_reprojection_heatmaps_gradient_reaches_the_3d_joints():
    """3D poses -> project_joints (torch, differentiable) -> sp3d_render_joints (HIP fwd/bwd): the MSE against target
    heat-maps has the same value and the same gradient w.r.t. the poses as the all-torch float64 evaluation."""
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.camera_pack import pack_cameras
    from selfpose3d_amd.reprojection import project_joints, reprojection_heatmaps
    dev = torch.device("cuda:0")
    B, V, P, J, h, w = 2, 5, 3, 15, 128, 240
    meta = syn.make_meta(B, V, (960, 512), ssv_style=True)
    cam = torch.from_numpy(pack_cameras(meta, B, (960, 512)))
    g = torch.Generator(device="cpu").manual_seed(29)
    joints = torch.stack([torch.rand((B, P, J), generator=g) * 3000 - 1500, torch.rand((B, P, J), generator=g) * 3000 - 2000,
                          torch.rand((B, P, J), generator=g) * 1600 + 100], -1)
    count = torch.tensor([3, 2], dtype=torch.int32)
    target = torch.rand((V, B, J, h, w), generator=g)
    jg = joints.to(dev).requires_grad_(True)
    hm = reprojection_heatmaps(jg, count, cam.to(dev), h, w)
    loss = torch.nn.functional.mse_loss(hm, target.to(dev))
    loss.backward()
    j64 = joints.double().requires_grad_(True)
    kps = project_joints(j64, cam.double(), 4.0)                                   # (V,B,P,J,2)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing="ij")
    ref = torch.zeros(V, B, J, h, w, dtype=torch.float64)
    rows = []
    for v in range(V):
        for b in range(B):
            kk = kps[v, b, :int(count[b])]
            x, y = kk[..., 0, None, None], kk[..., 1, None, None]
            rows.append(torch.clip(torch.exp(-(((xx - x) / 3.0) ** 2) / 2 - (((yy - y) / 3.0) ** 2) / 2).sum(0), 0.0, 1.0))
    ref = torch.stack(rows, 0).view(V, B, J, h, w)
    loss64 = torch.nn.functional.mse_loss(ref, target.double())
    loss64.backward()
    assert abs(float(loss.detach()) - float(loss64.detach())) <= 1e-6
    gs = max(1e-12, float(j64.grad.abs().max()))
    assert float((jg.grad.cpu().double() - j64.grad).abs().max()) <= 2e-3 * gs
    assert float(jg.grad[1, 2].abs().max()) == 0.0                                  # person beyond the count


@pytest.mark.gpu
def test_graphed_rootnet_follows_changing_meta_and_leaves_the_net_untouched():
    """GraphedRootNet: every replay consumes the camera table of ITS call (pinned ring + sp3d_fetch_ring as first graph
    node), also when the host runs several launches ahead; eager calls on the same net keep using their own meta."""
    from selfpose3d_amd import synthetic as syn
    from selfpose3d_amd.config import load_config
    from selfpose3d_amd.cuboid_proposal_net import CuboidProposalNet
    from selfpose3d_amd.graphs import GraphedRootNet
    dev = torch.device("cuda:0")
    cfg = load_config(None, NETWORK__IMAGE_SIZE=[384, 288], NETWORK__HEATMAP_SIZE=[96, 72],
                      MULTI_PERSON__INITIAL_CUBE_SIZE=[24, 24, 8])
    B, V, J = 2, 4, 15
    metas = [syn.make_meta(B, V, (384, 288)),
             syn.make_meta(B, V, (384, 288), rotations=[10.0, -20.0], scale_mults=[1.2, 0.9], ssv_style=True),
             syn.random_meta(B, V, (384, 288), seed=4, augment=True)]
    hms, _ = syn.people_heatmaps(B, V, J, 72, 96, (384, 288), seed=11)
    hms = [h.to(dev) for h in hms]
    net = CuboidProposalNet(cfg)
    syn.fill_parameters_deterministic(net, seed=5, scale=0.05)
    net.eval().to(dev)
    net.use_channels_last(True)
    with torch.no_grad():
        eager = [tuple(t.clone() for t in net(hms, m)) for m in metas]
        graphed = GraphedRootNet(net, hms, metas[0])
        # launch a burst without synchronising in between (host ahead of the GPU), meta changing every call
        outs = []
        for k in range(9):
            rc, gc = graphed(metas[k % 3])
            outs.append((k % 3, rc.clone(), gc.clone()))
        torch.cuda.synchronize()
        for which, rc, gc in outs:
            assert float((rc - eager[which][0]).abs().max()) <= 2e-4 * max(1.0, float(eager[which][0].abs().max())), which
            assert torch.equal(gc[:, :3, :3], eager[which][1][:, :3, :3]), which
        # the net itself is untouched: an eager call with another meta gives that meta's answer
        again = net(hms, metas[2])
        assert torch.equal(again[0], eager[2][0]) or float((again[0] - eager[2][0]).abs().max()) <= 1e-5
        assert net.project_layer._static_cam is None and net.project_layer.cache_packs is True
