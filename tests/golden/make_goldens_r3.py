#!/usr/bin/env python3
"""Round-3 golden vectors, by RUNNING THE REFERENCE'S OWN PYTHON on CPU (build container only; import shims of
make_goldens.py / make_goldens_r2.py; ``nn.Module.cuda`` is made a no-op because the reference moves its parameter-less
loss modules with ``.cuda()``, lib/models/multi_person_posenet.py:50,81):

  rootnet_48.npz       reference CuboidProposalNet -> V2VNet -> nms at the OTHER shipped grid, 48x48x12
                       (configs/panoptic/resnet50/prn32_cpn48x48x12_960x512_cam5.yaml): pins the generic inference-plan
                       path (no z-DFT / 88x88 plane kernels, other Winograd shapes)
  rootnet_160.npz      the same at BASELINE configs[3]: 10 views, 160x160x40 (the stress grid: generic plan path at 8x the voxels)
  train_step.npz       reference MultiPersonPoseNet.forward in TRAIN mode on a small scene (lib/models/
                       multi_person_posenet.py:36-102): loss_2d / loss_3d / loss_cord and the gradient of
                       backbone.final_layer.weight, with proposals from the root net and from ground truth (USE_GT)
  ssv_train_step.npz   reference MultiPersonPoseNetSSV.forward in TRAIN mode (pose-net stage: frozen root net,
                       attention net, two augmented view sets; lib/models/multi_person_posenet_ssv.py:197-501): every loss
                       term and two gradients

Inputs come from this repo's synthetic datasets (tests/golden_io.py train_batch), so the tests rebuild them exactly.

    python tests/golden/make_goldens_r3.py [name ...]
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")

import make_goldens as mg        # noqa: E402
import make_goldens_r2 as mg2    # noqa: E402
import golden_io as gio          # noqa: E402
from selfpose3d_amd import synthetic as syn   # noqa: E402


def g_rootnet_48():
    from models.cuboid_proposal_net import CuboidProposalNet
    from core.proposal import nms
    img, hm, V, J, cube = (960, 512), (240, 128), 5, 15, (48, 48, 12)
    cfg = mg.make_cfg(img, hm, syn.SPACE_SIZE, syn.SPACE_CENTER, cube, syn.FINE_GRID_SIZE, (32, 32, 32), J)
    net = CuboidProposalNet(cfg)
    syn.fill_parameters_deterministic(net, seed=171, scale=0.05)
    net.eval()
    meta = syn.make_meta(2, V, img)
    hms = mg2.mixed_heatmaps(V, J, hm[1], hm[0], img, seed=173)
    with torch.no_grad():
        root_cubes, grid_centers = net(hms, meta)
        vals, idx = nms(root_cubes, 10)
    rc = root_cubes.numpy()
    N = rc[0].size
    sub = np.arange(0, N, 7)
    np.savez_compressed(os.path.join(HERE, "rootnet_48.npz"), img=np.array(img), hm=np.array(hm), V=V, J=J, cube=np.array(cube),
                        hm_seed=173, param_seed=171, param_scale=0.05,
                        hm_sum=np.array([float(h.double().sum()) for h in hms]), sub_idx=sub,
                        root_sub=rc.reshape(2, N)[:, sub], root_sum=rc.astype(np.float64).sum(axis=(1, 2, 3)),
                        root_abs_sum=np.abs(rc.astype(np.float64)).sum(axis=(1, 2, 3)), nms_vals=vals.numpy(),
                        nms_idx=idx.numpy(), grid_centers=grid_centers.numpy())
    print("rootnet_48: root range", float(rc.min()), float(rc.max()), "top vals", vals.numpy()[:, :5])


def g_rootnet_160():
    """BASELINE configs[3] (the stress configuration): 10 views, 160x160x40 root grid, B=2"""
    from models.cuboid_proposal_net import CuboidProposalNet
    from core.proposal import nms
    img, hm, V, J, cube = (960, 512), (240, 128), 10, 15, (160, 160, 40)
    cfg = mg.make_cfg(img, hm, syn.SPACE_SIZE, syn.SPACE_CENTER, cube, syn.FINE_GRID_SIZE, (32, 32, 32), J)
    net = CuboidProposalNet(cfg)
    syn.fill_parameters_deterministic(net, seed=271, scale=0.05)
    net.eval()
    meta = syn.make_meta(2, V, img)
    hms = mg2.mixed_heatmaps(V, J, hm[1], hm[0], img, seed=273)
    with torch.no_grad():
        root_cubes, grid_centers = net(hms, meta)
        vals, idx = nms(root_cubes, 10)
    rc = root_cubes.numpy()
    N = rc[0].size
    sub = np.arange(0, N, 97)
    np.savez_compressed(os.path.join(HERE, "rootnet_160.npz"), img=np.array(img), hm=np.array(hm), V=V, J=J, cube=np.array(cube),
                        hm_seed=273, param_seed=271, param_scale=0.05,
                        hm_sum=np.array([float(h.double().sum()) for h in hms]), sub_idx=sub,
                        root_sub=rc.reshape(2, N)[:, sub], root_sum=rc.astype(np.float64).sum(axis=(1, 2, 3)),
                        root_abs_sum=np.abs(rc.astype(np.float64)).sum(axis=(1, 2, 3)), nms_vals=vals.numpy(),
                        nms_idx=idx.numpy(), grid_centers=grid_centers.numpy())
    print("rootnet_160: root range", float(rc.min()), float(rc.max()), "top vals", vals.numpy()[:, :5])


def _ref_cfg(ssv=False, **net):
    t = gio.TRAIN_SMALL
    cfg = mg2.full_cfg(t["img"], t["hm"], t["cube"], t["fine_cube"], t["J"], num_layers=t["layers"], batch=2,
                       SIGMA=t["sigma"], **net)
    cfg.MULTI_PERSON.MAX_PEOPLE_NUM = t["max_people"]
    cfg.MULTI_PERSON.THRESHOLD = t["threshold"]
    if ssv:
        cfg.MODEL = "multi_person_posenet_ssv"
        cfg.WITH_ATTN, cfg.ATTN_WEIGHT, cfg.ATTN_NUM_LAYERS = True, 0.1, 18
        cfg.USE_L1, cfg.L1_WEIGHT, cfg.L1_ATTN = True, 0.01, True
        cfg.TRAIN.L1_EPOCH = 0
    return cfg


def _to_float(meta_sets):
    return meta_sets


def g_train_step():
    import models.multi_person_posenet as mp
    rec = {}
    for tag, use_gt in (("net", False), ("gt", True)):
        cfg = _ref_cfg(USE_GT=use_gt)
        model = mp.get_multi_person_pose_net(cfg, is_train=True)
        gio.he_fill(model, seed=181)
        model.train()
        inputs, t2d, w2d, t3d, meta, _ = gio.train_batch(gio.train_cfg(USE_GT=use_gt), B=2, seed=5)
        pred, hms, gc, l2d, l3d, lcord = model(views=inputs, meta=meta, targets_2d=t2d, weights_2d=w2d, targets_3d=t3d[0])
        # per-term gradients first (they localise a mismatch), then the total as the training loop forms it
        fl = model.backbone.final_layer.weight
        for nm, term in (("2d", l2d), ("3d", l3d), ("cord", lcord)):
            if term.requires_grad:
                gt_, = torch.autograd.grad(term.mean(), fl, retain_graph=True, allow_unused=True)
                rec[f"{tag}_grad_final_{nm}"] = np.zeros(tuple(fl.shape), np.float32) if gt_ is None else gt_.numpy().copy()
        if not use_gt:
            ol = model.root_net.v2v_net.output_layer.weight
            fc = model.root_net.v2v_net.front_layers[0].block[0].weight
            ga, gb = torch.autograd.grad(l3d.mean(), (ol, fc), retain_graph=True)
            rec[f"{tag}_grad_root_out"], rec[f"{tag}_grad_root_front"] = ga.numpy().copy(), gb.numpy().copy()
        # float64 rerun of the same reference model: the yardstick for how exact ANY fp32 gradient of this net can be
        # (train-mode BatchNorm over a deep conv stack: fp32 backward passes differ from each other by ~0.5 %)
        import utils.cameras as ref_cameras
        unfold32 = ref_cameras.unfold_camera_param
        import models.project_layer as ref_pl
        xform32 = ref_pl.do_transform
        try:
            ref_pl.do_transform = lambda pts, t: xform32(pts, t.to(pts.dtype))      # (its affine is cast to fp32 too, :69-72)
            torch.set_default_dtype(torch.float64)
            # (the reference casts camera entries to fp32 explicitly, lib/utils/cameras.py:13-24: lift them for this pass)
            ref_cameras.unfold_camera_param = lambda cam, device=None: tuple(t.double() for t in unfold32(cam, device))
            m64 = mp.get_multi_person_pose_net(cfg, is_train=True)
            gio.he_fill(m64, seed=181)
            m64.double().train()
            dd = lambda x: [t.double() for t in x]
            _, _, _, a2, a3, ac = m64(views=dd(inputs), meta=meta, targets_2d=dd(t2d), weights_2d=dd(w2d),
                                      targets_3d=t3d[0].double())
            fl64 = m64.backbone.final_layer.weight
            for nm, term in (("2d", a2), ("3d", a3), ("cord", ac)):
                if term.requires_grad:
                    gt_, = torch.autograd.grad(term.mean(), fl64, retain_graph=True, allow_unused=True)
                    if gt_ is not None:
                        rec[f"{tag}_grad_final_{nm}_f64"] = gt_.numpy().astype(np.float64)
            if not use_gt:
                ga64, gb64 = torch.autograd.grad(a3.mean(), (m64.root_net.v2v_net.output_layer.weight,
                                                             m64.root_net.v2v_net.front_layers[0].block[0].weight),
                                                 retain_graph=True)
                rec[f"{tag}_grad_root_out_f64"], rec[f"{tag}_grad_root_front_f64"] = ga64.numpy().copy(), gb64.numpy().copy()
            (a2.mean() + a3.mean() + ac.mean()).backward()
            rec[f"{tag}_grad_final_f64"] = fl64.grad.numpy().astype(np.float64)
            gp64 = m64.pose_net.v2v_net.output_layer.weight.grad
            if gp64 is not None:
                rec[f"{tag}_grad_pose_out_f64"] = gp64.numpy().astype(np.float64)
            print("  float64 rerun: losses", float(a2), float(a3), float(ac))
        finally:
            torch.set_default_dtype(torch.float32)
            ref_cameras.unfold_camera_param = unfold32
            ref_pl.do_transform = xform32
        loss = l2d.mean() + l3d.mean() + lcord.mean()
        loss.backward()
        g = model.backbone.final_layer.weight.grad
        gp = model.pose_net.v2v_net.output_layer.weight.grad
        gp = torch.zeros_like(model.pose_net.v2v_net.output_layer.weight) if gp is None else gp    # pose net not reached
        rec.update({f"{tag}_loss_2d": float(l2d), f"{tag}_loss_3d": float(l3d), f"{tag}_loss_cord": float(lcord),
                    f"{tag}_grad_final": g.numpy().copy(), f"{tag}_grid_centers": gc.detach().numpy(),
                    f"{tag}_pred": pred.detach().numpy(), f"{tag}_hm_sum": np.array([float(h.double().sum()) for h in hms]),
                    f"{tag}_grad_pose_out": gp.numpy().copy()})
        print("train_step", tag, "losses", float(l2d), float(l3d), float(lcord), "valid", int((gc[:, :, 3] >= 0).sum()),
              "|grad|", float(g.abs().max()))
    rec["param_seed"], rec["data_seed"] = 181, 5
    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **rec)


def _lift(x):
    """every floating tensor of a (nested) batch structure -> float64"""
    if torch.is_tensor(x):
        return x.double() if x.is_floating_point() else x
    if isinstance(x, dict):
        return {k: _lift(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_lift(v) for v in x)
    return x


def _ssv_float64_rerun(mps, cfg, gc32):
    """float64 rerun of the same reference SSV step (lib/models/multi_person_posenet_ssv.py:197-501): the yardstick for how
    exact ANY fp32 gradient of this step can be, exactly as g_train_step does for the supervised step.  Runs AFTER the fp32
    pass (which it leaves bit-for-bit unchanged) on a freshly rebuilt batch: the reference's l1_matching_loss normalises
    meta['joints'] in place (:166-169), so the fp32 pass's batch is spent."""
    import utils.cameras as ref_cameras
    import models.project_layer as ref_pl
    import models.cuboid_proposal_net_soft as ref_soft
    unfold32, xform32 = ref_cameras.unfold_camera_param, ref_pl.do_transform
    out = {}
    try:
        ref_pl.do_transform = lambda pts, t: xform32(pts, t.to(pts.dtype))
        ref_cameras.unfold_camera_param = lambda cam, device=None: tuple(t.double() for t in unfold32(cam, device))
        torch.set_default_dtype(torch.float64)
        m64 = mps.get_multi_person_pose_net(cfg, is_train=True)
        gio.he_fill(m64, seed=191)
        m64.double().train()
        m64.root_net.eval()
        b = gio.train_batch(gio.train_cfg(ssv=True), B=2, seed=7, ssv=True)
        (in1, t1, w1, d1, m1, _, in2, t2, w2, d2, m2, _, in3, t3, w3, d3, m3, _) = _lift(b)
        _, _, gc64, l64 = m64(views1=in1, meta1=m1, targets_2d1=t1, weights_2d1=w1, targets_3d1=d1[0],
                              views2=in2, meta2=m2, targets_2d2=t2, weights_2d2=w2, targets_3d2=d2[0],
                              views3=in3, meta3=m3, targets_2d3=t3, weights_2d3=w3, targets_3d3=d3[0], epoch=1)
        # the float64 root net must propose the same cubes, or the two runs are not the same function
        # (same voxel maxima and validity flags; the mm coordinates differ by the fp32 rounding of get_real_loc)
        g64, g32 = gc64.detach().numpy(), gc32.numpy().astype(np.float64)
        same = bool(np.array_equal(g64[:, :, 3] >= 0, g32[:, :, 3] >= 0)
                    and np.abs(g64[:, :, :3] - g32[:, :, :3])[g32[:, :, 3] >= 0].max() < 1e-2)
        sum(v.mean() for v in l64.values() if v.requires_grad).backward()
        out = {"grad_final_f64": m64.backbone.final_layer.weight.grad.numpy().copy(),
               "grad_pose_out_f64": m64.pose_net.v2v_net.output_layer.weight.grad.numpy().copy(),
               "grad_attn_final_f64": m64.attn.backbone.final_layer.weight.grad.numpy().copy(),
               "f64_same_proposals": same, "grid_centers_f64": g64}
        out.update({"loss_f64_" + k: float(v.mean()) for k, v in l64.items()})
        print("  float64 rerun: same proposals", same, {k: float(v.mean()) for k, v in l64.items()})
    finally:
        torch.set_default_dtype(torch.float32)
        ref_cameras.unfold_camera_param = unfold32
        ref_pl.do_transform = xform32
    return out


def g_ssv_train_step():
    import models.multi_person_posenet_ssv as mps
    cfg = _ref_cfg(ssv=True, ROOTNET_ROOTHM=True, ROOTNET_TRAIN_SYNTH=True, FREEZE_ROOTNET=True, TRAIN_BACKBONE=True)
    model = mps.get_multi_person_pose_net(cfg, is_train=True)
    gio.he_fill(model, seed=191)
    model.train()
    model.root_net.eval()                                   # lib/core/function.py:46-48 (FREEZE_ROOTNET)
    b = gio.train_batch(gio.train_cfg(ssv=True), B=2, seed=7, ssv=True)
    (in1, t1, w1, d1, m1, _, in2, t2, w2, d2, m2, _, in3, t3, w3, d3, m3, _) = b
    pred, hm3, gc, losses = model(views1=in1, meta1=m1, targets_2d1=t1, weights_2d1=w1, targets_3d1=d1[0],
                                  views2=in2, meta2=m2, targets_2d2=t2, weights_2d2=w2, targets_3d2=d2[0],
                                  views3=in3, meta3=m3, targets_2d3=t3, weights_2d3=w3, targets_3d3=d3[0], epoch=1)
    total = sum(v.mean() for v in losses.values() if v.requires_grad)
    total.backward()
    rec = {"loss_" + k: float(v.mean()) for k, v in losses.items()}
    rec.update(keys=np.array(sorted(losses)), grid_centers=gc.detach().numpy(), pred=pred.detach().numpy(),
               grad_final=model.backbone.final_layer.weight.grad.numpy().copy(),
               grad_pose_out=model.pose_net.v2v_net.output_layer.weight.grad.numpy().copy(),
               grad_attn_final=model.attn.backbone.final_layer.weight.grad.numpy().copy(),
               param_seed=191, data_seed=7, epoch=1)
    rec.update(_ssv_float64_rerun(mps, cfg, gc.detach()))
    np.savez_compressed(os.path.join(HERE, "ssv_train_step.npz"), **rec)
    print("ssv_train_step:", {k: float(v.mean()) for k, v in losses.items()}, "valid", int((gc[:, :, 3] >= 0).sum()))


ALL = {"rootnet_48": g_rootnet_48, "rootnet_160": g_rootnet_160, "train_step": g_train_step, "ssv_train_step": g_ssv_train_step}

if __name__ == "__main__":
    mg2.install_shims()
    torch.nn.Module.cuda = lambda self, device=None: self        # no GPU here; the reference calls .cuda() on loss modules
    torch.set_num_threads(8)
    for n in (sys.argv[1:] or list(ALL)):
        ALL[n]()
