#!/usr/bin/env python3
"""Round-6 golden vectors, by RUNNING THE REFERENCE'S OWN PYTHON on CPU (build container only; import shims of
make_goldens.py / make_goldens_r2.py).  What round 5's review found unpinned:

  unproj_grad_root_full.npz   reference ProjectLayer + autograd (lib/models/project_layer.py:42-102 through F.grid_sample)
                              at the size the root-grid backward kernel RUNS at: B=4, V=5, J=15, 240x128 -> 80x80x20,
                              augmented crops (rotation / scale per sample) + flip, heat-maps in [-0.7, 1.7) so that the
                              clamp at 0 and 1 (:99) blocks gradient on ~10 % of the voxels.  Stored: sub-sampled grad_hm,
                              float64 sum / |sum| / position-weighted sum per (view, sample, joint), the clamp-mask
                              population per (sample, joint), sub-sampled cubes + sums (the B=4 forward of configs[1]'s shape).
  unproj_grad_fine_full.npz   the same at the pose stage's size: four 64^3 person cubes (grid_center (B,5), one row invalid),
                              augmented crops + flip.
  unproj_coarse_b4.npz        forward only, BASELINE configs[1] exactly: B=4, plain validation crops, U[0,1) maps.
  render_ssv_full.npz         the rendered (V,B,J,128,240) heat-maps of the self-supervised loss and their gradient w.r.t. the
                              3D joints, taken OUT OF the reference's own forward (lib/models/multi_person_posenet_ssv.py:
                              433-465): F.mse_loss is wrapped to catch `heatmaps_all_21 / _12`, the pose net's outputs are
                              caught by a forward wrapper; full heat-map size, small nets (ResNet-18, 24x24x8 / 16^3 cubes).
  config_schema.json          key names per section of lib/core/config.py (names only - what `_update_dict` checks, :253-257).

    python tests/golden/make_goldens_r6.py [name ...]
"""
import json
import os
import sys
import time
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

WIDE = (2.4, -0.7)               # heat-map = U[0,1) * 2.4 - 0.7  (hm_kind "random_wide", rebuilt by tests/golden_io.Case)
GRAD_STRIDE = 53                 # every 53rd element of grad_hm (V,B,J,h,w) is stored
CUBE_STRIDE = 37


def pos_weights(shape, seed):
    """fixed pseudo-random weights over a plane: a sum weighted with them moves when a gradient lands on the wrong pixel"""
    return np.random.default_rng(seed).standard_normal(shape)


def run_unproject(name, B, V, J, img, hm, grid_size, grid_center, cube, hm_kind, seed, rotations=None, scale_mults=None,
                  flip=None, ssv_style=False, with_grad=True):
    from models.project_layer import ProjectLayer
    t0 = time.time()
    cfg = mg.make_cfg(img, hm, syn.SPACE_SIZE, syn.SPACE_CENTER, cube, grid_size, cube, J)
    layer = ProjectLayer(cfg)
    meta = syn.make_meta(B, V, img, rotations=rotations, scale_mults=scale_mults, ssv_style=ssv_style)
    hms = syn.random_heatmaps(B, V, J, hm[1], hm[0], seed=seed)
    if hm_kind == "random_wide":
        hms = [h * WIDE[0] + WIDE[1] for h in hms]
    if with_grad:
        hms = [h.clone().requires_grad_(True) for h in hms]
    gc_t = torch.from_numpy(grid_center) if isinstance(grid_center, np.ndarray) else grid_center
    flip_t = None if flip is None else torch.tensor(flip, dtype=torch.bool)
    cubes, grids = layer(hms, meta, list(grid_size), gc_t, list(cube), flip_xcoords=flip_t)
    c = cubes.detach().numpy()
    g = grids.detach().numpy()
    N = cube[0] * cube[1] * cube[2]
    idx = np.arange(0, N, CUBE_STRIDE)
    rec = dict(B=B, V=V, J=J, img=np.array(img), hm=np.array(hm), grid_size=np.array(grid_size, np.float64),
               cube=np.array(cube), hm_kind=hm_kind, seed=seed,
               grid_center=np.asarray(grid_center, np.float64) if not isinstance(grid_center, list) else np.array(grid_center, np.float64),
               center_is_list=isinstance(grid_center, list),
               rotations=np.array([] if rotations is None else rotations, np.float64),
               scale_mults=np.array([] if scale_mults is None else scale_mults, np.float64),
               flip=np.array([] if flip is None else flip, bool), ssv_style=ssv_style,
               hm_sum=np.array([float(h.detach().double().sum()) for h in hms]),
               cubes_sum=np.float64(c.astype(np.float64).sum()),
               cubes_sum_per_joint=c.astype(np.float64).sum(axis=(0, 2, 3, 4)),
               cubes_sum_per_sample_joint=c.astype(np.float64).sum(axis=(2, 3, 4)),
               grids_sum=g.astype(np.float64).sum(axis=(0, 1)),
               sub_idx=idx, cubes_sub=c.reshape(B, J, N)[:, :, idx], grids_sub=g[:, idx],
               # voxels the clamp leaves alone (0 < value < 1): where the gradient passes for certain
               cubes_interior=((c > 0) & (c < 1)).sum(axis=(2, 3, 4)).astype(np.int64),
               cubes_at_zero=(c == 0).sum(axis=(2, 3, 4)).astype(np.int64),
               cubes_at_one=(c == 1).sum(axis=(2, 3, 4)).astype(np.int64))
    if with_grad:
        wgt = torch.from_numpy(np.random.default_rng(seed + 101).standard_normal(c.shape).astype(np.float32))
        (cubes * wgt).sum().backward()
        gh = np.stack([h.grad.numpy() for h in hms])                       # (V,B,J,h,w)
        g64 = gh.astype(np.float64)
        pw = pos_weights((hm[1], hm[0]), seed + 7)
        flat = gh.reshape(-1)
        rec.update(grad_seed=seed + 101, pos_seed=seed + 7, grad_stride=GRAD_STRIDE,
                   grad_sub=flat[::GRAD_STRIDE].copy(),
                   grad_sum=g64.sum(axis=(3, 4)), grad_abs_sum=np.abs(g64).sum(axis=(3, 4)),
                   grad_pos_sum=(g64 * pw).sum(axis=(3, 4)), grad_absmax=np.abs(gh).max(axis=(3, 4)),
                   grad_nonzero=(gh != 0).sum(axis=(3, 4)).astype(np.int64),
                   # two whole planes, for a look at every pixel of something
                   grad_plane_v0_b0_j2=gh[0, 0, 2].copy(), grad_plane_vl_bl_jl=gh[V - 1, B - 1, J - 1].copy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print(f"{name}: cubes {c.shape} interior {int(rec['cubes_interior'].sum())} at0 {int(rec['cubes_at_zero'].sum())} "
          f"at1 {int(rec['cubes_at_one'].sum())} of {c.size}; {time.time() - t0:.1f} s; "
          f"{os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.2f} MB")


def g_unproj_grad_root_full():
    run_unproject("unproj_grad_root_full", 4, 5, 15, (960, 512), (240, 128), syn.SPACE_SIZE, [list(syn.SPACE_CENTER)],
                  tuple(syn.INITIAL_CUBE_SIZE), "random_wide", seed=601, rotations=[0.0, 30.0, -30.0, 12.5],
                  scale_mults=[1.0, 1.3, 0.8, 1.1], flip=[False, True, True, False], ssv_style=True)


def g_unproj_grad_fine_full():
    gc = np.array([[300.0, -800.0, 900.0, 0.0, 0.9],
                   [-1200.0, 400.0, 1000.0, 2.0, 0.7],
                   [0.0, 0.0, 0.0, -1.0, 0.1],                              # invalid: skipped (:54), zero cubes, no gradient
                   [900.0, -1500.0, 750.0, 1.0, 0.5]], np.float32)
    run_unproject("unproj_grad_fine_full", 4, 5, 15, (960, 512), (240, 128), syn.FINE_GRID_SIZE, gc, (64, 64, 64),
                  "random_wide", seed=611, rotations=[15.0, -20.0, 0.0, 40.0], scale_mults=[1.1, 0.9, 1.0, 1.25],
                  flip=[True, False, False, True], ssv_style=True)


def g_unproj_coarse_b4():
    run_unproject("unproj_coarse_b4", 4, 5, 15, (960, 512), (240, 128), syn.SPACE_SIZE, [list(syn.SPACE_CENTER)],
                  tuple(syn.INITIAL_CUBE_SIZE), "random", seed=0, with_grad=False)


# ---------------------------------------------------------------------------------------------------------------------
RENDER = dict(img=(960, 512), hm=(240, 128), V=3, J=15, cube=(24, 24, 8), fine_cube=(16, 16, 16), max_people=4,
              layers=18, threshold=0.0, sigma=3, B=2, param_seed=197, data_seed=9)


def g_render_ssv_full():
    """the reference's SSV train forward with its rendering caught in flight"""
    import models.multi_person_posenet_ssv as mps
    import torch.nn.functional as F
    t0 = time.time()
    r = RENDER
    cfg = mg2.full_cfg(r["img"], r["hm"], r["cube"], r["fine_cube"], r["J"], num_layers=r["layers"], batch=r["B"], SIGMA=r["sigma"],
                       ROOTNET_ROOTHM=True, ROOTNET_TRAIN_SYNTH=True, FREEZE_ROOTNET=True, TRAIN_BACKBONE=True)
    cfg.MULTI_PERSON.MAX_PEOPLE_NUM = r["max_people"]
    cfg.MULTI_PERSON.THRESHOLD = r["threshold"]
    cfg.MODEL = "multi_person_posenet_ssv"
    cfg.WITH_ATTN, cfg.ATTN_WEIGHT, cfg.ATTN_NUM_LAYERS = True, 0.1, 18
    cfg.USE_L1, cfg.L1_WEIGHT, cfg.L1_ATTN = False, 0.01, False
    cfg.TRAIN.L1_EPOCH = 5
    model = mps.get_multi_person_pose_net(cfg, is_train=True)
    gio.he_fill(model, seed=r["param_seed"])
    model.train()
    model.root_net.eval()
    batch = gio.train_batch(gio.render_cfg(), B=r["B"], seed=r["data_seed"], ssv=True)
    (in1, t1, w1, d1, m1, _, in2, t2, w2, d2, m2, _, in3, t3, w3, d3, m3, _) = batch

    caught = {"maps": [], "pose": []}
    real_mse = F.mse_loss

    def mse_spy(inp, tgt, *a, **k):
        # F.mse_loss(targets_2d, heatmaps_all_XX, ...) (:469,475): the rendered stack is the SECOND argument, (V,B,J,h,w)
        # with a graph (the attention regulariser passes its maps first, :483)
        if torch.is_tensor(tgt) and tgt.dim() == 5 and tgt.requires_grad:
            if k.get("reduction") == "none":                                 # WITH_ATTN: the per-pixel form (:467-476)
                caught["maps"].append(tgt)
        return real_mse(inp, tgt, *a, **k)

    real_pose = model.pose_net.forward

    def pose_spy(hms, meta, gc, **k):
        out = real_pose(hms, meta, gc, **k)
        caught["pose"].append((1 if meta is m1 else 2, out))
        return out

    mps.F.mse_loss = mse_spy
    model.pose_net.forward = pose_spy
    try:
        _, _, gc, losses = model(views1=in1, meta1=m1, targets_2d1=t1, weights_2d1=w1, targets_3d1=d1[0],
                                 views2=in2, meta2=m2, targets_2d2=t2, weights_2d2=w2, targets_3d2=d2[0],
                                 views3=in3, meta3=m3, targets_2d3=t3, weights_2d3=w3, targets_3d3=d3[0], epoch=1)
    finally:
        mps.F.mse_loss = real_mse
    maps21, maps12 = caught["maps"]                                          # :469 then :475 (set 1's loss first)
    V, B, J, h, w = maps21.shape
    assert (V, B, J, h, w) == (r["V"], r["B"], r["J"], r["hm"][1], r["hm"][0]) and maps12.shape == maps21.shape
    count = (gc[:, :, 3] >= 0).sum(1)
    P = int(gc.shape[1])
    slots = {1: [], 2: []}
    for which, out in caught["pose"]:
        slots[which].append(out)
    assert len(slots[1]) == len(slots[2]) and 1 <= len(slots[1]) <= P
    rng = np.random.default_rng(r["data_seed"] + 301)
    w21 = torch.from_numpy(rng.standard_normal(tuple(maps21.shape)).astype(np.float32))
    w12 = torch.from_numpy(rng.standard_normal(tuple(maps12.shape)).astype(np.float32))
    grads = torch.autograd.grad((maps21 * w21).sum() + (maps12 * w12).sum(), slots[1] + slots[2], allow_unused=True)
    n = len(slots[1])

    def table(outs, gs):
        j = np.zeros((B, P, J, 3), np.float32)
        g = np.zeros((B, P, J, 3), np.float32)
        for s, (o, gg) in enumerate(zip(outs, gs)):
            j[:, s] = o.detach().numpy()
            if gg is not None:
                g[:, s] = gg.numpy()
        return j, g
    joints1, g1 = table(slots[1], grads[:n])                                 # set 1's poses are rendered into maps12
    joints2, g2 = table(slots[2], grads[n:])
    rec = dict(param_seed=r["param_seed"], data_seed=r["data_seed"], weight_seed=r["data_seed"] + 301, count=count.numpy(),
               grid_centers=gc.detach().numpy(), joints1=joints1, joints2=joints2, grad_joints1=g1, grad_joints2=g2,
               trans1=m1[0]["trans"].numpy(), trans2=m2[0]["trans"].numpy())
    pw = pos_weights((h, w), r["data_seed"] + 7)
    for nm, mp_ in (("maps21", maps21), ("maps12", maps12)):
        a = mp_.detach().numpy()
        a64 = a.astype(np.float64)
        rec[nm + "_sum"] = a64.sum(axis=(3, 4))
        rec[nm + "_pos_sum"] = (a64 * pw).sum(axis=(3, 4))
        rec[nm + "_max"] = a.max(axis=(3, 4))
        rec[nm + "_at_one"] = (a == 1).sum(axis=(3, 4)).astype(np.int64)
        rec[nm + "_full_b0"] = a[:, 0][:, [0, 2, J - 1]].copy()             # every pixel of 3 joints x V views of sample 0
        rec[nm + "_sub"] = a.reshape(-1)[::GRAD_STRIDE].copy()
    rec["pos_seed"] = r["data_seed"] + 7
    rec["sub_stride"] = GRAD_STRIDE
    np.savez_compressed(os.path.join(HERE, "render_ssv_full.npz"), **rec)
    print(f"render_ssv_full: maps {tuple(maps21.shape)} count {count.tolist()} slots {n} max {float(maps21.max()):.3f} "
          f"at_one {int(rec['maps21_at_one'].sum())} |g1| {np.abs(g1).max():.3e} |g2| {np.abs(g2).max():.3e}; "
          f"{time.time() - t0:.1f} s; {os.path.getsize(os.path.join(HERE, 'render_ssv_full.npz')) / 1e6:.2f} MB")


def g_train_step_full():
    """reference MultiPersonPoseNet.forward in TRAIN mode at BASELINE configs[2]'s sizes, proposals from ground truth
    (lib/models/multi_person_posenet.py:36-102 with USE_GT): losses, the gradient of backbone.final_layer.weight and of the
    pose net's output layer, in fp32 and in a float64 rerun (the yardstick, as g_train_step of make_goldens_r3.py)."""
    import models.multi_person_posenet as mp
    import utils.cameras as ref_cameras
    import models.project_layer as ref_pl
    t0 = time.time()
    t = gio.TRAIN_FULL
    cfg = mg2.full_cfg(t["img"], t["hm"], t["cube"], t["fine_cube"], t["J"], num_layers=t["layers"], batch=2, SIGMA=t["sigma"],
                       USE_GT=True)
    cfg.MULTI_PERSON.MAX_PEOPLE_NUM = t["max_people"]
    cfg.MULTI_PERSON.THRESHOLD = t["threshold"]
    rec = {"param_seed": 183, "data_seed": 6}

    def run(double):
        model = mp.get_multi_person_pose_net(cfg, is_train=True)
        gio.he_fill(model, seed=rec["param_seed"])
        inputs, t2d, w2d, t3d, meta, _ = gio.train_batch(gio.train_full_cfg(USE_GT=True), B=2, seed=rec["data_seed"])
        if double:
            model.double()
            dd = lambda x: [v.double() for v in x]
            inputs, t2d, w2d, t3d = dd(inputs), dd(t2d), dd(w2d), dd(t3d)
        model.train()
        pred, hms, gc, l2d, l3d, lcord = model(views=inputs, meta=meta, targets_2d=t2d, weights_2d=w2d, targets_3d=t3d[0])
        (l2d.mean() + l3d.mean() + lcord.mean()).backward()
        return model, pred, hms, gc, l2d, l3d, lcord

    model, pred, hms, gc, l2d, l3d, lcord = run(False)
    print(f"  fp32 pass {time.time() - t0:.0f} s: losses", float(l2d), float(l3d), float(lcord), "valid", int((gc[:, :, 3] >= 0).sum()))
    rec.update(loss_2d=float(l2d), loss_3d=float(l3d), loss_cord=float(lcord), grid_centers=gc.detach().numpy(),
               pred=pred.detach().numpy(), hm_sum=np.array([float(h.double().sum()) for h in hms]),
               grad_final=model.backbone.final_layer.weight.grad.numpy().copy(),
               grad_pose_out=model.pose_net.v2v_net.output_layer.weight.grad.numpy().copy(),
               grad_conv1_sub=model.backbone.conv1.weight.grad.numpy().reshape(-1)[::7].copy())
    del model
    unfold32, xform32 = ref_cameras.unfold_camera_param, ref_pl.do_transform
    try:
        ref_pl.do_transform = lambda pts, tt: xform32(pts, tt.to(pts.dtype))
        ref_cameras.unfold_camera_param = lambda cam, device=None: tuple(v.double() for v in unfold32(cam, device))
        torch.set_default_dtype(torch.float64)
        m64, _, _, _, a2, a3, ac = run(True)
        rec.update(loss_2d_f64=float(a2), loss_3d_f64=float(a3), loss_cord_f64=float(ac),
                   grad_final_f64=m64.backbone.final_layer.weight.grad.numpy().copy(),
                   grad_pose_out_f64=m64.pose_net.v2v_net.output_layer.weight.grad.numpy().copy(),
                   grad_conv1_sub_f64=m64.backbone.conv1.weight.grad.numpy().reshape(-1)[::7].copy())
    finally:
        torch.set_default_dtype(torch.float32)
        ref_cameras.unfold_camera_param, ref_pl.do_transform = unfold32, xform32
    np.savez_compressed(os.path.join(HERE, "train_step_full.npz"), **rec)
    print(f"train_step_full: {time.time() - t0:.0f} s, {os.path.getsize(os.path.join(HERE, 'train_step_full.npz')) / 1e6:.2f} MB")


def g_config_schema():
    easydict = type(sys)("easydict")
    easydict.EasyDict = mg.AD
    sys.modules.setdefault("easydict", easydict)
    import core.config as rc
    out = {k: (sorted(v.keys()) if isinstance(v, dict) else None) for k, v in rc.config.items()}
    with open(os.path.join(HERE, "config_schema.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("config_schema:", len(out), "top-level names,", sum(len(v) for v in out.values() if v), "section keys")


ALL = {"train_step_full": g_train_step_full, "config_schema": g_config_schema, "unproj_coarse_b4": g_unproj_coarse_b4,
       "unproj_grad_root_full": g_unproj_grad_root_full, "unproj_grad_fine_full": g_unproj_grad_fine_full,
       "render_ssv_full": g_render_ssv_full}

if __name__ == "__main__":
    mg2.install_shims()
    torch.nn.Module.cuda = lambda self, device=None: self        # no GPU here; the reference calls .cuda() on loss modules
    torch.set_num_threads(8)
    for nm in (sys.argv[1:] or [k for k in ALL if k != "train_step_full"]):      # (train_step_full: ~30 min of CPU, run by name)
        ALL[nm]()
