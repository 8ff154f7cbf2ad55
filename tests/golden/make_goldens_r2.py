#!/usr/bin/env python3
"""Round-2 golden vectors, again by RUNNING THE REFERENCE'S OWN PYTHON on CPU (build container only; same
import shims as make_goldens.py plus ``torchvision`` / ``easydict`` stubs for the SSV model's module-level imports):

  rootnet_full.npz        reference CuboidProposalNet at the BENCHMARKED size (B=2, J=15, 240x128, 80x80x20,
                          non-degenerate weights): root cubes (sub-sampled + float64 sums), NMS top-k, grid centres
  softargmax.npz          reference SoftArgmaxLayer (pose_regression_net.py:19-28)
  pose_resnet.npz         reference PoseResNet-50 / attention ResNet-18 on a (1,3,64,64) input, deterministic weights
  state_dict_keys.json    key -> shape of the reference MultiPersonPoseNet / MultiPersonPoseNetSSV state_dicts
  ssv_inference.npz       reference MultiPersonPoseNetSSV.do_inference on a small synthetic scene

    python tests/golden/make_goldens_r2.py [name ...]
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

import make_goldens as mg   # noqa: E402  (shims, AD, make_cfg)
from selfpose3d_amd import synthetic as syn   # noqa: E402

AD = mg.AD


def install_shims():
    mg._install_shims()
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.save_image = lambda *a, **k: None
    tv.utils = tvu
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.utils"] = tvu


def full_cfg(image_size, heatmap_size, cube, fine_cube, num_joints, num_layers=50, batch=1, **net):
    """attribute-dict with every key the reference's top-level models read (core/config.py defaults)"""
    network = dict(IMAGE_SIZE=list(image_size), HEATMAP_SIZE=list(heatmap_size), NUM_JOINTS=num_joints, BETA=100.0,
                   ROOTNET_ROOTHM=False, ROOTNET_TRAIN_SYNTH=False, USE_GT=False, TRAIN_ONLY_2D=False,
                   TRAIN_ONLY_ROOTNET=False, FREEZE_ROOTNET=False, SINGLE_AUG_TRAINING_POSENET=False,
                   ROOT_CONSISTENCY_LOSS=True, WEIGHT_ROOT_SYN=100.0, WEIGHT_ROOT_REG=1.0, INIT_TRAIN_EPOCHS_ROOTNET=0,
                   ROOTNET_SYN_RANGE=[[2500.0, -2000.0], [1500.0, -1500.0], [250.0, -300.0]], PRETRAINED="", SIGMA=3)
    network.update(net)
    return AD(
        BACKBONE_MODEL="pose_resnet", MODEL="multi_person_posenet", WITH_ATTN=False, ATTN_WEIGHT=0.1, ATTN_NUM_LAYERS=18,
        USE_L1=False, L1_WEIGHT=0.1, L1_ATTN=False, EVAL_ROOTNET_ONLY=False, COCO_TO_PANOPTIC_MAPPING=list(range(15)),
        NETWORK=AD(**network),
        POSE_RESNET=AD(NUM_LAYERS=num_layers, DECONV_WITH_BIAS=False, NUM_DECONV_LAYERS=3, NUM_DECONV_FILTERS=[256, 256, 256],
                       NUM_DECONV_KERNELS=[4, 4, 4], FINAL_CONV_KERNEL=1),
        MULTI_PERSON=AD(SPACE_SIZE=list(syn.SPACE_SIZE), SPACE_CENTER=list(syn.SPACE_CENTER), INITIAL_CUBE_SIZE=list(cube),
                        MAX_PEOPLE_NUM=10, THRESHOLD=0.3),
        PICT_STRUCT=AD(GRID_SIZE=list(syn.FINE_GRID_SIZE), CUBE_SIZE=list(fine_cube)),
        DATASET=AD(ROOTIDX=2, ROOTIDX_PSEUDO=2, TEST_DATASET="panoptic", TRAIN_DATASET="panoptic"),
        TRAIN=AD(BATCH_SIZE=batch, L1_EPOCH=5),
    )


def mixed_heatmaps(V, J, h, w, img, seed):
    """B=2: sample 0 uniform-random maps (scaled to 0.35: keeps the fused volume below the clamp), sample 1 'people'"""
    rnd = syn.random_heatmaps(2, V, J, h, w, seed=seed)
    ppl, _ = syn.people_heatmaps(2, V, J, h, w, img, seed=seed + 1)
    return [torch.stack([0.35 * rnd[v][0], ppl[v][1]]) for v in range(V)]


def g_rootnet_full():
    from models.cuboid_proposal_net import CuboidProposalNet
    from core.proposal import nms
    img, hm, V, J = (960, 512), (240, 128), 5, 15
    cfg = mg.make_cfg(img, hm, syn.SPACE_SIZE, syn.SPACE_CENTER, syn.INITIAL_CUBE_SIZE, syn.FINE_GRID_SIZE, (64, 64, 64), J)
    net = CuboidProposalNet(cfg)
    syn.fill_parameters_deterministic(net, seed=71, scale=0.05)
    net.eval()
    meta = syn.make_meta(2, V, img)
    hms = mixed_heatmaps(V, J, hm[1], hm[0], img, seed=73)
    with torch.no_grad():
        root_cubes, grid_centers = net(hms, meta)
        vals, idx = nms(root_cubes, 10)
    rc = root_cubes.numpy()
    N = rc[0].size
    sub = np.arange(0, N, 37)
    np.savez_compressed(os.path.join(HERE, "rootnet_full.npz"), img=np.array(img), hm=np.array(hm), V=V, J=J, hm_seed=73,
                        param_seed=71, param_scale=0.05, hm_sum=np.array([float(h.double().sum()) for h in hms]),
                        sub_idx=sub, root_sub=rc.reshape(2, N)[:, sub], root_sum=rc.astype(np.float64).sum(axis=(1, 2, 3)),
                        root_abs_sum=np.abs(rc.astype(np.float64)).sum(axis=(1, 2, 3)), root_max=rc.max(axis=(1, 2, 3)),
                        nms_vals=vals.numpy(), nms_idx=idx.numpy(), grid_centers=grid_centers.numpy())
    print("rootnet_full: root range", float(rc.min()), float(rc.max()), "top vals", vals.numpy()[:, :5])


def g_softargmax():
    from models.pose_regression_net import SoftArgmaxLayer
    rng = np.random.default_rng(81)
    x = torch.from_numpy((rng.random((3, 4, 8, 8, 8), dtype=np.float32) * 0.08))
    x[0, 1, 3, 4, 5] = 0.6            # one sharp peak, one flat channel, the rest noisy
    x[1, 2] = 0.01
    lin = [np.linspace(-1000, 1000, 8) + c for c in (100.0, -300.0, 900.0)]
    g = np.stack(np.meshgrid(*lin, indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)
    grids = torch.from_numpy(np.concatenate([g, g + 50.0, g - 777.0], 0))
    layer = SoftArgmaxLayer(AD(NETWORK=AD(BETA=100.0)))
    out = layer(x, grids)
    np.savez_compressed(os.path.join(HERE, "softargmax.npz"), x=x.numpy(), grids=grids.numpy(), beta=100.0, out=out.numpy())
    print("softargmax:", out.shape, out[0, 1].numpy())


def g_pose_resnet():
    import models.pose_resnet as pr
    rec = {}
    x = torch.from_numpy(np.random.default_rng(91).standard_normal((1, 3, 64, 64)).astype(np.float32))
    rec["x_seed"] = 91
    for name, layers in (("r50", 50), ("attn18", 18)):
        cfg = full_cfg((960, 512), (240, 128), (80, 80, 20), (64, 64, 64), 15, num_layers=layers)
        if name == "r50":
            net = pr.get_pose_net(cfg, is_train=False)
        else:
            cfg.ATTN_NUM_LAYERS = layers
            net = pr.get_pose_attn_net(cfg, is_train=False)
        # He-style deterministic weights so that activations survive 50 layers (N(0, 0.05) would vanish)
        rng = np.random.default_rng(93)
        sd = net.state_dict()
        with torch.no_grad():
            for k in sorted(sd):
                t = sd[k]
                if not torch.is_floating_point(t):
                    continue
                if t.dim() == 4:
                    fan = t.shape[1] * t.shape[2] * t.shape[3]
                    a = rng.standard_normal(tuple(t.shape)).astype(np.float32) * np.sqrt(2.0 / fan)
                elif k.endswith("running_var"):
                    a = 1.0 + 0.1 * np.abs(rng.standard_normal(tuple(t.shape))).astype(np.float32)
                elif k.endswith("weight"):
                    a = 1.0 + 0.05 * rng.standard_normal(tuple(t.shape)).astype(np.float32)
                else:
                    a = 0.05 * rng.standard_normal(tuple(t.shape)).astype(np.float32)
                t.copy_(torch.from_numpy(a))
        net.eval()
        with torch.no_grad():
            y = net(x)
        rec[name + "_out"] = y.numpy()
        rec[name + "_keys"] = np.array(sorted(sd.keys()))
        print("pose_resnet", name, y.shape, float(y.abs().mean()), len(sd))
    rec["w_seed"] = 93
    np.savez_compressed(os.path.join(HERE, "pose_resnet.npz"), **rec)


def _keys(model):
    return {k: list(v.shape) for k, v in model.state_dict().items()}


def g_state_dict_keys():
    import models.multi_person_posenet as mp
    import models.multi_person_posenet_ssv as mps
    cfg = full_cfg((960, 512), (240, 128), (80, 80, 20), (64, 64, 64), 15)
    out = {"multi_person_posenet": _keys(mp.get_multi_person_pose_net(cfg, is_train=False))}
    cfg_s = full_cfg((960, 512), (240, 128), (80, 80, 20), (64, 64, 64), 15, ROOTNET_ROOTHM=True, ROOTNET_TRAIN_SYNTH=True)
    cfg_s.WITH_ATTN = True
    out["multi_person_posenet_ssv_attn"] = _keys(mps.get_multi_person_pose_net(cfg_s, is_train=False))
    cfg_s.WITH_ATTN = False
    out["multi_person_posenet_ssv"] = _keys(mps.get_multi_person_pose_net(cfg_s, is_train=False))
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("state_dict_keys:", {k: len(v) for k, v in out.items()})


def g_ssv_inference():
    import models.multi_person_posenet_ssv as mps
    img, hm, V, J, B = (384, 288), (96, 72), 4, 4, 2
    cfg = full_cfg(img, hm, (24, 24, 8), (16, 16, 16), J, ROOTNET_ROOTHM=True, ROOTNET_TRAIN_SYNTH=True)
    cfg.MULTI_PERSON.THRESHOLD = 0.0
    cfg.BACKBONE_MODEL = ""                       # heat-maps are inputs (do_inference(input_heatmaps=...), :112-113)
    model = mps.get_multi_person_pose_net(cfg, is_train=False)
    syn.fill_parameters_deterministic(model, seed=95, scale=0.05)
    model.eval()
    meta = syn.make_meta(B, V, img, ssv_style=True)
    hms, _ = syn.people_heatmaps(B, V, J, hm[1], hm[0], img, seed=97)
    with torch.no_grad():
        pred, _, grid_centers = model(views1=None, meta1=meta, input_heatmaps1=hms, inference=True)
    np.savez_compressed(os.path.join(HERE, "ssv_inference.npz"), img=np.array(img), hm=np.array(hm), V=V, J=J, B=B,
                        cube=np.array([24, 24, 8]), fine_cube=np.array([16, 16, 16]), hm_seed=97, param_seed=95,
                        param_scale=0.05, threshold=0.0, pred=pred.numpy(), grid_centers=grid_centers.numpy(),
                        keys=np.array(sorted(model.state_dict().keys())))
    print("ssv_inference: valid", int((grid_centers[:, :, 3] >= 0).sum()), "pred[0,0,:2]", pred[0, 0, :2].numpy())


ALL = {"rootnet_full": g_rootnet_full, "softargmax": g_softargmax, "pose_resnet": g_pose_resnet,
       "state_dict_keys": g_state_dict_keys, "ssv_inference": g_ssv_inference}

if __name__ == "__main__":
    install_shims()
    torch.set_num_threads(8)
    for n in (sys.argv[1:] or list(ALL)):
        ALL[n]()
