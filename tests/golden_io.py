"""Rebuild the exact inputs a golden .npz was generated on (tests/golden/make_goldens.py)."""
import os

import numpy as np
import torch

from selfpose3d_amd import synthetic as syn
from selfpose3d_amd.camera_pack import pack_cameras

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


class Case:
    """inputs of one unprojection golden case, in reference (ProjectLayer.forward) form and packed form"""

    def __init__(self, name):
        g = load(name)
        self.g = g
        self.name = name
        self.B, self.V, self.J = int(g["B"]), int(g["V"]), int(g["J"])
        self.img = [int(v) for v in g["img"]]
        self.hm = [int(v) for v in g["hm"]]
        self.cube = [int(v) for v in g["cube"]]
        self.grid_size = [float(v) for v in g["grid_size"]]
        rot = g["rotations"] if len(g["rotations"]) else None
        sm = g["scale_mults"] if len(g["scale_mults"]) else None
        self.flip = torch.tensor(g["flip"]) if len(g["flip"]) else None
        self.meta = syn.make_meta(self.B, self.V, self.img, rotations=rot, scale_mults=sm,
                                  ssv_style=bool(g["ssv_style"]))
        w, h = self.hm
        if str(g["hm_kind"]) == "random":
            self.hms = syn.random_heatmaps(self.B, self.V, self.J, h, w, seed=int(g["seed"]))
        elif str(g["hm_kind"]) == "random_wide":                  # round 6: values outside [0, 1], the clamp blocks gradient
            self.hms = [x * 2.4 + -0.7 for x in syn.random_heatmaps(self.B, self.V, self.J, h, w, seed=int(g["seed"]))]
        else:
            self.hms, _ = syn.people_heatmaps(self.B, self.V, self.J, h, w, self.img, seed=int(g["seed"]))
        # inputs must be the ones the golden was computed on
        sums = np.array([float(x.double().sum()) for x in self.hms])
        assert np.allclose(sums, g["hm_sum"], rtol=0, atol=1e-6 * max(1.0, float(np.abs(sums).max()))), name
        gc = g["grid_center"]
        if bool(g["center_is_list"]):
            self.grid_center = [[float(v) for v in gc.reshape(-1)[:3]]]
            self.centers = np.repeat(np.asarray(gc, np.float32).reshape(1, 3), self.B, 0)
            self.valid = np.ones(self.B, np.uint8)
        else:
            self.grid_center = torch.from_numpy(np.asarray(gc, np.float32))
            self.centers = np.asarray(gc[:, :3], np.float32)
            self.valid = (gc[:, 3] >= 0).astype(np.uint8)
        self.cam = pack_cameras(self.meta, self.B, self.img, self.flip)
        self.N = self.cube[0] * self.cube[1] * self.cube[2]

    def expected(self):
        """-> (cubes (B,J,n_sel), grids (B,n_sel,3), flat voxel indices or None)"""
        g = self.g
        if "cubes" in g:
            return g["cubes"].reshape(self.B, self.J, self.N), g["grids"], None
        return g["cubes_sub"], g["grids_sub"], g["sub_idx"]


SMALL_CASES = ["unproj_coarse_small", "unproj_coarse_j1_v1", "unproj_coarse_aug", "unproj_fine_small",
               "unproj_grad_small", "unproj_grad_fine_aug"]
FULL_CASES = ["unproj_coarse_full_96x72", "unproj_coarse_full_240x128", "unproj_fine_full_240x128",
              "unproj_stress_v10", "unproj_people_coarse", "unproj_coarse_b4"]
GRAD_FULL_CASES = ["unproj_grad_root_full", "unproj_grad_fine_full"]      # round 6: the backward at the sizes it runs at


# ---- round 3: model-level training goldens (tests/golden/make_goldens_r3.py) -----------------------------------------
TRAIN_SMALL = dict(img=(128, 96), hm=(32, 24), V=3, J=15, cube=(24, 24, 8), fine_cube=(16, 16, 16), max_people=4,
                   layers=18, threshold=0.0, sigma=3)


def train_cfg(ssv=False, **net):
    """this repo's config for the small training scene (the golden script builds the reference's cfg from TRAIN_SMALL)"""
    from selfpose3d_amd.config import load_config
    t = TRAIN_SMALL
    kw = dict(NETWORK__IMAGE_SIZE=list(t["img"]), NETWORK__HEATMAP_SIZE=list(t["hm"]), NETWORK__NUM_JOINTS=t["J"],
              NETWORK__SIGMA=t["sigma"], NETWORK__TRAIN_BACKBONE=True, DATASET__CAMERA_NUM=t["V"],
              POSE_RESNET__NUM_LAYERS=t["layers"], MULTI_PERSON__INITIAL_CUBE_SIZE=list(t["cube"]),
              MULTI_PERSON__MAX_PEOPLE_NUM=t["max_people"], MULTI_PERSON__THRESHOLD=t["threshold"],
              PICT_STRUCT__CUBE_SIZE=list(t["fine_cube"]), TRAIN__BATCH_SIZE=2)
    if ssv:
        kw.update(MODEL="multi_person_posenet_ssv", WITH_SSV=True, WITH_ATTN=True, ATTN_WEIGHT=0.1, ATTN_NUM_LAYERS=18,
                  USE_L1=True, L1_WEIGHT=0.01, L1_ATTN=True, TRAIN__L1_EPOCH=0, NETWORK__ROOTNET_ROOTHM=True,
                  NETWORK__ROOTNET_TRAIN_SYNTH=True, NETWORK__FREEZE_ROOTNET=True, DATASET__ROT_FACTOR1=30,
                  DATASET__ROT_FACTOR2=30, DATASET__SCALE_FACTOR1=0.25, DATASET__SCALE_FACTOR2=0.25)
    for k, v in net.items():
        kw["NETWORK__" + k] = v
    return load_config(None, **kw)


def train_batch(cfg, B=2, seed=5, ssv=False):
    """a collated batch of the synthetic (SSV) dataset - the exact inputs of the training goldens"""
    from torch.utils.data import default_collate
    from selfpose3d_amd.synthetic_dataset import SyntheticPanoptic, SyntheticPanopticSSV
    ds = (SyntheticPanopticSSV if ssv else SyntheticPanoptic)(cfg, num_frames=B, seed=seed, max_people=3)
    return default_collate([ds[i] for i in range(B)])


def he_fill(model, seed):
    """deterministic, depth-safe parameter fill keyed by sorted state_dict names (same keys in the reference and here):
    He-scaled conv / linear weights, BatchNorm gamma ~ 1, running_var >= 1, small biases / means"""
    rng = np.random.default_rng(seed)
    sd = model.state_dict()
    with torch.no_grad():
        for k in sorted(sd):
            t = sd[k]
            if not torch.is_floating_point(t):
                continue
            if t.dim() >= 4:
                fan = int(np.prod(t.shape[1:]))
                a = rng.standard_normal(tuple(t.shape)).astype(np.float32) * np.sqrt(2.0 / fan)
            elif k.endswith("running_var"):
                a = 1.0 + 0.1 * np.abs(rng.standard_normal(tuple(t.shape))).astype(np.float32)
            elif k.endswith("weight"):
                a = 1.0 + 0.05 * rng.standard_normal(tuple(t.shape)).astype(np.float32)
            else:
                a = 0.05 * rng.standard_normal(tuple(t.shape)).astype(np.float32)
            t.copy_(torch.from_numpy(a))
    return model


# ---- round 6: the SSV rendering caught inside the reference's forward (tests/golden/make_goldens_r6.py) -----------------
RENDER_FULL = dict(img=(960, 512), hm=(240, 128), V=3, J=15, cube=(24, 24, 8), fine_cube=(16, 16, 16), max_people=4,
                   layers=18, threshold=0.0, sigma=3)


def render_cfg():
    """the small SSV scene at FULL heat-map size (240x128): what render_ssv_full.npz was generated on"""
    global TRAIN_SMALL
    keep = TRAIN_SMALL
    TRAIN_SMALL = RENDER_FULL
    try:
        return train_cfg(ssv=True)
    finally:
        TRAIN_SMALL = keep


# ---- round 6: one supervised train step at FULL size (tests/golden/make_goldens_r6.py g_train_step_full) --------------------
TRAIN_FULL = dict(img=(960, 512), hm=(240, 128), V=5, J=15, cube=(80, 80, 20), fine_cube=(64, 64, 64), max_people=10,
                  layers=50, threshold=0.3, sigma=3)


def train_full_cfg(**net):
    """BASELINE configs[2]'s sizes (ResNet-50, 5 x 960x512, 80x80x20 root grid, 64^3 pose cubes, batch 2)"""
    global TRAIN_SMALL
    keep = TRAIN_SMALL
    TRAIN_SMALL = TRAIN_FULL
    try:
        return train_cfg(**net)
    finally:
        TRAIN_SMALL = keep


# ---- round 4: the pose stage at full size (tests/golden/make_goldens_r4.py) --------------------------------------------
POSENET_FULL = dict(img=(960, 512), hm=(240, 128), V=5, J=15, B=2, fine_cube=(64, 64, 64), hm_seed=411, pose_seed=413,
                    param_scale=0.05, stride=211)


def posenet_full_inputs(device="cpu"):
    """heat-maps, meta and the (B, K=2, 5) proposal table of the posenet_full case (generator, tests and bench.py's
    pose_stage check all build them here): the synthetic people scene, proposal centres near (not on) two people's roots,
    sample 1 of slot 1 invalid (flag < 0)"""
    from selfpose3d_amd import synthetic as syn
    c = POSENET_FULL
    meta = syn.make_meta(c["B"], c["V"], c["img"])
    hms, pts = syn.people_heatmaps(c["B"], c["V"], c["J"], c["hm"][1], c["hm"][0], c["img"], seed=c["hm_seed"], device=device)
    gc = np.zeros((c["B"], 2, 5), np.float32)
    off = np.array([[37.0, -21.0, 55.0], [-44.0, 62.0, -18.0]], np.float32)
    for b in range(c["B"]):
        for k in range(2):
            root = pts[b][k % pts[b].shape[0], 2]                 # the person's root joint (ROOTIDX = 2)
            gc[b, k, :3] = root.astype(np.float32) + off[k]
            gc[b, k, 3] = 0.0
            gc[b, k, 4] = 0.9 - 0.1 * k
    gc[1, 1, 3] = -1.0                                            # the invalid proposal: skipped by ProjectLayer and the V2V
    return hms, meta, torch.from_numpy(gc).to(device)
