"""Experiment configuration: attribute-dict defaults + YAML overlay.

Reads the reference's YAML files unchanged (e.g.
/root/reference/configs/panoptic/resnet50/prn64_cpn80x80x20_960x512_cam5.yaml); semantics of
``update_config`` follow /root/reference/lib/core/config.py:233-274 (nested overlay, unknown
top-level names AND unknown keys inside a section rejected, as `_update_dict` does at :253-257).
"""
from __future__ import annotations

import copy

import yaml


class AttrDict(dict):
    """dict with attribute access (the reference uses easydict, absent from this image)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(d):
    if isinstance(d, dict):
        return AttrDict({k: _wrap(v) for k, v in d.items()})
    return d


def default_config() -> AttrDict:
    """Panoptic 5-camera defaults (prn64_cpn80x80x20_960x512_cam5.yaml) for the keys the hot path and its callers
    read; every other key of the reference's schema (lib/core/config.py:17-231) is present with the reference's default
    so that the overlay can reject a key the reference would reject.  ``SEED`` is this build's only extension key
    (per-rank RNG streams, tools/train_3d.py)."""
    return _wrap({
        "OUTPUT_DIR": "output", "LOG_DIR": "log", "DATA_DIR": "", "BACKBONE_MODEL": "pose_resnet",
        "MODEL": "multi_person_posenet", "GPUS": "0", "WORKERS": 4, "PRINT_FREQ": 100,
        "WITH_SSV": False, "WITH_ATTN": False, "ATTN_WEIGHT": 0.1, "ATTN_NUM_LAYERS": 18,
        "USE_L1": False, "L1_WEIGHT": 0.1, "L1_ATTN": False, "MIN_VIEWS_CHECK": 1, "EVAL_ROOTNET_ONLY": False,
        "COCO_TO_PANOPTIC_MAPPING": [5, 0, 11, 5, 7, 9, 11, 13, 15, 6, 8, 10, 12, 14, 16],
        "SEED": 0,
        # HigherHRNet description: read by no module of this build (BACKBONE_MODEL is pose_resnet), kept as schema
        "MODEL_EXTRA": {"PRETRAINED_LAYERS": ["*"], "FINAL_CONV_KERNEL": 1, "STEM_INPLANES": 64,
                        "STAGE2": {}, "STAGE3": {}, "STAGE4": {}, "DECONV": {}},
        "CUDNN": {"BENCHMARK": True, "DETERMINISTIC": False, "ENABLED": True},
        "NETWORK": {
            "PRETRAINED": "", "PRETRAINED_BACKBONE": "", "PRETRAINED_BACKBONE_PSEUDOGT": False,
            "TRAIN_BACKBONE": False, "TRAIN_ONLY_2D": False,
            "NUM_JOINTS": 15, "INPUT_SIZE": 512, "IMAGE_SIZE": [960, 512], "HEATMAP_SIZE": [240, 128],
            "IMAGE_SIZE_ORIG": [1920, 1080], "SIGMA": 3, "TARGET_TYPE": "gaussian", "AGGRE": True,
            "USE_GT": False, "BETA": 100.0, "ROOTNET_ROOTHM": False, "ROOTNET_TRAIN_SYNTH": False,
            "INIT_TRAIN_EPOCHS_ROOTNET": 0, "INIT_ROOTNET": "", "TRAIN_ONLY_ROOTNET": False,
            "ROOTNET_BUFFER_SIZE": 5000, "FREEZE_ROOTNET": False, "INIT_ALL": "",
            "SINGLE_AUG_TRAINING_POSENET": False, "ROOT_CONSISTENCY_LOSS": True,
            "WEIGHT_ROOT_SYN": 100.0, "WEIGHT_ROOT_REG": 1.0,
            "ROOTNET_SYN_RANGE": [[2500.0, -2000.0], [1500.0, -1500.0], [250.0, -300.0]],
        },
        "POSE_RESNET": {"NUM_LAYERS": 50, "DECONV_WITH_BIAS": False, "NUM_DECONV_LAYERS": 3,
                        "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_DECONV_KERNELS": [4, 4, 4],
                        "FINAL_CONV_KERNEL": 1},
        "LOSS": {"USE_TARGET_WEIGHT": True, "USE_DIFFERENT_JOINTS_WEIGHT": False},
        "DATASET": {"ROOT": "", "TRAIN_DATASET": "panoptic", "TEST_DATASET": "panoptic",
                    "TRAIN_SUBSET": "train", "TEST_SUBSET": "validation", "ROOTIDX": 2,
                    "ROOTIDX_PSEUDO": 2, "CAMERA_NUM": 5, "CAMERAS": [0, 1, 2, 3, 4], "CAMERA_NUM_TOTAL": 5,
                    "DATA_FORMAT": "jpg", "BBOX": 2000, "CROP": True, "COLOR_RGB": False, "FLIP": True,
                    "DATA_AUGMENTATION": True, "SCALE_FACTOR": 0, "SCALE_FACTOR1": 0, "SCALE_FACTOR2": 0,
                    "ROT_FACTOR": 0, "ROT_FACTOR1": 0, "ROT_FACTOR2": 0, "APPLY_CUTOUT": False,
                    "APPLY_RANDAUG": False, "SUFFIX": "sub", "GT_3D_FILE": "panoptic_training_pose.pkl",
                    "TRAIN_PSEUDO_GT3D": False},
        "TRAIN": {"BATCH_SIZE": 2, "SHUFFLE": True, "BEGIN_EPOCH": 0, "END_EPOCH": 10, "RESUME": False,
                  "OPTIMIZER": "adam", "LR": 1e-4, "LR_FACTOR": 0.1, "LR_STEP": [90, 110], "L1_EPOCH": 5, "WD": 1e-4,
                  "MOMENTUM": 0.9, "NESTEROV": False, "GAMMA1": 0.99, "GAMMA2": 0.0},
        "TEST": {"BATCH_SIZE": 4, "MODEL_FILE": "model_best.pth.tar", "STATE": "", "FLIP_TEST": False,
                 "POST_PROCESS": False, "SHIFT_HEATMAP": False, "USE_GT_BBOX": False, "IMAGE_THRE": 0.1,
                 "NMS_THRE": 0.6, "OKS_THRE": 0.5, "IN_VIS_THRE": 0.0, "BBOX_FILE": "", "BBOX_THRE": 1.0,
                 "MATCH_IOU_THRE": 0.3, "DETECTOR": "fpn_dcn", "DETECTOR_DIR": "",
                 "HEATMAP_LOCATION_FILE": "predicted_heatmaps.h5"},
        "DEBUG": {"DEBUG": False, "SAVE_BATCH_IMAGES_GT": False, "SAVE_BATCH_IMAGES_PRED": False,
                  "SAVE_HEATMAPS_GT": False, "SAVE_HEATMAPS_PRED": False, "SAVE_3D_POSES": False,
                  "SAVE_3D_ROOTS": False},
        "MULTI_PERSON": {"SPACE_SIZE": [8000.0, 8000.0, 2000.0], "SPACE_CENTER": [0.0, -500.0, 800.0],
                         "ESTIMATED_SPACE_CENTER": [0.0, -500.0, 800.0],
                         "INITIAL_CUBE_SIZE": [80, 80, 20], "MAX_PEOPLE_NUM": 10, "THRESHOLD": 0.3},
        "PICT_STRUCT": {"FIRST_NBINS": 16, "PAIRWISE_FILE": "", "RECUR_NBINS": 2, "RECUR_DEPTH": 10,
                        "LIMB_LENGTH_TOLERANCE": 150, "GRID_SIZE": [2000.0, 2000.0, 2000.0],
                        "CUBE_SIZE": [64, 64, 64], "DEBUG": False, "TEST_PAIRWISE": False, "SHOW_ORIIMG": False,
                        "SHOW_CROPIMG": False, "SHOW_HEATIMG": False},
    })


def _overlay(dst: AttrDict, src: dict):
    """the reference's two-level rule (lib/core/config.py:233-274): a top-level name must exist; inside a section a
    leaf name must exist (``ValueError("SECTION.KEY not exist in config.py")``, :253-257); what hangs below a section's
    key is taken whole.  ``HEATMAP_SIZE`` / ``IMAGE_SIZE`` given as one int mean a square (:243-252)."""
    for k, v in src.items():
        if k not in dst:
            raise ValueError(f"{k} not exist in config.py")               # :273-274
        if isinstance(v, dict) and isinstance(dst[k], dict):
            for vk, vv in v.items():
                if vk not in dst[k]:
                    raise ValueError(f"{k}.{vk} not exist in config.py")  # :256-257
                if k == "NETWORK" and vk in ("HEATMAP_SIZE", "IMAGE_SIZE") and isinstance(vv, int):
                    vv = [vv, vv]
                dst[k][vk] = _wrap(vv)
        elif isinstance(v, dict) != isinstance(dst[k], dict):
            raise ValueError(f"{k}: a section and a value cannot replace each other")
        else:
            dst[k] = v


def update_config(cfg: AttrDict, config_file: str) -> AttrDict:
    with open(config_file) as f:
        exp = yaml.safe_load(f) or {}
    _overlay(cfg, exp)
    return cfg


def load_config(config_file: str | None = None, **overrides) -> AttrDict:
    cfg = default_config()
    if config_file:
        update_config(cfg, config_file)
    for dotted, v in overrides.items():
        node = cfg
        parts = dotted.split("__")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg
