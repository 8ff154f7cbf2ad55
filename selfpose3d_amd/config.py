"""Experiment configuration: attribute-dict defaults + YAML overlay.

Reads the reference's YAML files unchanged (e.g.
/root/reference/configs/panoptic/resnet50/prn64_cpn80x80x20_960x512_cam5.yaml); semantics of
``update_config`` follow /root/reference/lib/core/config.py:233-274 (nested overlay, unknown
top-level sections / keys rejected).  Only the keys the hot path and its callers read get
defaults here; any other key present in a YAML is accepted into its (known) section.
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
    """Panoptic 5-camera defaults (prn64_cpn80x80x20_960x512_cam5.yaml + core/config.py)."""
    return _wrap({
        "OUTPUT_DIR": "output", "LOG_DIR": "log", "DATA_DIR": "", "BACKBONE_MODEL": "pose_resnet",
        "MODEL": "multi_person_posenet", "GPUS": "0", "WORKERS": 4, "PRINT_FREQ": 100,
        "WITH_SSV": False, "WITH_ATTN": False,
        "CUDNN": {"BENCHMARK": True, "DETERMINISTIC": False, "ENABLED": True},
        "NETWORK": {
            "PRETRAINED": "", "PRETRAINED_BACKBONE": "", "TRAIN_BACKBONE": False, "TRAIN_ONLY_2D": False,
            "NUM_JOINTS": 15, "IMAGE_SIZE": [960, 512], "HEATMAP_SIZE": [240, 128], "SIGMA": 3,
            "TARGET_TYPE": "gaussian", "USE_GT": False, "BETA": 100.0, "ROOTNET_ROOTHM": False,
            "TRAIN_ONLY_ROOTNET": False, "FREEZE_ROOTNET": False,
        },
        "POSE_RESNET": {"NUM_LAYERS": 50, "DECONV_WITH_BIAS": False, "NUM_DECONV_LAYERS": 3,
                        "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_DECONV_KERNELS": [4, 4, 4],
                        "FINAL_CONV_KERNEL": 1},
        "LOSS": {"USE_TARGET_WEIGHT": True},
        "DATASET": {"ROOT": "", "TRAIN_DATASET": "panoptic", "TEST_DATASET": "panoptic", "ROOTIDX": 2,
                    "ROOTIDX_PSEUDO": 2, "CAMERA_NUM": 5, "DATA_FORMAT": "jpg"},
        "TRAIN": {"BATCH_SIZE": 2, "SHUFFLE": True, "BEGIN_EPOCH": 0, "END_EPOCH": 10, "RESUME": False,
                  "OPTIMIZER": "adam", "LR": 1e-4, "LR_FACTOR": 0.1, "LR_STEP": [90, 110], "WD": 1e-4,
                  "MOMENTUM": 0.9, "NESTEROV": False, "GAMMA1": 0.99, "GAMMA2": 0.0},
        "TEST": {"BATCH_SIZE": 4, "MODEL_FILE": "model_best.pth.tar", "STATE": ""},
        "DEBUG": {"DEBUG": False},
        "MULTI_PERSON": {"SPACE_SIZE": [8000.0, 8000.0, 2000.0], "SPACE_CENTER": [0.0, -500.0, 800.0],
                         "INITIAL_CUBE_SIZE": [80, 80, 20], "MAX_PEOPLE_NUM": 10, "THRESHOLD": 0.3},
        "PICT_STRUCT": {"GRID_SIZE": [2000.0, 2000.0, 2000.0], "CUBE_SIZE": [64, 64, 64]},
    })


def _overlay(dst: AttrDict, src: dict, path: str):
    for k, v in src.items():
        if isinstance(v, dict):
            if k not in dst:
                raise ValueError(f"{path}{k} not exist in config")     # core/config.py:273-274
            if not isinstance(dst[k], dict):
                raise ValueError(f"{path}{k} is not a section")
            _overlay(dst[k], v, path + k + ".")
        else:
            dst[k] = v


def update_config(cfg: AttrDict, config_file: str) -> AttrDict:
    with open(config_file) as f:
        exp = yaml.safe_load(f) or {}
    _overlay(cfg, exp, "")
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
