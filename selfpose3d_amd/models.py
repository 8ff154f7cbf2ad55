"""Model registry: ``cfg.MODEL`` -> top-level network, the dispatch the reference's entry points do with
``eval('models.' + config.MODEL + '.get_multi_person_pose_net')`` (/root/reference/tools/train_3d.py:121,
tools/validate_3d.py:82).  Unknown names raise - a config is never silently run as a different model."""
from __future__ import annotations

MODELS = ("multi_person_posenet", "multi_person_posenet_ssv")


def get_multi_person_pose_net(cfg, is_train: bool = True):
    name = str(cfg.MODEL)
    if name == "multi_person_posenet":
        from .multi_person_posenet import get_multi_person_pose_net as f
    elif name == "multi_person_posenet_ssv":
        from .multi_person_posenet_ssv import get_multi_person_pose_net as f
    else:
        raise ValueError(f"MODEL: {name!r} is not built here (known: {', '.join(MODELS)})")
    return f(cfg, is_train=is_train)


def is_ssv(cfg) -> bool:
    return str(cfg.MODEL) == "multi_person_posenet_ssv"
