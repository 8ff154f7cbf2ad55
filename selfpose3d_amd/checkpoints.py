"""Checkpoint files and the hand-off between the three training stages.

The reference trains in stages (backbone -> root net -> pose net), each stage starting from files the previous
one wrote; the YAML names them (`NETWORK.PRETRAINED_BACKBONE`, `NETWORK.INIT_ROOTNET`, `NETWORK.INIT_ALL`) and
/root/reference/tools/train_3d.py:150-180 loads them before the optional `TRAIN.RESUME`.  This module is that
block for a bare (not DataParallel-wrapped) model, one process per GPU: every rank reads the file itself
(`map_location="cpu"`, a few hundred MB at most), so no broadcast is needed and ranks cannot disagree.

A named file that does not exist is an error (the reference's `torch.load` raises too): a stage that silently
trains against random weights of the previous stage is the failure this module exists to prevent.
File names / dict keys of checkpoints: lib/utils/utils.py:84-115.
"""
from __future__ import annotations

import logging
import os

import torch
from torch import nn

logger = logging.getLogger(__name__)
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(path: str) -> dict:
    if not os.path.isfile(path):
        raise FileNotFoundError(f"checkpoint named by the config does not exist: {path}")
    sd = torch.load(path, map_location="cpu")
    if not isinstance(sd, dict):
        raise TypeError(f"{path}: expected a state_dict, got {type(sd).__name__}")
    return sd


def _sub_state(sd: dict, marker: str) -> dict:
    """entries of a whole-model state_dict that belong to one sub-module: keys that CONTAIN the marker, with
    "<marker>." removed (tools/train_3d.py:153-157,164-168 - substring test and str.replace, kept as they are)"""
    return {k.replace(marker + ".", ""): v for k, v in sd.items() if marker in k}


def load_backbone_panoptic(model: nn.Module, pretrained_file: str) -> nn.Module:
    """2D-backbone checkpoint of the supervised pipeline into ``model.backbone`` (lib/utils/utils.py:118-149):
    the path is taken relative to the repository root, a DataParallel "module." prefix is dropped, tensors are kept
    when name and shape agree, and a final layer with a different joint count is Xavier-initialised (weights) /
    zeroed (bias) with the first min(J_file, J_model) filters copied over."""
    path = os.path.abspath(os.path.join(REPO_ROOT, pretrained_file))
    own = model.backbone.state_dict()
    picked = {}
    for k, v in _read(path).items():
        name = k.replace("module.", "")
        if name in own and v.shape == own[name].shape:
            picked[name] = v
        elif name in ("final_layer.weight", "final_layer.bias"):
            fresh = torch.zeros_like(own[name])
            if name.endswith("weight"):
                nn.init.xavier_uniform_(fresh)
            n = min(fresh.shape[0], v.shape[0])
            fresh[:n] = v[:n]
            picked[name] = fresh
            logger.info(f"=> final layer {name}: {n} of {fresh.shape[0]} filters from the file, the rest re-initialised")
    logger.info(f"=> load backbone state_dict from {path}")
    model.backbone.load_state_dict(picked)                               # strict, as utils.py:147
    return model


def init_from_config(model: nn.Module, cfg) -> list[str]:
    """the stage hand-off of tools/train_3d.py:150-180, in its order; returns what was loaded (for the log / tests)"""
    done = []
    net = cfg.NETWORK
    if net.PRETRAINED_BACKBONE:
        if net.PRETRAINED_BACKBONE_PSEUDOGT:                             # a whole-model file of the backbone stage
            logger.info(f"=> loading backbone from = {net.PRETRAINED_BACKBONE}")
            model.backbone.load_state_dict(_sub_state(_read(net.PRETRAINED_BACKBONE), "backbone"), strict=True)
        else:
            load_backbone_panoptic(model, net.PRETRAINED_BACKBONE)
        done.append("PRETRAINED_BACKBONE")
    if net.INIT_ROOTNET:
        if getattr(model, "root_net", None) is None:
            raise ValueError("NETWORK.INIT_ROOTNET is set but the model has no root net (TRAIN_ONLY_2D?)")
        logger.info(f"=> loading rootnet from = {net.INIT_ROOTNET}")
        model.root_net.load_state_dict(_sub_state(_read(net.INIT_ROOTNET), "root_net"), strict=True)
        done.append("INIT_ROOTNET")
    if net.INIT_ALL:
        logger.info(f"=> loading all from = {net.INIT_ALL}")
        model.load_state_dict(_read(net.INIT_ALL), strict=True)
        done.append("INIT_ALL")
    return done


def save_checkpoint(state, is_best, out_dir, filename="checkpoint.pth.tar"):
    """file names / dict keys of the reference (lib/utils/utils.py:109-115)"""
    torch.save(state, os.path.join(out_dir, filename))
    torch.save(state["state_dict"], os.path.join(out_dir, f"model_epoch_{state['epoch']}.pth.tar"))
    if is_best and "state_dict" in state:
        torch.save(state["state_dict"], os.path.join(out_dir, "model_best.pth.tar"))


def load_checkpoint(model, optimizer, out_dir, filename="checkpoint.pth.tar"):
    """TRAIN.RESUME (lib/utils/utils.py:84-106): (start_epoch, best_precision, last_epoch); no file = fresh start"""
    f = os.path.join(out_dir, filename)
    if not os.path.isfile(f):
        logger.info(f"=> no checkpoint found at {f}")
        return 0, 0.0, -1
    ck = torch.load(f, map_location="cpu")
    model.load_state_dict(ck["state_dict"])
    optimizer.load_state_dict(ck["optimizer"])
    return ck["epoch"], ck.get("precision", 0.0), ck["epoch"] - 1
