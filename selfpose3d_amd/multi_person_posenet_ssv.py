"""Self-supervised top-level model (``MODEL: multi_person_posenet_ssv``), INFERENCE path.

Interface, return tuple and state_dict prefixes (``backbone.``, ``attn.``, ``pose_net.``, ``root_net.``)
follow /root/reference/lib/models/multi_person_posenet_ssv.py:29-153: ``do_inference`` = backbone per view
-> ``CuboidProposalNetSoft`` (HIP unprojection, V2V, fused NMS/top-k) -> pose net on every valid proposal.
As in ``MultiPersonPoseNet`` the views run through the backbone as one batch and all proposals are
unprojected in ONE launch and regressed in batched V2V calls (the reference loops over MAX_PEOPLE_NUM
candidates, :142-148); in eval mode the results are the loop's.

The SSV *training* forward (two augmented passes, Hungarian matching, differentiable re-rendering,
:197-501) is NOT built: calling ``forward`` without ``inference=True`` raises instead of silently
training something else.  Its kernels exist separately (``reprojection.py``, ``sp3d_render_joints_*``,
``CuboidProposalNetSoft.train_rootnet``).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import pose_resnet
from .cuboid_proposal_net_soft import CuboidProposalNetSoft
from .pose_regression_net import PoseRegressionNet


class MultiPersonPoseNetSSV(nn.Module):
    def __init__(self, backbone, cfg, attn=None):
        super().__init__()
        self.num_cand = int(cfg.MULTI_PERSON.MAX_PEOPLE_NUM)
        self.num_joints = int(cfg.NETWORK.NUM_JOINTS)
        self.backbone = backbone
        self.WITH_ATTN = bool(cfg.get("WITH_ATTN", False))
        if self.WITH_ATTN:
            self.attn = attn
        self.use_root_gt = bool(cfg.NETWORK.USE_GT)
        self.train_only_2d = bool(cfg.NETWORK.TRAIN_ONLY_2D)
        self.train_only_rootnet = bool(cfg.NETWORK.get("TRAIN_ONLY_ROOTNET", False))
        self.eval_rootnet_only = bool(cfg.get("EVAL_ROOTNET_ONLY", False))
        self.root_id = cfg.DATASET.ROOTIDX
        if self.train_only_2d:                                  # multi_person_posenet_ssv.py:63-69
            self.use_root_gt = True
        elif not self.train_only_rootnet:
            self.pose_net = PoseRegressionNet(cfg)
        if not self.use_root_gt:                                # :71
            self.root_net = CuboidProposalNetSoft(cfg)

    def use_channels_last(self, on: bool = True):
        for name in ("root_net", "pose_net"):
            if hasattr(self, name):
                getattr(self, name).use_channels_last(on)
        for net in (self.backbone, getattr(self, "attn", None)):
            if net is not None:
                net.to(memory_format=torch.channels_last if on else torch.contiguous_format)
        return self

    def _heatmaps(self, views, input_heatmaps):
        if views is None:
            return input_heatmaps
        if hasattr(self.backbone, "forward_views"):
            return self.backbone.forward_views(views)
        return [self.backbone(v) for v in views]

    @torch.no_grad()
    def do_inference(self, views=None, meta=None, input_heatmaps=None, visualize_attn=False):
        all_heatmaps = self._heatmaps(views, input_heatmaps)                     # :106-113
        attns = None
        if visualize_attn and views is not None:                                # :115-120
            attns = torch.stack([self.attn(v) for v in views], 0)
        device = all_heatmaps[0].device
        B = all_heatmaps[0].shape[0]
        if self.use_root_gt:                                                     # :125-132
            num_person = meta[0]["num_person"]
            grid_centers = torch.zeros(B, self.num_cand, 5, device=device)
            grid_centers[:, :, 0:3] = meta[0]["roots_3d"].float().to(device)
            grid_centers[:, :, 3] = -1.0
            for i in range(B):
                n = int(num_person[i])
                grid_centers[i, :n, 3] = torch.arange(n, device=device, dtype=torch.float32)
                grid_centers[i, :n, 4] = 1.0
        else:
            _, _, _, grid_centers = self.root_net(all_heatmaps, meta)            # :134
        pred = torch.zeros(B, self.num_cand, self.num_joints, 5, device=device)
        pred[:, :, :, 3:] = grid_centers[:, :, 3:].reshape(B, -1, 1, 2)          # :136-137
        if not self.eval_rootnet_only and not self.train_only_rootnet and not self.train_only_2d:
            if all_heatmaps[0].is_cuda:
                pred[:, :, :, 0:3] = self.pose_net.forward_batched(all_heatmaps, meta, grid_centers)
            else:                                                                # :142-148 (no GPU: the reference's loop)
                for n in range(self.num_cand):
                    if bool((pred[:, n, 0, 3] >= 0).any()):
                        pred[:, n, :, 0:3] = self.pose_net(all_heatmaps, meta, grid_centers[:, n])
        if visualize_attn:
            return pred, all_heatmaps, grid_centers, attns
        return pred, all_heatmaps, grid_centers

    def forward(self, views1=None, meta1=None, targets_2d1=None, weights_2d1=None, targets_3d1=None, input_heatmaps1=None,
                views2=None, meta2=None, targets_2d2=None, weights_2d2=None, targets_3d2=None, input_heatmaps2=None,
                views3=None, meta3=None, targets_2d3=None, weights_2d3=None, targets_3d3=None, input_heatmaps3=None,
                inference=False, visualize_attn=False, epoch=0):
        if inference:                                                            # :221-222
            return self.do_inference(views=views1, meta=meta1, input_heatmaps=input_heatmaps1,
                                     visualize_attn=visualize_attn)
        raise NotImplementedError(
            "MultiPersonPoseNetSSV: only the inference path (forward(..., inference=True) / do_inference) is built; "
            "the self-supervised training forward of the reference (multi_person_posenet_ssv.py:197-501) is out of "
            "this repo's hot-path scope.  Train the supervised model (MODEL: multi_person_posenet) or call "
            "CuboidProposalNetSoft / reprojection.render_* directly.")


def get_multi_person_pose_net(cfg, is_train: bool = True):
    """factory with the reference's name and dispatch (multi_person_posenet_ssv.py:504-514)"""
    backbone = pose_resnet.get_pose_net(cfg, is_train=is_train) if cfg.BACKBONE_MODEL else None
    if cfg.get("WITH_ATTN", False):
        return MultiPersonPoseNetSSV(backbone, cfg, pose_resnet.get_pose_attn_net(cfg, is_train=is_train))
    return MultiPersonPoseNetSSV(backbone, cfg)
