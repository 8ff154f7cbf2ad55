"""Top-level supervised model: backbone -> root net (coarse unprojection) -> pose net (fine
unprojection per proposal).  Interface, outputs and state_dict prefixes (``backbone.``,
``root_net.``, ``pose_net.``) follow /root/reference/lib/models/multi_person_posenet.py:20-111.

MI355X-side differences (same results): the V views run through the backbone as one batch, the
heat-maps are re-tiled once and shared by the coarse and all fine projections, and in inference all
person proposals are unprojected in ONE launch and regressed in batched V2V calls instead of a
MAX_PEOPLE_NUM-iteration loop.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import pose_resnet
from .cuboid_proposal_net import CuboidProposalNet
from .loss import PerJointL1Loss, PerJointMSELoss
from .pose_regression_net import PoseRegressionNet


class MultiPersonPoseNet(nn.Module):
    batch_slots_in_training = os.environ.get("SP3D_BATCH_SLOTS", "1") != "0"      # train mode: the pose net sees all candidate slots in one pass (False: one call per slot)

    def __init__(self, backbone, cfg):
        super().__init__()
        self.num_cand = int(cfg.MULTI_PERSON.MAX_PEOPLE_NUM)
        self.num_joints = int(cfg.NETWORK.NUM_JOINTS)
        self.train_only_2d = bool(cfg.NETWORK.TRAIN_ONLY_2D)
        self.backbone = backbone
        if not self.train_only_2d:
            self.root_net = CuboidProposalNet(cfg)
            self.pose_net = PoseRegressionNet(cfg)
        self.USE_GT = bool(cfg.NETWORK.USE_GT)
        self.root_id = cfg.DATASET.ROOTIDX
        self.mse = PerJointMSELoss()
        self.l1 = PerJointL1Loss()

    def use_channels_last(self, on: bool = True):
        if not self.train_only_2d:
            self.root_net.use_channels_last(on)
            self.pose_net.use_channels_last(on)
        if self.backbone is not None:
            pose_resnet.set_backbone_memory_format(self.backbone, on)
        return self

    def heatmaps(self, views, input_heatmaps):
        if views is None:
            return input_heatmaps
        if hasattr(self.backbone, "forward_views"):
            return self.backbone.forward_views(views)
        return [self.backbone(v) for v in views]

    def forward(self, views=None, meta=None, targets_2d=None, weights_2d=None, targets_3d=None, input_heatmaps=None):
        all_heatmaps = self.heatmaps(views, input_heatmaps)
        device = all_heatmaps[0].device
        B = all_heatmaps[0].shape[0]
        zero = torch.zeros((), device=device)

        loss_2d = zero.clone()                                                   # multi_person_posenet.py:50-55
        if targets_2d is not None:
            for t, w, o in zip(targets_2d, weights_2d, all_heatmaps):
                loss_2d = loss_2d + self.mse(o, t.to(device), True, w.to(device))
            loss_2d = loss_2d / len(all_heatmaps)
        if self.train_only_2d:
            return loss_2d, all_heatmaps

        loss_3d = zero.clone()
        if self.USE_GT:                                                          # :61-68 proposals from ground truth
            num_person = meta[0]["num_person"]
            grid_centers = torch.zeros(B, self.num_cand, 5, device=device)
            grid_centers[:, :, 0:3] = meta[0]["roots_3d"].float().to(device)
            grid_centers[:, :, 3] = -1.0
            for i in range(B):
                n = int(num_person[i])
                grid_centers[i, :n, 3] = torch.arange(n, device=device, dtype=torch.float32)
                grid_centers[i, :n, 4] = 1.0
        else:
            root_cubes, grid_centers = self.root_net(all_heatmaps, meta)
            if targets_3d is not None:
                loss_3d = self.mse(root_cubes, targets_3d.to(device))
            del root_cubes

        pred = torch.zeros(B, self.num_cand, self.num_joints, 5, device=device)
        pred[:, :, :, 3:] = grid_centers[:, :, 3:].reshape(B, -1, 1, 2)          # :77-78
        loss_cord = zero.clone()

        if not self.training:
            pred[:, :, :, 0:3] = self.pose_net.forward_batched(all_heatmaps, meta, grid_centers)
            return pred, all_heatmaps, grid_centers, loss_2d, loss_3d, loss_cord

        # training: the reference's per-candidate loop (BatchNorm batch statistics are per call)
        have_gt = "joints_3d" in meta[0] and "joints_3d_vis" in meta[0]
        flags = grid_centers[:, :, 3].detach().cpu()                              # one sync for the whole loop
        count = 0
        # round 5: all slots in ONE pose-net pass whose BatchNorm layers keep every slot's statistics apart
        # (pose_regression_net.forward_slots) - the loop's results, 3 calls of (2,2,1) cubes -> 1 call of 5
        slots = None
        if self.batch_slots_in_training and all_heatmaps[0].is_cuda and self.pose_net.can_batch_slots():
            slots = self.pose_net.forward_slots([(all_heatmaps, meta, None)], grid_centers, flags=flags)[0]
        for n in range(self.num_cand):
            rows = flags[:, n] >= 0
            if not bool(rows.any()):
                continue
            single = slots[:, n] if slots is not None else self.pose_net(all_heatmaps, meta, grid_centers[:, n])
            pred[:, n, :, 0:3] = single.detach()
            if have_gt:                                                          # :92-100 running mean of L1 terms
                gt_3d = meta[0]["joints_3d"].float().to(device)
                vis = meta[0]["joints_3d_vis"].float().to(device)
                for i in torch.nonzero(rows).flatten().tolist():
                    g = int(flags[i, n])
                    count += 1
                    term = self.l1(single[i:i + 1], gt_3d[i:i + 1, g], True, vis[i:i + 1, g, :, 0:1])
                    loss_cord = (loss_cord * (count - 1) + term) / count
        # Sub-nets no loss term went through in THIS iteration on THIS rank (no valid proposal or no ground truth: pose
        # net; proposals from ground truth or no 3D target: root net; heat-maps handed in: backbone) are tied to the loss
        # with zero weight, so that every DDP rank runs the same gradient all-reduce with find_unused_parameters=False
        # (engine.anchor_unreached; the reference's zero-weighted dummy forwards, multi_person_posenet_ssv.py:290,429)
        from .engine import anchor_unreached
        reached = set()
        if count > 0:
            reached.add("pose_net")
        if not self.USE_GT and targets_3d is not None:
            reached.add("root_net")
        if views is not None and (targets_2d is not None or reached):
            reached.add("backbone")
        loss_cord = anchor_unreached(loss_cord, {"backbone": self.backbone, "root_net": self.root_net,
                                                 "pose_net": self.pose_net}, reached)
        return pred, all_heatmaps, grid_centers, loss_2d, loss_3d, loss_cord


def get_multi_person_pose_net(cfg, is_train: bool = True):
    backbone = pose_resnet.get_pose_net(cfg, is_train=is_train) if cfg.BACKBONE_MODEL else None
    return MultiPersonPoseNet(backbone, cfg)
