"""Self-supervised top-level model (``MODEL: multi_person_posenet_ssv``): inference and the training forward.

Interface, return tuples, loss-dict keys and state_dict prefixes (``backbone.``, ``attn.``, ``pose_net.``,
``root_net.``) follow /root/reference/lib/models/multi_person_posenet_ssv.py:29-501.

``do_inference`` (:105-153) = backbone per view -> ``CuboidProposalNetSoft`` (HIP unprojection, V2V, fused NMS/top-k) ->
pose net on every valid proposal; the views run through the backbone as one batch and all proposals are unprojected
in ONE launch and regressed in batched V2V calls (the reference loops over MAX_PEOPLE_NUM candidates, :142-148).

``forward`` without ``inference`` is the self-supervised training step (:197-501), assembled from the pieces this repo
already has: three view sets (two affine-augmented, one plain) -> heat-maps (+ attention maps) -> root net on set 3
(frozen stage) or on all three with the synthetic-root branch (root-net stage) -> pose net on sets 1 and 2 for every
valid proposal -> each set's poses re-projected into the OTHER set's crops (``reprojection.project_joints`` ==
``cameras.project_pose_batch``) and rendered as Gaussian heat-maps by the HIP kernel pair ``sp3d_render_joints_fwd/bwd``
(differentiable; the reference builds (P,J,h,w) temporaries per view and sample in Python loops, :409-465) -> attention-
weighted MSE against the pseudo-label maps, attention regulariser, optional Hungarian L1 term (:155-194).
Deliberate differences, both value-neutral: the reference keeps parameters in the autograd graph with zero-weighted
DUMMY FORWARDS (a (1,3,512,960) backbone pass, :290; a zero cube through the pose net's V2V, :429,496,499); here the same
zero-valued terms come from ``engine.zero_anchor`` (no extra convolutions).  Datasets, RandAugment and the evaluation
code of the SSL pipeline are out of scope; ``synthetic_dataset.SyntheticPanopticSSV`` feeds the loop.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import pose_resnet
from .cuboid_proposal_net_soft import CuboidProposalNetSoft
from .pose_regression_net import PoseRegressionNet


class MultiPersonPoseNetSSV(nn.Module):
    batch_slots_in_training = os.environ.get("SP3D_BATCH_SLOTS", "1") != "0"      # train mode: the pose net sees all slots (and both view sets) in one pass (False: loop)

    def __init__(self, backbone, cfg, attn=None):
        super().__init__()
        self.num_cand = int(cfg.MULTI_PERSON.MAX_PEOPLE_NUM)
        self.num_joints = int(cfg.NETWORK.NUM_JOINTS)
        self.backbone = backbone
        self.WITH_ATTN = bool(cfg.get("WITH_ATTN", False))
        if self.WITH_ATTN:
            self.attn = attn
        self.attn_weight = float(cfg.get("ATTN_WEIGHT", 0.1))
        self.USE_L1 = bool(cfg.get("USE_L1", False))
        self.L1_WEIGHT = float(cfg.get("L1_WEIGHT", 0.1))
        self.L1_ATTN = bool(cfg.get("L1_ATTN", False))
        self.L1_EPOCH = int(cfg.TRAIN.get("L1_EPOCH", 5)) if "TRAIN" in cfg else 5
        self.width, self.height = int(cfg.NETWORK.IMAGE_SIZE[0]), int(cfg.NETWORK.IMAGE_SIZE[1])
        self.heatmap_width, self.heatmap_height = int(cfg.NETWORK.HEATMAP_SIZE[0]), int(cfg.NETWORK.HEATMAP_SIZE[1])
        self.rootnet_train_synth = bool(cfg.NETWORK.get("ROOTNET_TRAIN_SYNTH", False))
        self.freeze_rootnet = bool(cfg.NETWORK.get("FREEZE_ROOTNET", False))
        self.single_aug_training_posenet = bool(cfg.NETWORK.get("SINGLE_AUG_TRAINING_POSENET", False))
        self.root_reg_loss = bool(cfg.NETWORK.get("ROOT_CONSISTENCY_LOSS", True))
        self.weight_root_syn = float(cfg.NETWORK.get("WEIGHT_ROOT_SYN", 100.0))
        self.weight_root_reg = float(cfg.NETWORK.get("WEIGHT_ROOT_REG", 1.0))
        self.init_train_epochs_rootnet = int(cfg.NETWORK.get("INIT_TRAIN_EPOCHS_ROOTNET", 0))
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
                pose_resnet.set_backbone_memory_format(net, on)
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
            attns = torch.stack(self.attn.forward_views(views), 0)
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

    # -- training forward ------------------------------------------------------------------------------------
    def l1_matching_loss(self, kps, count, meta):
        """Hungarian-matched mean |pred - pseudo-GT| of the re-projected 2D joints, per (view, sample)
        (multi_person_posenet_ssv.py:155-194).  kps (V,B,P,J,2) network-input pixels, count (B,) valid poses.
        (The reference normalises ``meta[...]['joints']`` and the predictions IN PLACE; here the same quotients are
        formed out of place.)"""
        from scipy.optimize import linear_sum_assignment
        V, B = kps.shape[:2]
        dev = kps.device
        size = torch.tensor([float(self.width), float(self.height)], device=dev)
        losses = torch.zeros(V * B, device=dev)
        for nv in range(V):
            joints = meta[nv]["joints"].to(dev, torch.float32)
            vis = meta[nv]["joints_vis"].to(dev, torch.float32)
            for bs in range(B):
                num_gt = int((joints[bs].sum(-1).sum(-1) != 0).sum())
                num_pred = int(count[bs])
                if num_pred == 0 or num_gt == 0:
                    continue
                target = joints[bs, :num_gt] / size
                pred = kps[nv, bs, :num_pred] / size
                d = ((pred[None] - target[:, None]) * vis[bs, :num_gt, None]).abs().mean(dim=(-1, -2))   # (gt, pred)
                mx, my = linear_sum_assignment(d.detach().cpu().numpy())
                losses[nv * B + bs] = d[torch.as_tensor(mx, device=dev), torch.as_tensor(my, device=dev)].sum()
        if self.L1_ATTN:                                                     # drop the worst (view, sample) (:188-191)
            mask = torch.ones(V * B, device=dev)
            mask[torch.argmax(losses)] = 0.0
            return (losses * mask).sum() / (B * V - 1)
        return losses.mean()

    def _pose_pass(self, heatmaps, meta, grid_centers, flip, poses=None):
        """pose net on every candidate that has a valid proposal in some sample (:361-381): (B, num_cand, J, 5);
        ``poses``: (B, num_cand, J, 3) already computed by the slot-batched pass"""
        B = grid_centers.shape[0]
        tail = grid_centers[:, :, 3:].reshape(B, -1, 1, 2).expand(-1, -1, self.num_joints, -1)
        if poses is not None:
            return torch.cat([poses, tail], dim=3)
        if self.batch_slots_in_training and heatmaps[0].is_cuda and self.pose_net.can_batch_slots():
            return torch.cat([self.pose_net.forward_slots([(heatmaps, meta, flip)], grid_centers)[0], tail], dim=3)
        pred = torch.zeros(B, self.num_cand, self.num_joints, 5, device=grid_centers.device)
        pred[:, :, :, 3:] = grid_centers[:, :, 3:].reshape(B, -1, 1, 2)
        flags = grid_centers[:, :, 3].detach().cpu()
        for n in range(self.num_cand):
            if bool((flags[:, n] >= 0).any()):
                pred[:, n, :, 0:3] = self.pose_net(heatmaps, meta, grid_centers[:, n], flip_xcoords=flip)
        return pred

    def _reprojection_maps(self, pred, count, cam, trans):
        from .reprojection import reprojection_heatmaps
        return reprojection_heatmaps(pred[..., :3], count, cam, self.heatmap_height, self.heatmap_width, 4.0, 3.0, trans)

    def forward(self, views1=None, meta1=None, targets_2d1=None, weights_2d1=None, targets_3d1=None, input_heatmaps1=None,
                views2=None, meta2=None, targets_2d2=None, weights_2d2=None, targets_3d2=None, input_heatmaps2=None,
                views3=None, meta3=None, targets_2d3=None, weights_2d3=None, targets_3d3=None, input_heatmaps3=None,
                inference=False, visualize_attn=False, epoch=0):
        if inference:                                                            # :221-222
            return self.do_inference(views=views1, meta=meta1, input_heatmaps=input_heatmaps1,
                                     visualize_attn=visualize_attn)
        import torch.nn.functional as F
        from .engine import anchor_unreached, zero_anchor
        hm3 = self._heatmaps(views3, input_heatmaps3)                             # set 3: no augmentation (:226-232)
        attns1 = attns2 = None
        if self.WITH_ATTN:                                                       # :234-244
            if views1 is not None:
                attns1 = torch.stack(self.attn.forward_views(views1), 0)
            if views2 is not None:
                attns2 = torch.stack(self.attn.forward_views(views2), 0)
        hm1 = self._heatmaps(views1, input_heatmaps1)
        hm2 = self._heatmaps(views2, input_heatmaps2)
        device = hm1[0].device
        B = hm1[0].shape[0]
        zero = torch.zeros((), device=device)

        def anchored(module):        # the value of the reference's zero-weighted dummy forwards, without running them
            a = zero_anchor(module.parameters()) if module is not None else None
            return zero.clone() if a is None else a

        losses = {}
        # sub-nets some loss term has gone through so far; done() ties the others to loss_2d with zero weight on every
        # return path (engine.anchor_unreached), so DDP's static-graph mode holds for every flag combination
        subnets = {"backbone": self.backbone, "attn": self.attn if self.WITH_ATTN else None,
                   "root_net": getattr(self, "root_net", None), "pose_net": getattr(self, "pose_net", None)}
        reached = set()

        def done(*ret):
            losses["loss_2d"] = anchor_unreached(losses["loss_2d"], subnets, reached)
            return ret
        if targets_2d1 is not None and targets_2d2 is not None:                  # :281-288
            t1 = torch.stack([t.to(device) for t in targets_2d1])
            t2 = torch.stack([t.to(device) for t in targets_2d2])
            t3 = torch.stack([t.to(device) for t in targets_2d3])
            losses["loss_2d"] = (F.mse_loss(t1, torch.stack(list(hm1))) + F.mse_loss(t2, torch.stack(list(hm2))) +
                                 F.mse_loss(t3, torch.stack(list(hm3)))) / 3.0
            if views1 is not None and views2 is not None and views3 is not None:
                reached.add("backbone")
        else:
            t1 = t2 = None
            losses["loss_2d"] = anchored(self.backbone)                           # :290
            reached.add("backbone")
        if self.train_only_2d:
            return done(None, hm3, None, losses)

        flip1 = meta1[0].get("hflip") if meta1 is not None else None
        flip2 = meta2[0].get("hflip") if meta2 is not None else None
        flip3 = meta3[0].get("hflip") if meta3 is not None else None
        if self.use_root_gt:                                                     # :297-304
            num_person = meta3[0]["num_person"]
            grid_centers = torch.zeros(B, self.num_cand, 5, device=device)
            grid_centers[:, :, 0:3] = meta3[0]["roots_3d"].float().to(device)
            grid_centers[:, :, 3] = -1.0
            for i in range(B):
                n = int(num_person[i])
                grid_centers[i, :n, 3] = torch.arange(n, device=device, dtype=torch.float32)
                grid_centers[i, :n, 4] = 1.0
        elif self.freeze_rootnet:                                                # :306-307
            # nothing differentiable comes out of a frozen root net (proposal indices, flags, scores that only travel
            # into the returned pred[..., 3:]): run it without a graph - in eval mode that is the fused inference plan
            with torch.no_grad():
                _, _, _, grid_centers = self.root_net([h.detach() for h in hm3], meta3, flip_xcoords=flip3)
        elif self.rootnet_train_synth:                                           # :309-330
            main1, syn1, tgt1, _ = self.root_net(hm1, meta1, flip_xcoords=flip1)
            main2, syn2, tgt2, _ = self.root_net(hm2, meta2, flip_xcoords=flip2)
            main3, syn3, tgt3, grid_centers = self.root_net(hm3, meta3, flip_xcoords=flip3)
            losses["loss_root_syn"] = self.weight_root_syn * (F.mse_loss(syn1, tgt1) + F.mse_loss(syn2, tgt2) +
                                                              F.mse_loss(syn3, tgt3))
            main3 = main3.detach()
            reached.add("root_net")
            if self.root_reg_loss:
                losses["loss_root_reg"] = self.weight_root_reg * (F.mse_loss(main1, main3) + F.mse_loss(main2, main3))
        else:                                                                    # :331-335
            rc1, _, _, _ = self.root_net(hm1, meta1, flip_xcoords=flip1)
            rc2, _, _, _ = self.root_net(hm2, meta2, flip_xcoords=flip2)
            _, _, _, grid_centers = self.root_net(hm3, meta3, flip_xcoords=flip3)
            losses["loss_root_reg"] = F.mse_loss(rc1, targets_3d1.to(device)) + F.mse_loss(rc2, targets_3d2.to(device))
            reached.add("root_net")
        if self.train_only_rootnet:
            return done(None, hm3, grid_centers, losses)

        if epoch < self.init_train_epochs_rootnet:                               # :497-499
            losses["loss_pose3d_ssv"] = anchored(self.pose_net)
            reached.add("pose_net")
            return done(None, hm3, grid_centers, losses)

        from .camera_pack import pack_cameras
        count = (grid_centers[:, :, 3] >= 0).sum(1)                              # valid proposals lead the list (:386)
        both = None
        if (self.batch_slots_in_training and not self.single_aug_training_posenet and hm1[0].is_cuda and
                self.pose_net.can_batch_slots()):
            # both view sets and all slots in ONE pose-net pass; the grouped BatchNorm layers keep the statistics of each of
            # the reference's calls apart and update the running statistics in ITS order: slot 0 set 1, slot 0 set 2, slot 1
            # set 1, ... (multi_person_posenet_ssv.py:361-381)
            both = self.pose_net.forward_slots([(hm1, meta1, flip1), (hm2, meta2, flip2)], grid_centers)
        pred1 = self._pose_pass(hm1, meta1, grid_centers, flip1, both[0] if both is not None else None)
        cam = torch.from_numpy(pack_cameras(meta1, B, [self.width, self.height])).to(device)   # proj_cameras (:397)
        trans1 = meta1[0]["trans"]
        have_people = int(count[0]) > 0              # the reference tests sample 0 only (pred1[0].shape[0] > 0, :409,433)
        if self.single_aug_training_posenet:                                     # :409-429
            pred_out = pred1.detach().clone()
            if have_people:
                maps11 = self._reprojection_maps(pred1, count, cam, trans1)
                losses["loss_pose3d_ssv"] = F.mse_loss(t1, maps11) if t1 is not None else zero.clone()
                if t1 is not None:
                    reached.add("pose_net")
            else:
                losses["loss_pose3d_ssv"] = anchored(self.pose_net)
                reached.add("pose_net")
            return done(pred_out, hm3, grid_centers, losses)

        pred2 = self._pose_pass(hm2, meta2, grid_centers, flip2, both[1] if both is not None else None)
        pred_out = pred2.detach().clone()
        trans2 = meta2[0]["trans"]
        if have_people:                                                          # :433-486
            from .reprojection import project_joints
            maps21 = self._reprojection_maps(pred2, count, cam, trans1)           # set 2's poses in set 1's crops
            maps12 = self._reprojection_maps(pred1, count, cam, trans2)           # set 1's poses in set 2's crops
            l1 = l2 = zero.clone()
            if t1 is not None:
                l1 = (F.mse_loss(t1, maps21, reduction="none") * attns1).mean() if self.WITH_ATTN else F.mse_loss(t1, maps21)
            if t2 is not None:
                l2 = (F.mse_loss(t2, maps12, reduction="none") * attns2).mean() if self.WITH_ATTN else F.mse_loss(t2, maps12)
            losses["loss_pose3d_ssv"] = l1 + l2
            if t1 is not None or t2 is not None:
                reached.add("pose_net")
            if self.WITH_ATTN:
                reached.add("attn")
            if self.WITH_ATTN:
                losses["loss_attn_ssv"] = (F.mse_loss(attns1, torch.ones_like(attns1)) +
                                           F.mse_loss(attns2, torch.ones_like(attns2))) * self.attn_weight
            if self.USE_L1 and epoch >= self.L1_EPOCH:
                kps12 = project_joints(pred1[..., :3], cam, 1.0, trans2)          # network-input pixels (V,B,P,J,2)
                kps21 = project_joints(pred2[..., :3], cam, 1.0, trans1)
                losses["loss_pose3d_l1_ssv"] = (self.l1_matching_loss(kps12, count, meta2) +
                                                self.l1_matching_loss(kps21, count, meta1)) * self.L1_WEIGHT
        else:                                                                    # :487-495
            if self.WITH_ATTN:
                losses["loss_attn_ssv"] = (F.mse_loss(attns1, torch.ones_like(attns1)) +
                                           F.mse_loss(attns2, torch.ones_like(attns2))) * 0.0
                reached.add("attn")
            if self.USE_L1 and epoch >= self.L1_EPOCH:
                losses["loss_pose3d_l1_ssv"] = zero.clone()
            losses["loss_pose3d_ssv"] = anchored(self.pose_net)
            reached.add("pose_net")
        return done(pred_out, hm3, grid_centers, losses)


def get_multi_person_pose_net(cfg, is_train: bool = True):
    """factory with the reference's name and dispatch (multi_person_posenet_ssv.py:504-514)"""
    backbone = pose_resnet.get_pose_net(cfg, is_train=is_train) if cfg.BACKBONE_MODEL else None
    if cfg.get("WITH_ATTN", False):
        return MultiPersonPoseNetSSV(backbone, cfg, pose_resnet.get_pose_attn_net(cfg, is_train=is_train))
    return MultiPersonPoseNetSSV(backbone, cfg)
