"""Root-localisation net: unproject (coarse grid) -> V2V -> NMS/top-k proposals.

Interface of /root/reference/lib/models/cuboid_proposal_net.py:86-122 (``CuboidProposalNet``)
and :13-83 (``ProposalLayer``); checkpoint keys ``v2v_net.*`` are identical.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .project_layer import ProjectLayer
from .proposal import nms_with_locations
from .v2v_net import V2VNet


class ProposalLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.grid_size = [float(v) for v in cfg.MULTI_PERSON.SPACE_SIZE]
        self.cube_size = [int(v) for v in cfg.MULTI_PERSON.INITIAL_CUBE_SIZE]
        self.grid_center = [float(v) for v in cfg.MULTI_PERSON.SPACE_CENTER]
        self.num_cand = int(cfg.MULTI_PERSON.MAX_PEOPLE_NUM)
        self.threshold = float(cfg.MULTI_PERSON.THRESHOLD)

    @staticmethod
    def match_to_gt(locs, gt_3d, num_person):
        """index of the nearest GT root within 500 mm, else -1 (cuboid_proposal_net.py:25-40), batched."""
        B, K, _ = locs.shape
        P = gt_3d.shape[1]
        d = torch.sqrt(((locs[:, :, None, :] - gt_3d[:, None, :, :]) ** 2).sum(-1))      # (B,K,P)
        live = torch.arange(P, device=locs.device)[None, None, :] < num_person.to(locs.device).view(B, 1, 1)
        d = torch.where(live, d, torch.full_like(d, float("inf")))
        min_dist, min_gt = d.min(dim=-1)
        out = min_gt.float()
        out[min_dist > 500.0] = -1.0
        return out

    def forward(self, root_cubes, meta):
        B = root_cubes.shape[0]
        if root_cubes.is_cuda and not (self.training and ("roots_3d" in meta[0] and "num_person" in meta[0])):
            # eval: [x,y,z, (score > THRESHOLD) - 1, score] straight from the NMS merge kernel
            from . import _lib
            return _lib.nms_proposals(root_cubes.detach(), self.num_cand, self.grid_size, self.grid_center, self.threshold)
        vals, _idx, locs = nms_with_locations(root_cubes, self.num_cand, self.grid_size, self.grid_center)
        grid_centers = torch.zeros(B, self.num_cand, 5, device=root_cubes.device)
        grid_centers[:, :, 0:3] = locs
        grid_centers[:, :, 4] = vals
        if self.training and ("roots_3d" in meta[0] and "num_person" in meta[0]):
            gt = meta[0]["roots_3d"].float().to(root_cubes.device)
            grid_centers[:, :, 3] = self.match_to_gt(locs, gt, meta[0]["num_person"])
        else:
            grid_centers[:, :, 3] = (vals > self.threshold).float() - 1.0
        return grid_centers


class CuboidProposalNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.grid_size = [float(v) for v in cfg.MULTI_PERSON.SPACE_SIZE]
        self.cube_size = [int(v) for v in cfg.MULTI_PERSON.INITIAL_CUBE_SIZE]
        self.grid_center = [float(v) for v in cfg.MULTI_PERSON.SPACE_CENTER]
        self.rootnet_roothm = bool(cfg.NETWORK.ROOTNET_ROOTHM)
        self.root_id = cfg.DATASET.ROOTIDX_PSEUDO
        self.project_layer = ProjectLayer(cfg)
        self.v2v_net = V2VNet(1 if self.rootnet_roothm else int(cfg.NETWORK.NUM_JOINTS), 1)
        self.proposal_layer = ProposalLayer(cfg)
        self.channels_last = False      # set by use_channels_last()

    def use_channels_last(self, on: bool = True):
        """run the V2V stack in torch.channels_last_3d (MIOpen NDHWC kernels, no internal transposes)"""
        self.channels_last = bool(on)
        self.v2v_net.to(memory_format=torch.channels_last_3d if on else torch.contiguous_format)
        return self

    def forward(self, all_heatmaps, meta, flip_xcoords=None):
        if self.rootnet_roothm:                                   # root-joint channel only (:103-108)
            hms = [a[:, self.root_id:self.root_id + 1].contiguous() for a in all_heatmaps]
        else:
            hms = all_heatmaps
        planar = self.v2v_net.wants_planar_input() and hms[0].is_cuda      # FFT opening conv: plain J-channel cubes
        # ... which the unprojection kernel writes straight into that conv's zero-padded input buffer
        direct = planar and hms[0].shape[1] <= 16 and not torch.is_grad_enabled()
        # ... on this grid as channels-last 16-channel cubes (its z pass is a HIP kernel that reads them as they are),
        cl16 = direct and self.v2v_net.wants_channels_last_cubes(*self.cube_size)
        # ... otherwise straight into that conv's zero-padded planar input buffer
        # ... or, round 6, not at all: the kernel keeps a 4 x 4 bundle of z columns in LDS and emits their z-spectrum, which
        # is what the opening conv computes from the cubes first (sp3d_unproject_fwd_zdft; bit-identical, one launch and
        # 2 x 32.8 MB of HBM traffic less at B = 4)
        sz = self.v2v_net.wants_zspectrum(*self.cube_size, hms[0].shape[1]) if cl16 else None
        if sz is not None and self.project_layer.io_dtype == torch.float32 and self.project_layer.jp_for(hms[0].shape[1]) == 16 \
                and self.project_layer.mode in ("auto", "nhwc"):
            from .v2v_net import TiledZSpectrum
            spec = self.project_layer.get_voxel_zspectrum(hms, meta, self.grid_size, [self.grid_center], self.cube_size, sz,
                                                          flip_xcoords=flip_xcoords)
            root_cubes = self.v2v_net(TiledZSpectrum(spec, *self.cube_size, sz)).squeeze(1)
            return root_cubes, self.proposal_layer(root_cubes, meta)
        out = self.v2v_net.input_view(hms[0].shape[0], *self.cube_size, hms[0].device) if direct and not cl16 else None
        cubes, _ = self.project_layer.get_voxel(hms, meta, self.grid_size, [self.grid_center], self.cube_size,
                                                flip_xcoords=flip_xcoords, want_grids=False,
                                                pad_channels=cl16 or not planar,
                                                channels_last=cl16 or (self.channels_last and not planar), out=out)
        root_cubes = self.v2v_net(cubes).squeeze(1)
        return root_cubes, self.proposal_layer(root_cubes, meta)
