"""Self-supervised root net (``CuboidProposalNetSoft``): the supervised root net plus, in training, a
synthetic-root branch: random 3D roots -> 3D Gaussian target volume -> 2D Gaussian heat-maps in
every camera -> unprojection -> V2V.  Interface, return tuples and state_dict keys follow
/root/reference/lib/models/cuboid_proposal_net_soft.py:18-276.

Differences from the reference (same semantics):
  * inference path = CuboidProposalNet's (HIP unprojection, fused NMS/top-k);
  * ``train_rootnet`` is vectorised on the device - no ``.item()`` / ``searchsorted`` loop per root
    (:168-203) and no per-view/per-sample Python loops (:209-227) - and therefore also works for a
    per-GPU batch > 1 (the reference's ``expand(1, num_roots, 1)`` at :158 only works for B = 1,
    SURVEY App. D-3); heat-maps are emitted as (B,1,h,w) per view.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .cuboid_proposal_net import ProposalLayer
from .project_layer import ProjectLayer
from .v2v_net import V2VNet


class ProposalLayerSoft(ProposalLayer):
    """threshold-only flags, no GT matching (cuboid_proposal_net_soft.py:54-68)"""

    def forward(self, root_cubes, meta=None, grids=None):
        was = self.training
        self.training = False
        try:
            return super().forward(root_cubes, meta)
        finally:
            self.training = was


class CuboidProposalNetSoft(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.grid_size = [float(v) for v in cfg.MULTI_PERSON.SPACE_SIZE]
        self.cube_size = [int(v) for v in cfg.MULTI_PERSON.INITIAL_CUBE_SIZE]
        self.grid_center = [float(v) for v in cfg.MULTI_PERSON.SPACE_CENTER]
        self.root_id = cfg.DATASET.ROOTIDX
        self.rootnet_roothm = bool(cfg.NETWORK.get("ROOTNET_ROOTHM", False))
        self.rootnet_train_synth = bool(cfg.NETWORK.get("ROOTNET_TRAIN_SYNTH", False))
        self.max_num_people = int(cfg.MULTI_PERSON.MAX_PEOPLE_NUM)
        rng = cfg.NETWORK.get("ROOTNET_SYN_RANGE", [[2500.0, -2000.0], [1500.0, -1500.0], [250.0, -300.0]])
        self.project_layer = ProjectLayer(cfg)
        self.v2v_net = V2VNet(1 if self.rootnet_roothm else int(cfg.NETWORK.NUM_JOINTS), 1)
        self.proposal_layer = ProposalLayerSoft(cfg)
        self.channels_last = False
        self.cur_sigma = 200.0
        self.noise_std = 0.02
        self.hm_w, self.hm_h = [int(v) for v in cfg.NETWORK.HEATMAP_SIZE]
        self.stride = float(cfg.NETWORK.IMAGE_SIZE[0]) / float(self.hm_w)        # the reference hard-codes 4.0 (:216)
        lin = [np.linspace(-s / 2, s / 2, n) + c for s, n, c in zip(self.grid_size, self.cube_size, self.grid_center)]
        self.lo = [float(l.min() + r[0]) for l, r in zip(lin, rng)]
        self.hi = [float(l.max() + r[1]) for l, r in zip(lin, rng)]
        for name, l in zip(("grid1Dx", "grid1Dy", "grid1Dz"), lin):               # fp32 grids as :111-119
            self.register_buffer(name, torch.from_numpy(l).to(torch.float32), persistent=False)

    def use_channels_last(self, on: bool = True):
        self.channels_last = bool(on)
        self.v2v_net.to(memory_format=torch.channels_last_3d if on else torch.contiguous_format)
        return self

    # -- shared: heat-maps -> root cubes ---------------------------------------------------------------
    def _root_cubes(self, hms, meta, flip_xcoords):
        planar = self.v2v_net.wants_planar_input() and hms[0].is_cuda      # FFT opening conv: plain J-channel cubes
        out = self.v2v_net.input_view(hms[0].shape[0], *self.cube_size, hms[0].device) \
            if planar and hms[0].shape[1] <= 16 and not torch.is_grad_enabled() else None
        cubes, _ = self.project_layer.get_voxel(hms, meta, self.grid_size, [self.grid_center], self.cube_size,
                                                flip_xcoords=flip_xcoords, want_grids=False, pad_channels=not planar,
                                                channels_last=self.channels_last and not planar, out=out)
        return self.v2v_net(cubes).squeeze(1)

    def get_grid_centres(self, all_heatmaps, meta, flip_xcoords=None):
        if self.rootnet_roothm:
            hms = [a[:, self.root_id:self.root_id + 1].contiguous() for a in all_heatmaps]
        else:
            hms = all_heatmaps
        root_cubes = self._root_cubes(hms, meta, flip_xcoords)
        return root_cubes, self.proposal_layer(root_cubes, meta, None)

    # -- synthetic-root branch (training) ---------------------------------------------------------------
    @torch.no_grad()
    def sample_roots(self, batch_size, device, generator=None):
        """(B,R,3) random roots inside the shrunken capture space (:155-163); R in [1, max_people)"""
        R = int(torch.randint(1, max(2, self.max_num_people), (1,), generator=generator).item())
        u = torch.rand(batch_size, R, 2, generator=generator)
        xy = torch.stack([(self.hi[0] - self.lo[0]) * u[..., 0] + self.lo[0],
                          (self.hi[1] - self.lo[1]) * u[..., 1] + self.lo[1]], -1)
        z = (self.hi[2] - self.lo[2]) * torch.rand(batch_size, 1, 1, generator=generator) + self.lo[2]
        z = z.expand(batch_size, R, 1) + torch.randn(batch_size, R, 1, generator=generator) * 50.0
        return torch.cat([xy, z], -1).to(device=device, dtype=torch.float32)

    @torch.no_grad()
    def target_cubes(self, roots):
        """max over roots of exp(-d^2/2s^2), each root restricted to its +-3 sigma index window exactly as
        the searchsorted windows of :171-199 (grid point g is inside iff mu-3s <= g <= mu+3s), clipped."""
        s = self.cur_sigma
        gx, gy, gz = self.grid1Dx, self.grid1Dy, self.grid1Dz
        if roots.is_cuda:
            from . import _lib
            return _lib.gaussian_target_3d(roots, gx, gy, gz, s)
        mu = roots[:, :, None, None, None, :]                                    # (B,R,1,1,1,3)
        dx = gx.view(1, 1, -1, 1, 1) - mu[..., 0]
        dy = gy.view(1, 1, 1, -1, 1) - mu[..., 1]
        dz = gz.view(1, 1, 1, 1, -1) - mu[..., 2]
        inside = (dx.abs() <= 3 * s) & (dy.abs() <= 3 * s) & (dz.abs() <= 3 * s)
        g = torch.exp(-(dx ** 2 + dy ** 2 + dz ** 2) / (2 * s ** 2)) * inside
        return g.amax(dim=1).clamp_(0, 1)

    @torch.no_grad()
    def render_root_heatmaps(self, roots, meta, generator=None):
        """2D Gaussians (sigma 3 heat-map px) of the projected roots, summed over roots, clipped, plus
        N(0, 0.02) noise, clipped (:205-227; projection = cameras.project_point_radial_batch :58-108,
        i.e. NO r^2 clamp, then the crop affine ``meta[0]['trans']``).  -> list[V] of (B,1,h,w)."""
        dev = roots.device
        B = roots.shape[0]
        if roots.is_cuda:
            return self._render_hip(roots, meta, generator)
        ys = torch.arange(self.hm_h, device=dev, dtype=torch.float32).view(1, 1, -1, 1)
        xs = torch.arange(self.hm_w, device=dev, dtype=torch.float32).view(1, 1, 1, -1)
        trans = meta[0].get("trans")
        if trans is None:
            from .camera_pack import get_affine_transform_batch
            trans = torch.from_numpy(get_affine_transform_batch(
                meta[0]["center"].numpy(), meta[0]["scale"].numpy(), np.asarray(meta[0]["rotation"], np.float64),
                self.project_layer.img_size).astype(np.float32))
        trans = trans.to(dev, torch.float32).view(B, 2, 3)
        out = []
        for m in meta:
            cam = m["camera"]
            Rm = cam["R"].to(dev, torch.float32).view(B, 3, 3)
            T = cam["T"].to(dev, torch.float32).view(B, 1, 3)
            f = torch.stack([cam["fx"], cam["fy"]], -1).to(dev, torch.float32).view(B, 1, 2)
            c = torch.stack([cam["cx"], cam["cy"]], -1).to(dev, torch.float32).view(B, 1, 2)
            k = cam["k"].to(dev, torch.float32).view(B, 1, 3)
            p = cam["p"].to(dev, torch.float32).view(B, 1, 2)
            xc = torch.einsum("bij,brj->bri", Rm, roots - T)
            y = xc[..., :2] / (xc[..., 2:3] + 1e-5)
            r2 = (y ** 2).sum(-1, keepdim=True)
            radial = 1 + k[..., 0:1] * r2 + k[..., 1:2] * r2 ** 2 + k[..., 2:3] * r2 ** 3
            tan = p[..., 0:1] * y[..., 1:2] + p[..., 1:2] * y[..., 0:1]
            y = y * (radial + 2 * tan) + torch.cat([p[..., 1:2], p[..., 0:1]], -1) * r2
            px = f * y + c
            q = torch.einsum("bij,brj->bri", trans[:, :, :2], px) + trans[:, None, :, 2]
            q = q / self.stride
            g = torch.exp(-(((xs - q[..., 0, None, None]) / 3.0) ** 2) / 2 - (((ys - q[..., 1, None, None]) / 3.0) ** 2) / 2)
            hm = g.sum(1, keepdim=True).clamp_(0, 1)
            if self.noise_std > 0:
                hm = (hm + self.noise_std * torch.randn(hm.shape, generator=generator, device="cpu").to(dev)
                      if generator is not None else hm + self.noise_std * torch.randn_like(hm)).clamp_(0, 1)
            out.append(hm)
        return out

    @torch.no_grad()
    def _render_hip(self, roots, meta, generator=None):
        """GPU path of render_root_heatmaps: one launch for all views (sp3d_render_root_heatmaps)"""
        from . import _lib
        from .camera_pack import CAM_A, finish, pack_cameras
        B = roots.shape[0]
        tab = pack_cameras(meta, B, self.project_layer.img_size)
        trans = meta[0].get("trans")
        if trans is not None:                      # the reference applies meta[0]['trans'] to every view (:208)
            tab[:, :, CAM_A:CAM_A + 6] = trans.detach().cpu().numpy().reshape(B, 1, 6).astype(np.float32)
        else:
            tab[:, :, CAM_A:CAM_A + 6] = tab[:, :1, CAM_A:CAM_A + 6]
        finish(tab)                                # the derived block follows the edited affine: the table is valid for every kernel
        cam = torch.from_numpy(tab).to(roots.device)
        hm = _lib.render_root_heatmaps(roots, cam, self.hm_h, self.hm_w, self.stride)
        if self.noise_std > 0:
            noise = torch.randn(hm.shape, generator=generator, device="cpu").to(hm.device) if generator is not None \
                else torch.randn_like(hm)
            hm = (hm + self.noise_std * noise).clamp_(0, 1)
        return list(hm.unbind(0))

    def seed_sampler(self, base_seed: int, rank: int | None = None):
        """Give the synthetic-root sampler its own per-rank stream: with one process per GPU every rank would
        otherwise draw the SAME roots from identically seeded global RNGs (the reference's DataParallel threads
        share one process RNG, cuboid_proposal_net_soft.py:155-160, so its replicas differ by construction)."""
        from .distributed import rank_generator
        self.generator = rank_generator(base_seed, rank)
        return self

    def train_rootnet(self, batch_size, meta, pred_hms=None, flip_xcoords=None, generator=None):
        generator = generator if generator is not None else getattr(self, "generator", None)
        dev = self.grid1Dx.device
        roots = self.sample_roots(batch_size, dev, generator)
        target = self.target_cubes(roots)
        hms = self.render_root_heatmaps(roots, meta, generator)
        if not self.rootnet_roothm:      # the synthetic branch renders the root channel only
            J = self.v2v_net.front_layers[0].block[0].in_channels
            hms = [h.expand(-1, J, -1, -1).contiguous() for h in hms]
        return self._root_cubes(hms, meta, flip_xcoords), target

    def forward(self, all_heatmaps, meta, flip_xcoords=None):
        root_cubes, grid_centers = self.get_grid_centres(all_heatmaps, meta, flip_xcoords)
        if self.rootnet_train_synth and self.training:
            syn, target = self.train_rootnet(all_heatmaps[0].shape[0], meta, all_heatmaps, flip_xcoords)
            return root_cubes, syn, target, grid_centers
        return root_cubes, None, None, grid_centers
