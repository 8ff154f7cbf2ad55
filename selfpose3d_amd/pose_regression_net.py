"""Per-proposal pose net: unproject (fine grid around each proposal) -> V2V -> soft-argmax.

Interface of /root/reference/lib/models/pose_regression_net.py:14-53; checkpoint keys
``v2v_net.*`` identical.  The soft-argmax (softmax(beta x) . grid) is one HIP reduction
kernel (sp3d_soft_argmax) in inference; under autograd it stays on torch ops.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .project_layer import ProjectLayer
from .v2v_net import V2VNet


class SoftArgmaxLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.beta = float(cfg.NETWORK.BETA)

    def forward(self, x, grids):
        if torch.is_grad_enabled() and x.requires_grad:
            B, C = x.shape[:2]
            p = F.softmax(self.beta * x.reshape(B, C, -1, 1), dim=2)
            return (p * grids.unsqueeze(1)).sum(dim=2)
        return _lib.soft_argmax(x, grids, self.beta)


class PoseRegressionNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.grid_size = [float(v) for v in cfg.PICT_STRUCT.GRID_SIZE]
        self.cube_size = [int(v) for v in cfg.PICT_STRUCT.CUBE_SIZE]
        self.project_layer = ProjectLayer(cfg)
        self.v2v_net = V2VNet(int(cfg.NETWORK.NUM_JOINTS), int(cfg.NETWORK.NUM_JOINTS))
        self.soft_argmax_layer = SoftArgmaxLayer(cfg)
        self.channels_last = False

    def use_channels_last(self, on: bool = True):
        self.channels_last = bool(on)
        self.v2v_net.to(memory_format=torch.channels_last_3d if on else torch.contiguous_format)
        return self

    @torch.no_grad()
    def forward_batched(self, all_heatmaps, meta, grid_centers, flip_xcoords=None, max_cubes_per_call: int = 8):
        """All person proposals of a batch at once (inference): grid_centers (B,K,5) -> pred (B,K,J,3).

        Replaces the reference's per-candidate loop (lib/models/multi_person_posenet.py:84-88, K
        calls of this net with the full batch): ONE indexed unprojection launch over the valid
        (b,k) pairs, V2V in chunks of ``max_cubes_per_call`` cubes, soft-argmax with in-kernel
        grids.  Every proposal is independent in eval mode (BatchNorm uses running statistics), so
        the result equals the loop's."""
        B, K = grid_centers.shape[:2]
        J = all_heatmaps[0].shape[1]
        device = all_heatmaps[0].device
        pred = torch.zeros(B, K, J, 3, device=device)
        pairs = torch.nonzero(grid_centers[:, :, 3] >= 0)                 # (P,2), one host sync
        P = int(pairs.shape[0])
        if P == 0:
            return pred
        bi, ki = pairs[:, 0], pairs[:, 1]
        centers = grid_centers[bi, ki, :3].contiguous()
        flip = None if flip_xcoords is None else flip_xcoords
        planar = self.v2v_net.wants_planar_input()                       # FFT opening conv: plain J-channel cubes
        direct = self.v2v_net.input_chunk_views(P, max_cubes_per_call, *self.cube_size, device) \
            if planar and J <= 16 else None
        if direct is not None:
            # the kernel writes every cube straight into the zero-padded input buffer of the FFT opening conv
            whole, chunks = direct
            self.project_layer.get_voxel(all_heatmaps, meta, self.grid_size, centers, self.cube_size, flip_xcoords=flip,
                                         want_grids=False, sample_of=bi, out=whole)
            outs, s0 = [], 0
            for n, view in chunks:
                y = self.v2v_net(view)[:n]          # a tail chunk is rounded up to a power of two (stale cubes, ignored)
                outs.append(_lib.soft_argmax_grid(y, centers[s0:s0 + n], self.grid_size, self.cube_size,
                                                  self.soft_argmax_layer.beta))
                s0 += n
            pred[bi, ki] = torch.cat(outs, 0)
            return pred
        cubes, _ = self.project_layer.get_voxel(all_heatmaps, meta, self.grid_size, centers, self.cube_size,
                                                flip_xcoords=flip, want_grids=False, pad_channels=not planar,
                                                channels_last=self.channels_last and not planar, sample_of=bi)
        outs = []
        for s0 in range(0, P, max_cubes_per_call):
            chunk = cubes[s0:s0 + max_cubes_per_call]
            cen = centers[s0:s0 + max_cubes_per_call]
            n = chunk.shape[0]
            # MIOpen re-tunes per distinct batch size: round the tail chunk up to a power of two by
            # repeating its last cube, so only log2(max)+1 shapes ever reach the conv stack
            m = 1 << (n - 1).bit_length()
            if m != n:
                chunk = torch.cat([chunk, chunk[-1:].expand(m - n, -1, -1, -1, -1)], 0)
                if self.channels_last and not planar:
                    chunk = chunk.contiguous(memory_format=torch.channels_last_3d)
            y = self.v2v_net(chunk)[:n]
            outs.append(_lib.soft_argmax_grid(y, cen, self.grid_size, self.cube_size, self.soft_argmax_layer.beta))
        pred[bi, ki] = torch.cat(outs, 0)
        return pred

    # cube counts the slot-batched training call is padded to (zero cubes in a BatchNorm group of their own, dropped
    # afterwards).  The number of valid cubes changes from step to step with the people in the frames (1 .. B x K), and every
    # distinct count is its own set of MIOpen convolution shapes = one kernel search each.  "auto" (default): round up to a
    # multiple of ceil(B K / 4), so the library only ever sees FOUR batch sizes (B = 2, K = 10: 5, 10, 15, 20); a tuple:
    # round up to the next listed size; None: never pad (every count searched once, kept in the user find-db)
    slot_pad_sizes = "auto"

    def can_batch_slots(self) -> bool:
        """all normalisation layers still are grouped BatchNorm (a SyncBatchNorm conversion, say, replaces them: then the
        per-slot loop is the only way to keep the reference's per-call statistics)"""
        from .grouped_bn import GroupedBatchNorm3d
        norms = [m for m in self.v2v_net.modules() if isinstance(m, nn.modules.batchnorm._NormBase)]
        return len(norms) > 0 and all(isinstance(m, GroupedBatchNorm3d) for m in norms)

    def forward_slots(self, sets, grid_centers, flags=None):
        """TRAIN-mode pose net on every candidate slot of a batch in ONE pass, with autograd - the reference's loop
        (lib/models/multi_person_posenet.py:84-88; multi_person_posenet_ssv.py:361-381 with two heat-map sets per slot):

            for n in range(num_cand):  if any(flag[:, n] >= 0):  for (heatmaps, meta, flip) in sets:
                pred[:, n] = pose_net(heatmaps, meta, grid_centers[:, n], flip)      # V2VNet on the VALID cubes of slot n

        Every such call is one BatchNorm batch of its own in the reference; here the cubes of all calls go through V2VNet
        together and its grouped BatchNorm layers (grouped_bn.py) keep each call's statistics apart and update the running
        statistics call by call in the loop's order.  ``sets``: list of (all_heatmaps, meta, flip_xcoords);
        grid_centers (B,K,5); ``flags``: its column 3 on the host (saves the sync).  -> list of (B,K,J,3), one per set."""
        from .grouped_bn import group_spec, bn_groups
        B, K = grid_centers.shape[:2]
        device = grid_centers.device
        J = sets[0][0][0].shape[1]
        if flags is None:
            flags = grid_centers[:, :, 3].detach().cpu()
        pairs = [(b, n) for n in range(K) for b in range(B) if float(flags[b, n]) >= 0]       # slot-major
        P, ns = len(pairs), len(sets)
        preds = [torch.zeros(B, K, J, 3, device=device, dtype=sets[0][0][0].dtype) for _ in sets]
        if P == 0:
            return preds
        bi = torch.tensor([b for b, _ in pairs], device=device)
        ki = torch.tensor([n for _, n in pairs], device=device)
        centers = grid_centers[bi, ki].detach()                    # (P,5), flags all >= 0 (proposals carry no gradient)
        slots = sorted({n for _, n in pairs})
        rank = {n: r for r, n in enumerate(slots)}
        # group of a cube = its call in the loop's order: (slot rank, set index)
        group_of, sizes = [], [0] * (len(slots) * ns)
        for si in range(ns):
            for _, n in pairs:
                g = rank[n] * ns + si
                group_of.append(g)
                sizes[g] += 1
        planar = self.v2v_net.wants_planar_input() and sets[0][0][0].is_cuda
        cl = self.channels_last and not planar
        cubes, grids = [], []
        fused_sa = sets[0][0][0].dtype == torch.float32 and next(self.v2v_net.parameters()).dtype == torch.float32   # HIP pair: fp32
        for heatmaps, meta, flip in sets:
            c, g = self.project_layer.get_voxel(heatmaps, meta, self.grid_size, centers, self.cube_size, flip_xcoords=flip,
                                                want_grids=not fused_sa, pad_channels=not planar, channels_last=cl, sample_of=bi)
            cubes.append(c)
            grids.append(g)
        x = cubes[0] if ns == 1 else torch.cat(cubes, 0)
        total, n_update = ns * P, len(sizes)
        pad = self.slot_pad_sizes
        if pad == "auto":
            q = -(-(B * K * ns) // 4)
            pad = (q, 2 * q, 3 * q, 4 * q)
        if pad:
            target = next((s for s in sorted(pad) if s >= total), total)
            if target > total:                                     # zero cubes, a group of their own, no running update
                x = torch.cat([x, x.new_zeros((target - total,) + tuple(x.shape[1:]))], 0)
                group_of += [len(sizes)] * (target - total)
                sizes = sizes + [target - total]
        if cl:
            x = x.contiguous(memory_format=torch.channels_last_3d)
        spec = group_spec(sizes, device, group_of=group_of, n_update=n_update)
        with bn_groups(self.v2v_net, spec):
            y = self.v2v_net(x)
        for si in range(ns):
            ys = y[si * P:(si + 1) * P]
            if fused_sa and ys.dtype == torch.float32:
                # soft-argmax with in-kernel voxel centres, forward and backward as HIP kernels (no (P,J,N,3) temporary)
                poses = _lib.soft_argmax_grid_autograd(ys, centers[:, :3], self.grid_size, self.cube_size, self.soft_argmax_layer.beta)
            else:
                poses = self.soft_argmax_layer(ys, grids[si])                            # (P,J,3)
            preds[si] = preds[si].index_put((bi, ki), poses)
        return preds

    def forward(self, all_heatmaps, meta, grid_centers, flip_xcoords=None):
        B, J = all_heatmaps[0].shape[:2]
        device = all_heatmaps[0].device
        pred = torch.zeros(B, J, 3, device=device)
        planar = self.v2v_net.wants_planar_input() and all_heatmaps[0].is_cuda
        cubes, grids = self.project_layer.get_voxel(all_heatmaps, meta, self.grid_size, grid_centers, self.cube_size,
                                                    flip_xcoords=flip_xcoords, pad_channels=not planar,
                                                    channels_last=self.channels_last and not planar)
        index = grid_centers[:, 3] >= 0
        if bool(index.any()):
            valid_cubes = self.v2v_net(cubes[index])
            pred[index] = self.soft_argmax_layer(valid_cubes, grids[index])
        return pred
