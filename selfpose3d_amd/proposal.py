"""3D NMS + top-k on the GPU (fused HIP kernels), API of the reference's ``core.proposal``.

/root/reference/lib/core/proposal.py:35-48 issues max_pool3d, eq, mul, topk and six index
ops with (B,N) temporaries; ``nms`` here is two launches of sp3d_nms_topk (include/sp3d.h)
with a deterministic tie-break (larger value, then lower flat index - torch leaves it
unspecified).
"""
from __future__ import annotations

import torch

from . import _lib


def nms(root_cubes: torch.Tensor, max_num: int):
    """root_cubes (B,X,Y,Z) -> (topk_values (B,k) fp32, topk_unravel_index (B,k,3) int64)"""
    vals, idx, _ = _lib.nms_topk(root_cubes.detach(), int(max_num))
    return vals, idx


def nms_with_locations(root_cubes: torch.Tensor, max_num: int, grid_size, grid_center):
    """as ``nms`` plus the voxel centres in mm (ProposalLayer.get_real_loc,
    lib/models/cuboid_proposal_net.py:42-52) computed in the same launch."""
    return _lib.nms_topk(root_cubes.detach(), int(max_num), grid_size, grid_center)
