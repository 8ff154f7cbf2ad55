"""``torch.ops.selfpose3d_mi.*`` - the operator-registry face of the C ABI (SURVEY.md §8(b) "Torch op the build
registers").  Thin: argument checks, output allocation and the ctypes call of ``_lib``; HIP dispatch key only - on
CPU tensors the dispatcher raises NotImplementedError (there is no CPU implementation in the product; the CPU
restatement under ``oracle/`` is test infrastructure).

  unproject_fwd(hm, cam, centers, valid, grid_size[3], cube_size[3], img_size[2], hm_size[2], joints=-1)
      -> (cubes (B,J,X,Y,Z), grids (B,N,3))            ProjectLayer.get_voxel, lib/models/project_layer.py:42-102
      hm   (V,B,h,w,Jp) channels-last (Jp in 4/8/12/16, channels >= joints are padding) or (V,B,J,h,w) planar
      cam  (B,V,64) fp32 table of include/sp3d.h (camera_pack.pack_cameras)
      centers (B,3) fp32 mm, valid (B,) uint8 (0 = skipped row, written as zeros, project_layer.py:54)
  unproject_bwd(grad_cubes, hm, cam, centers, valid, grid_size, cube_size, img_size, hm_size, joints=-1)
      -> grad_hm, same shape/layout as hm             autograd of the above w.r.t. the heat-maps (J <= 16: pass-mask forward +
                                                      the packed scatter of the C ABI's training pair; else the planar scatter)
Autograd is registered, so ``unproject_fwd`` is differentiable in ``hm`` (grids carry no gradient, as in the reference).
"""
from typing import Sequence, Tuple

import torch

from . import _lib

_NS = "selfpose3d_mi"


def _layout(hm: torch.Tensor, hm_size: Sequence[int], joints: int):
    w, h = int(hm_size[0]), int(hm_size[1])
    if hm.dim() != 5:
        raise _lib.Sp3dError("hm must be (V,B,h,w,Jp) or (V,B,J,h,w)")
    cl = tuple(hm.shape[2:4]) == (h, w)
    pl = tuple(hm.shape[3:5]) == (h, w)
    if cl and pl and joints > 0:                      # ambiguous shape: the declared joint count decides
        pl = int(hm.shape[2]) == joints
        cl = not pl
    if cl:
        jp = int(hm.shape[4])
        return _lib.LAYOUT_NHWC, jp, (joints if joints > 0 else jp)
    if pl:
        return _lib.LAYOUT_PLANAR, 0, int(hm.shape[2])
    raise _lib.Sp3dError(f"hm {tuple(hm.shape)} matches neither layout for heat-map size (w={w}, h={h})")


@torch.library.custom_op(f"{_NS}::unproject_fwd", mutates_args=(), device_types="cuda")
def unproject_fwd(hm: torch.Tensor, cam: torch.Tensor, centers: torch.Tensor, valid: torch.Tensor,
                  grid_size: Sequence[float], cube_size: Sequence[int], img_size: Sequence[int],
                  hm_size: Sequence[int], joints: int = -1) -> Tuple[torch.Tensor, torch.Tensor]:
    layout, jp, J = _layout(hm, hm_size, joints)
    hm = hm.contiguous()
    V, B = int(hm.shape[0]), int(hm.shape[1])
    cubes, grids = _lib.unproject_fwd([hm[c] for c in range(V)], layout, jp, cam.contiguous().float(),
                                      centers.contiguous().float(), valid.contiguous().to(torch.uint8), B, J,
                                      int(hm_size[1]), int(hm_size[0]), cube_size, grid_size, img_size, True)
    return cubes, grids


@unproject_fwd.register_fake
def _(hm, cam, centers, valid, grid_size, cube_size, img_size, hm_size, joints=-1):
    _, _, J = _layout(hm, hm_size, joints)
    B = hm.shape[1]
    X, Y, Z = (int(c) for c in cube_size)
    return hm.new_empty((B, J, X, Y, Z), dtype=torch.float32), hm.new_empty((B, X * Y * Z, 3), dtype=torch.float32)


@torch.library.custom_op(f"{_NS}::unproject_bwd", mutates_args=(), device_types="cuda")
def unproject_bwd(grad_cubes: torch.Tensor, hm: torch.Tensor, cam: torch.Tensor, centers: torch.Tensor,
                  valid: torch.Tensor, grid_size: Sequence[float], cube_size: Sequence[int], img_size: Sequence[int],
                  hm_size: Sequence[int], joints: int = -1) -> torch.Tensor:
    layout, jp, J = _layout(hm, hm_size, joints)
    V, B = int(hm.shape[0]), int(hm.shape[1])
    w, h = int(hm_size[0]), int(hm_size[1])
    camf, cen, val = cam.contiguous().float(), centers.contiguous().float(), valid.contiguous().to(torch.uint8)
    # (an NHWC input padded to more than 16 channels, or not to a multiple of 4, is outside the packed kernels' range:
    # it takes the planar scatter below, as before round 4)
    if J <= 16 and w >= 2 and h >= 2 and (layout != _lib.LAYOUT_NHWC or (jp <= 16 and jp % 4 == 0)):
        # the training pair of the C ABI (include/sp3d.h): one forward pass over channels-last maps for the clamp pass mask,
        # then the line-coalesced / block-merge scatter - 10-50x faster than the planar scatter below, same sums
        if layout == _lib.LAYOUT_NHWC:
            packed = hm.contiguous().float()
        else:
            jp = (J + 3) // 4 * 4
            packed = _lib.pack_heatmaps([hm[c] for c in range(V)], jp=jp)
        X, Y, Z = (int(c) for c in cube_size)
        mask = torch.empty((B, X * Y * Z), dtype=torch.int16, device=hm.device)
        _lib.unproject_fwd([packed[c] for c in range(V)], _lib.LAYOUT_NHWC, jp, camf, cen, val, B, J, h, w, cube_size, grid_size,
                           img_size, False, pass_mask=mask)
        g = _lib.unproject_bwd_packed(camf, cen, val, grad_cubes, mask, B, V, J, jp, h, w, cube_size, grid_size, img_size,
                                      return_packed=True)                  # (V,B,h,w,jp), pad channels zero
        if layout == _lib.LAYOUT_NHWC:
            return g
        return g[..., :J].permute(0, 1, 4, 2, 3).contiguous()
    planar = hm if layout == _lib.LAYOUT_PLANAR else hm[..., :J].permute(0, 1, 4, 2, 3)
    planar = planar.contiguous().float()
    g = _lib.unproject_bwd([planar[c] for c in range(V)], camf, cen, val, grad_cubes, cube_size, grid_size, img_size)
    g = torch.stack(list(g), 0)                                            # (V,B,J,h,w)
    if layout == _lib.LAYOUT_PLANAR:
        return g
    out = torch.zeros_like(hm, dtype=torch.float32)
    out[..., :J] = g.permute(0, 1, 3, 4, 2)
    return out


@unproject_bwd.register_fake
def _(grad_cubes, hm, cam, centers, valid, grid_size, cube_size, img_size, hm_size, joints=-1):
    return hm.new_empty(hm.shape, dtype=torch.float32)


def _setup(ctx, inputs, output):
    hm, cam, centers, valid, grid_size, cube_size, img_size, hm_size, joints = inputs
    ctx.save_for_backward(hm, cam, centers, valid)
    ctx.mark_non_differentiable(output[1])           # grids carry no gradient
    ctx.geom = (list(grid_size), list(cube_size), list(img_size), list(hm_size), joints)


def _backward(ctx, grad_cubes, grad_grids):
    hm, cam, centers, valid = ctx.saved_tensors
    gs, cs, im, hs, joints = ctx.geom
    g = torch.ops.selfpose3d_mi.unproject_bwd(grad_cubes.contiguous(), hm, cam, centers, valid, gs, cs, im, hs, joints)
    return g.to(hm.dtype), None, None, None, None, None, None, None, None


unproject_fwd.register_autograd(_backward, setup_context=_setup)
