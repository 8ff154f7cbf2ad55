"""3D joints -> per-view heat-maps for the self-supervised pose loss.

Reference: /root/reference/lib/utils/cameras.py:58-118 (``project_pose_batch``: pin-hole + distortion WITHOUT the
r^2 clamp of ``project_pose``, then the crop affine) and lib/models/multi_person_posenet_ssv.py:409-465 (the Python
loops over views x samples that render sigma-3 Gaussians of the projected joints, sum over people and clip).
Here: one vectorised, differentiable projection over (sample, view, person, joint) on the packed camera table, and the
HIP rendering kernels (``sp3d_render_joints_fwd/bwd``) behind an autograd Function - gradients reach the 3D joints.
"""
from __future__ import annotations

from typing import Optional

import torch

from .camera_pack import CAM_A, CAM_C, CAM_F, CAM_K, CAM_P, CAM_R, CAM_T


def project_joints(joints: torch.Tensor, cam: torch.Tensor, stride: float = 4.0,
                   trans: Optional[torch.Tensor] = None) -> torch.Tensor:
    """joints (B,P,J,3) world mm, cam (B,V,64) packed table on the same device -> (V,B,P,J,2) heat-map pixels.
    ``trans`` (B,2,3): one crop affine for all views of a sample (the reference passes ``meta[0]['trans']``);
    default: the per-view affine of the table."""
    B, V = cam.shape[:2]
    if joints.dtype == torch.float64:            # float64 callers (mixed-precision pins): the projection follows the poses' type
        cam = cam.double()
    R = cam[..., CAM_R:CAM_R + 9].reshape(B, V, 1, 1, 3, 3)
    T = cam[..., CAM_T:CAM_T + 3].reshape(B, V, 1, 1, 3)
    f = cam[..., CAM_F:CAM_F + 2].reshape(B, V, 1, 1, 2)
    c = cam[..., CAM_C:CAM_C + 2].reshape(B, V, 1, 1, 2)
    k = cam[..., CAM_K:CAM_K + 3].reshape(B, V, 1, 1, 3)
    p = cam[..., CAM_P:CAM_P + 2].reshape(B, V, 1, 1, 2)
    A = (cam[..., CAM_A:CAM_A + 6].reshape(B, V, 2, 3) if trans is None
         else trans.to(cam).reshape(B, 1, 2, 3).expand(B, V, 2, 3)).reshape(B, V, 1, 1, 2, 3)
    d = joints[:, None].to(cam.dtype) - T                                   # (B,V,P,J,3)
    xc = (R * d[..., None, :]).sum(-1)                                      # R (X - T)
    y = xc[..., :2] / (xc[..., 2:3] + 1e-5)
    r2 = (y * y).sum(-1, keepdim=True)
    radial = 1 + k[..., 0:1] * r2 + k[..., 1:2] * r2 ** 2 + k[..., 2:3] * r2 ** 3
    tan = p[..., 0:1] * y[..., 1:2] + p[..., 1:2] * y[..., 0:1]
    y = y * (radial + 2 * tan) + torch.cat([p[..., 1:2], p[..., 0:1]], -1) * r2
    px = f * y + c
    q = (A[..., :2] * px[..., None, :]).sum(-1) + A[..., 2]
    return (q / stride).permute(1, 0, 2, 3, 4)


def reprojection_heatmaps(joints: torch.Tensor, count: Optional[torch.Tensor], cam: torch.Tensor, h: int, w: int,
                          stride: float = 4.0, sigma: float = 3.0, trans: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B,P,J,3) predicted poses (``count`` (B,) valid people per sample) -> (V,B,J,h,w) clipped sums of Gaussians,
    differentiable w.r.t. the poses (GPU only: the rendering is the HIP kernel pair)."""
    from . import _lib
    kps = project_joints(joints, cam, stride, trans)                         # (V,B,P,J,2)
    V, B, P, J = kps.shape[:4]
    cnt = None if count is None else count.to(torch.int32).reshape(1, B).expand(V, B).reshape(-1)
    hm = _lib.render_joint_heatmaps(kps.reshape(V * B, P, J, 2), cnt, h, w, sigma)      # fp32 kernels, result in kps' type
    return hm.view(V, B, J, h, w)
