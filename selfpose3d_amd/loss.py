"""Per-joint weighted MSE / L1 (API of /root/reference/lib/core/loss.py:39-74)."""
import torch.nn as nn
import torch.nn.functional as F


class _PerJoint(nn.Module):
    _fn = staticmethod(F.mse_loss)

    def forward(self, output, target, use_target_weight=False, target_weight=None):
        if use_target_weight:
            b, j = output.shape[:2]
            w = target_weight
            return self._fn(output.reshape(b, j, -1) * w, target.reshape(b, j, -1) * w)
        return self._fn(output, target)


class PerJointMSELoss(_PerJoint):
    _fn = staticmethod(F.mse_loss)


class PerJointL1Loss(_PerJoint):
    _fn = staticmethod(F.l1_loss)
