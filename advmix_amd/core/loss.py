"""Mirror of lib/core/loss.py: ``JointsMSELoss(use_target_weight, smooth_L1=False)``.
NOTE the reference's inverted flag (loss.py:16-21): the default ``smooth_L1=False`` - the only
way it is constructed (tools/train.py:111) - selects SmoothL1(beta=1); ``True`` selects MSE."""
import torch.nn as nn

from .. import ops


class JointsMSELoss(nn.Module):
    def __init__(self, use_target_weight, smooth_L1=False):
        super().__init__()
        self.use_target_weight = use_target_weight
        self.mse = bool(smooth_L1)

    def forward(self, output, target, target_weight):
        if output.dim() != 4:
            raise NotImplementedError('coordinate-regression outputs are not on the hot path')
        return ops.joints_loss(output, target, target_weight, self.use_target_weight, self.mse)
