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

    def blend(self, output, target_a, scale_a, target_b, scale_b, target_weight):
        """``scale_a * self(output, target_a, w) + scale_b * self(output, target_b, w)`` as one launch and one autograd node
        (``target_b`` None: ``scale_a * self(output, target_a, w)``): what lib/core/function.py:151-153 and :161 spell with
        torch arithmetic around the criterion.  The loops use it when the criterion offers it."""
        if output.dim() != 4:
            raise NotImplementedError('coordinate-regression outputs are not on the hot path')
        return ops.joints_loss_blend(output, target_a, scale_a, target_b, scale_b, target_weight, self.use_target_weight, self.mse)
