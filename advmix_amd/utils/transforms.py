"""Mirror of the two lib/utils/transforms.py entry points the validation path uses, on the device."""
import torch

from .. import ops


def flip_back(output_flipped, matched_parts, args=None, cfg=None, dim=None):
    """transforms.py:16-41 for heat-maps: reverse W and swap the left/right joint channels.
    ``output_flipped``: CUDA tensor [B,J,H,W]; returns a CUDA tensor (the reference returns numpy after a
    device->host copy - keep the result on the device and feed it to ``ops.flip_merge``/torch ops)."""
    if not isinstance(output_flipped, torch.Tensor) or output_flipped.dim() != 4:
        raise NotImplementedError('only the heat-map ([B,J,H,W] tensor) branch is on the validation path')
    return ops.flip_merge(None, output_flipped, matched_parts, shift=False, merge=False)
