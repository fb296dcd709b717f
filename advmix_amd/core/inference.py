"""Mirror of lib/core/inference.py:22-49 ``get_max_preds`` with the argmax on the device."""
import numpy as np
import torch

from .. import ops


def get_max_preds(batch_heatmaps):
    """Accepts a CUDA tensor [B,J,H,W] (device argmax, only [B,J] ints/floats cross PCIe)
    or, like the reference, a numpy array (host path for callers that already hold one)."""
    if isinstance(batch_heatmaps, torch.Tensor):
        B, J, H, W = batch_heatmaps.shape
        idx, mx = ops.heatmap_argmax(batch_heatmaps.detach())
        idx = idx.cpu().numpy().astype(np.int64)
        maxvals = mx.cpu().numpy().reshape(B, J, 1)
    else:
        assert batch_heatmaps.ndim == 4, 'batch_images should be 4-ndim'
        B, J, H, W = batch_heatmaps.shape
        flat = batch_heatmaps.reshape(B, J, -1)
        idx = np.argmax(flat, 2)
        maxvals = np.amax(flat, 2).reshape(B, J, 1)
    preds = np.zeros((B, J, 2), np.float32)
    preds[:, :, 0] = idx % W
    preds[:, :, 1] = np.floor(idx / W)
    preds *= np.tile(np.greater(maxvals, 0.0), (1, 1, 2)).astype(np.float32)
    return preds, maxvals
