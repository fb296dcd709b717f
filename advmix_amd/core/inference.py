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


def get_final_preds(config, args, batch_heatmaps, center, scale, cal_hm_coord=True, coord=None, reg_hm=False):
    """inference.py:52-95, heat-map branch (``cal_hm_coord`` and no regressed ``coord``): argmax, the
    POST_PROCESS quarter-pixel shift and the inverse crop transform in ONE device launch; only
    [B,J,3] floats come back.  Returns (preds float32 [B,J,2], maxvals float32 [B,J,1]) like the reference."""
    if not cal_hm_coord or coord is not None or reg_hm:
        raise NotImplementedError('only the heat-map branch of get_final_preds is on the validation path')
    if not isinstance(batch_heatmaps, torch.Tensor):
        raise TypeError('get_final_preds needs the CUDA heat-map tensor (no host copy on this path)')
    _, preds, mx = ops.final_preds(batch_heatmaps, center, scale, bool(config.TEST.POST_PROCESS))
    packed = torch.cat([preds, mx.unsqueeze(2)], dim=2).cpu().numpy()       # one D2H of [B,J,3]
    return np.ascontiguousarray(packed[:, :, 0:2]), np.ascontiguousarray(packed[:, :, 2:3])
