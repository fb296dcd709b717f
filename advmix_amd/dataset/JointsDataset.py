"""Device side of lib/dataset/JointsDataset.py::generate_target (:412-491, gaussian branch): the whole
batch's heat-maps and target weights in one launch instead of per-sample numpy in the workers."""
import ctypes

import numpy as np
import torch

from .._lib import call


def gaussian_patch(sigma):
    """JointsDataset.py:463-470 verbatim arithmetic (numpy float32), computed once: the unnormalised
    (2*3*sigma+1)^2 gaussian every visible joint gets pasted."""
    tmp_size = sigma * 3
    size = 2 * tmp_size + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, np.newaxis]
    x0 = y0 = size // 2
    return np.exp(- ((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2)).astype(np.float32), int(tmp_size)


class TargetRenderer:
    """``render(joints, joints_vis, grid=None)`` -> (target [B,J,Hh,Wh], target_weight [B,J,1]) CUDA tensors.
    joints / joints_vis: [B,J,3] float64 (as the reference's ``meta`` holds them), in input-image pixels;
    grid: the GridMask table of ``advaug.pack_grid`` when the targets are for the masked view (its
    visibility rule, advaug.py:159-168, is applied first and the updated joints_vis is returned too)."""

    def __init__(self, image_size, heatmap_size, sigma=2, joints_weight=None, device='cuda'):
        self.W, self.H = int(image_size[0]), int(image_size[1])
        self.Wh, self.Hh = int(heatmap_size[0]), int(heatmap_size[1])
        g, self.tmp = gaussian_patch(sigma)
        self.device = torch.device(device)
        self.g = torch.from_numpy(g).to(self.device).contiguous()
        self.jw = None
        if joints_weight is not None:                      # LOSS.USE_DIFFERENT_JOINTS_WEIGHT (:488-489)
            self.jw = torch.as_tensor(np.asarray(joints_weight, dtype=np.float32).reshape(-1)).to(self.device)

    def render(self, joints, joints_vis, grid=None):
        j = torch.as_tensor(joints, dtype=torch.float64).to(self.device).contiguous()
        v = torch.as_tensor(joints_vis, dtype=torch.float64).to(self.device).contiguous()
        B, J, _ = j.shape
        target = torch.empty((B, J, self.Hh, self.Wh), device=self.device, dtype=torch.float32)
        tw = torch.empty((B, J, 1), device=self.device, dtype=torch.float32)
        vis_out = torch.empty_like(v) if grid is not None else None
        P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())      # noqa: E731
        call('advmix_render_targets', P(j), P(v), P(grid), P(self.g), self.tmp, P(self.jw), P(target), P(tw),
             P(vis_out), B, J, self.H, self.W, self.Hh, self.Wh,
             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return (target, tw) if grid is None else (target, tw, vis_out)
