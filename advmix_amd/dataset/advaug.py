"""Device side of lib/dataset/advaug.py's GridMask view and of the ToTensor + Normalize transform.

The reference builds its three views per sample in the DataLoader workers (MixCombine, advaug.py:173-207):
``transform(base)`` three times plus a [1.5h, 1.5w] numpy mask for GridMask.  Here the worker only draws
GridMask's random numbers (same numpy RNG calls in the same order, ``grid_params``) and ships the uint8
crops; ``make_views`` turns a batch of them into the three normalised NCHW float views in one launch."""
import ctypes

import numpy as np
import torch

from .._lib import call

IMAGENET_MEAN = (0.485, 0.456, 0.406)        # tools/train.py:116-118
IMAGENET_STD = (0.229, 0.224, 0.225)


def grid_params(h, w, ratio=0.5, prob=0.7, rotate=1, rng=np.random):
    """The random draws of ``grid_aug`` (advaug.py:112-140) in its order: keep-probability, period d,
    stripe offsets, rotation.  Returns (d, l, st_h, st_w), or None when the sample keeps its image.
    Only ``rotate == 1`` (no rotation - the only value MixCombine passes, advaug.py:192) is supported."""
    if rotate != 1:
        raise NotImplementedError('GridMask rotation is not on the hot path (MixCombine passes rotate=1)')
    if rng.rand() > prob:
        return None
    d = rng.randint(2, min(h, w))
    if ratio == 1:
        l = rng.randint(1, d)
    else:
        l = min(max(int(d * ratio + 0.5), 1), d - 1)
    st_h = rng.randint(d)
    st_w = rng.randint(d)
    rng.randint(rotate)                                    # r (always 0), drawn to keep the stream aligned
    return int(d), int(l), int(st_h), int(st_w)


def pack_grid(params, device):
    """int32 [B,4] device table for a batch of ``grid_params`` results (None -> d = 0 = no mask)."""
    t = np.zeros((len(params), 4), dtype=np.int32)
    for i, p in enumerate(params):
        if p is not None:
            t[i] = p
    return torch.from_numpy(t).to(device)


def make_views(base_u8, aug_u8=None, grid=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """base_u8 / aug_u8: uint8 CUDA tensors [B,H,W,3] (the warped crop and its AutoAugment version);
    grid: int32 [B,4] from ``pack_grid`` or None.  Returns [clean, autoaug, gridmask] float32 NCHW views,
    i.e. ``Normalize(ToTensor(.))`` of each, the third multiplied by GridMask's mask (mode = 1)."""
    if not base_u8.is_cuda or base_u8.dtype != torch.uint8 or base_u8.dim() != 4 or base_u8.shape[3] != 3:
        raise TypeError('make_views needs a uint8 CUDA tensor [B,H,W,3]')
    base_u8 = base_u8.contiguous()
    B, H, W, _ = base_u8.shape
    if aug_u8 is not None:
        if aug_u8.shape != base_u8.shape or aug_u8.dtype != torch.uint8 or not aug_u8.is_cuda:
            raise TypeError('aug_u8 must match base_u8')
        aug_u8 = aug_u8.contiguous()
    views = [torch.empty((B, 3, H, W), device=base_u8.device, dtype=torch.float32) for _ in range(3)]
    m = (ctypes.c_float * 3)(*np.asarray(mean, dtype=np.float32))
    s = (ctypes.c_float * 3)(*np.asarray(std, dtype=np.float32))
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())      # noqa: E731
    call('advmix_make_views', P(base_u8), P(aug_u8), P(grid), ctypes.cast(m, ctypes.c_void_p),
         ctypes.cast(s, ctypes.c_void_p), P(views[0]), P(views[1]), P(views[2]), B, H, W,
         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return views
