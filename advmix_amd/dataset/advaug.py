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


# ---- the AutoAugment view (ImageNetPolicy, advaug.py:10-108) ---------------------------------------------------------
# The reference applies PIL operations in the DataLoader workers (MixCombine, advaug.py:180-187): at ~540 images/s per
# GPU that is ~4,300 PIL AutoAugment calls a second for an 8-GPU node.  Here the worker only DRAWS (Python's ``random``,
# the same calls in the same order: ``autoaug_params``) and the operations run on the device, bit-identical to Pillow.
# The policy table only ever reaches these five operations.
AA_EQUALIZE, AA_POSTERIZE, AA_SOLARIZE, AA_INVERT, AA_SHARPNESS = 1, 2, 3, 4, 5
_AA_CODE = {'equalize': AA_EQUALIZE, 'posterize': AA_POSTERIZE, 'solarize': AA_SOLARIZE, 'invert': AA_INVERT,
            'sharpness': AA_SHARPNESS}
# advaug.py:22-35: (p1, operation1, magnitude_idx1, p2, operation2, magnitude_idx2)
IMAGENET_POLICIES = (
    (0.8, 'equalize', 8, 0.6, 'equalize', 3), (0.6, 'posterize', 7, 0.6, 'posterize', 6),
    (0.4, 'equalize', 7, 0.2, 'solarize', 4), (0.6, 'solarize', 3, 0.6, 'equalize', 7),
    (0.8, 'posterize', 5, 1.0, 'equalize', 2), (0.6, 'equalize', 8, 0.4, 'posterize', 6),
    (0.0, 'equalize', 7, 0.8, 'equalize', 8), (0.6, 'invert', 4, 1.0, 'equalize', 8),
    (0.4, 'sharpness', 7, 0.6, 'invert', 8), (0.4, 'equalize', 7, 0.2, 'solarize', 4),
    (0.6, 'invert', 4, 1.0, 'equalize', 8), (0.8, 'equalize', 8, 0.6, 'equalize', 3),
)
# advaug.py:50-65, the ``ranges`` rows of the reachable operations
_AA_RANGES = {'posterize': np.round(np.linspace(8, 4, 10), 0).astype(int), 'solarize': np.linspace(256, 0, 10),
              'sharpness': np.linspace(0.0, 0.9, 10)}


def autoaug_params(rng=None):
    """The random draws of ``ImageNetPolicy.__call__`` + ``SubPolicy.__call__`` (advaug.py:38-40, 102-105) in their
    order - ``randint(0, 11)``; ``random()`` against p1; ``choice([-1, 1])`` if the first operation is sharpness and
    fires; ``random()`` against p2; ``choice`` likewise - on ``rng`` (default: Python's ``random`` module, which is what
    the reference uses).  Returns the (code, parameter) pairs that fire, in order (zero, one or two)."""
    import random as _random
    rng = _random if rng is None else rng
    p1, op1, m1, p2, op2, m2 = IMAGENET_POLICIES[rng.randint(0, len(IMAGENET_POLICIES) - 1)]
    out = []
    for p, op, m in ((p1, op1, m1), (p2, op2, m2)):
        if rng.random() < p:
            mag = _AA_RANGES[op][m] if op in _AA_RANGES else 0
            if op == 'sharpness':
                mag = 1 + mag * rng.choice([-1, 1])
            out.append((_AA_CODE[op], float(mag)))
    return out


def pack_autoaug(params, device):
    """int32 [B,4] device table {code1, param1, code2, param2} for a batch of ``autoaug_params`` results: posterize's
    parameter becomes its bit mask (ImageOps.posterize), solarize's threshold and sharpness's factor travel as float32
    bits (Pillow converts both to a C float)."""
    t = np.zeros((len(params), 4), dtype=np.int32)
    for i, ops in enumerate(params):
        if len(ops) > 2:
            raise ValueError('a sub-policy applies at most two operations')
        for k, (code, par) in enumerate(ops):
            t[i, 2 * k] = code
            if code == AA_POSTERIZE:
                t[i, 2 * k + 1] = ~(2 ** (8 - int(par)) - 1)
            elif code == AA_SOLARIZE:
                # ``i < threshold`` is evaluated in Python (double) by ImageOps.solarize for integer i: any float32 between
                # the same two integers decides identically, so the float32 nearest the threshold is exact
                t[i, 2 * k + 1] = np.float32(par).view(np.int32)
            elif code == AA_SHARPNESS:
                t[i, 2 * k + 1] = np.float32(par).view(np.int32)
    return torch.from_numpy(t).to(device)


def auto_augment(base_u8, ops):
    """The AutoAugment view of a batch of uint8 crops [B,H,W,3] (CUDA) for the packed draws ``ops`` (``pack_autoaug``):
    what ``np.array(ImageNetPolicy()(Image.fromarray(crop)))`` returns for the same draws, computed on the device."""
    if not base_u8.is_cuda or base_u8.dtype != torch.uint8 or base_u8.dim() != 4 or base_u8.shape[3] != 3:
        raise TypeError('auto_augment needs a uint8 CUDA tensor [B,H,W,3]')
    base_u8 = base_u8.contiguous()
    B, H, W, _ = base_u8.shape
    if ops.shape != (B, 4) or ops.dtype != torch.int32 or not ops.is_cuda:
        raise TypeError('ops must be the int32 [B,4] table of pack_autoaug')
    out, tmp = torch.empty_like(base_u8), torch.empty_like(base_u8)
    P = lambda t: ctypes.c_void_p(t.data_ptr())              # noqa: E731
    call('advmix_autoaug', P(base_u8), P(ops.contiguous()), P(tmp), P(out), B, H, W,
         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return out
