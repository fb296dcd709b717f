"""Seeded synthetic batches shared by gen_golden.py, the tests and bench.py's
cpu_baseline leg (oracle side; test infra only).  SURVEY.md §8 d2."""
import zlib
import numpy as np
import torch

from . import detinit


def synth_batch(tag, B, J, H, W):
    """3 N(0,1) views [B,3,H,W]; Gaussian sigma=2 heatmaps [B,J,H/4,W/4] (13x13
    support as JointsDataset.py:468-486); target_weight [B,J,1] in {0,1}, P(1)=0.8."""
    views = [detinit.normal('%s.view%d' % (tag, k), (B, 3, H, W)) for k in range(3)]
    hh, ww = H // 4, W // 4
    rng = np.random.Generator(np.random.Philox(key=zlib.crc32(tag.encode())))
    cx = rng.integers(0, ww, (B, J))
    cy = rng.integers(0, hh, (B, J))
    ys, xs = np.mgrid[0:hh, 0:ww]
    tgt = np.exp(-((xs[None, None] - cx[..., None, None]) ** 2 +
                   (ys[None, None] - cy[..., None, None]) ** 2) / (2 * 2.0 ** 2)).astype(np.float32)
    tgt[tgt < np.exp(-4.5)] = 0
    tw = (rng.random((B, J, 1)) < 0.8).astype(np.float32)
    return views, torch.from_numpy(tgt), torch.from_numpy(tw)


def strided(t, n=4096):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].cpu().numpy().copy()


def checksum(sd, names):
    return {k: [float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in names}
