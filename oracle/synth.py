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


def synth_heatmaps(tag, B, J, H, W):
    """Noise + one gaussian bump per joint; joint 0 of every sample is all-negative (pred_mask branch),
    joint 1 peaks on the border (no quarter-pixel shift).  Philox-keyed: tests rebuild it, nothing stored."""
    hm = detinit.normal(tag + '.noise', (B, J, H, W), 0.05).numpy()
    cx = (detinit.uniform(tag + '.cx', (B, J)).numpy() * (W - 1)).round().astype(int)
    cy = (detinit.uniform(tag + '.cy', (B, J)).numpy() * (H - 1)).round().astype(int)
    cx[:, 1] = 0
    yy, xx = np.mgrid[0:H, 0:W]
    for b in range(B):
        for j in range(J):
            hm[b, j] += np.exp(-((xx - cx[b, j]) ** 2 + (yy - cy[b, j]) ** 2) / 8.0).astype(np.float32)
        hm[b, 0] = -np.abs(hm[b, 0]) - 0.01
    return np.ascontiguousarray(hm, dtype=np.float32)


def synth_boxes(tag, B):
    """float32 center [B,2] / scale [B,2] like coco.py::_box2cs (scale in units of 200 px, aspect 0.75)."""
    c = (detinit.uniform(tag + '.c', (B, 2)).numpy() * np.array([600, 440]) + 20).astype(np.float32)
    sw = (detinit.uniform(tag + '.s', (B,)).numpy() * 2.5 + 0.4).astype(np.float32)
    s = np.stack([sw, sw / np.float32(0.75)], 1).astype(np.float32)
    score = (detinit.uniform(tag + '.score', (B,)).numpy() * 0.9 + 0.1).astype(np.float32)
    return c, s, score
