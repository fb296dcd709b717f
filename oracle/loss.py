"""Joints loss + in-loop accuracy (oracle; test infra only).

``joints_loss`` follows lib/core/loss.py:15-65.  NOTE the inverted flag there:
``smooth_L1=False`` (the only way it is constructed, tools/train.py:111)
selects nn.SmoothL1Loss (beta=1, mean); ``smooth_L1=True`` selects nn.MSELoss.
``accuracy`` follows lib/core/evaluate.py:41-99 + lib/core/inference.py:22-49.
"""
import numpy as np
import torch
import torch.nn.functional as F


def joints_loss(output, target, target_weight, use_target_weight=True, smooth_L1=False):
    B, J = output.shape[0], output.shape[1]
    p = output.float().reshape(B, J, -1)
    g = target.float().reshape(B, J, -1)
    if use_target_weight:
        w = target_weight.float().reshape(B, J, 1)
        p, g = p * w, g * w
    tot = 0
    for j in range(J):                                    # loss.py:46-63
        if smooth_L1:
            tot = tot + 0.5 * F.mse_loss(p[:, j], g[:, j])
        else:
            tot = tot + 0.5 * F.smooth_l1_loss(p[:, j], g[:, j])
    return tot / J


def max_preds(hm):
    """inference.py:22-49 on a numpy [B,J,H,W] array -> (preds[B,J,2] float32, maxvals)."""
    B, J, H, W = hm.shape
    flat = hm.reshape(B, J, -1)
    idx = flat.argmax(2)
    mv = flat.max(2)
    preds = np.stack([idx % W, idx // W], -1).astype(np.float32)
    preds *= (mv > 0.0)[..., None].astype(np.float32)
    return preds, mv[..., None]


def accuracy(output, target, thr=0.5):
    """evaluate.py:41-99 (hm_type gaussian, args None). Returns (acc[J+1], avg, cnt, pred).  ``thr`` is accepted and, as in
    the reference, NOT used: evaluate.py:90 calls ``dist_acc(dists[idx[i]])`` with dist_acc's own default of 0.5."""
    thr = 0.5
    out = output.detach().cpu().numpy()
    tgt = target.detach().cpu().numpy()
    pred, _ = max_preds(out)
    gt, _ = max_preds(tgt)
    h, w = out.shape[2], out.shape[3]
    norm = np.ones((pred.shape[0], 2)) * np.array([h, w]) / 10
    B, J = pred.shape[:2]
    dists = np.zeros((J, B))
    for n in range(B):                                    # evaluate.py:15-27
        for c in range(J):
            if gt[n, c, 0] > 1 and gt[n, c, 1] > 1:
                dists[c, n] = np.linalg.norm(pred[n, c] / norm[n] - gt[n, c] / norm[n])
            else:
                dists[c, n] = -1
    acc = np.zeros(J + 1)
    avg, cnt = 0.0, 0
    for j in range(J):
        valid = dists[j] != -1
        acc[j + 1] = (dists[j][valid] < thr).sum() / valid.sum() if valid.sum() > 0 else -1
        if acc[j + 1] >= 0:
            avg += acc[j + 1]
            cnt += 1
    avg = avg / cnt if cnt else 0
    if cnt:
        acc[0] = avg
    return acc, avg, cnt, pred
