"""Mirror of lib/core/evaluate.py:41-99 ``accuracy`` (PCK on heat-map argmax).  The argmax runs
on the GPU; only [B,J] indices are copied to the host instead of two full heat-map tensors."""
import numpy as np

from .inference import get_max_preds


def calc_dists(preds, target, normalize):
    preds = preds.astype(np.float32)
    target = target.astype(np.float32)
    valid = (target[:, :, 0] > 1) & (target[:, :, 1] > 1)                  # evaluate.py:21
    d = np.linalg.norm(preds / normalize[:, None, :] - target / normalize[:, None, :], axis=2)
    return np.where(valid, d, -1.0).T                                      # [J, B]


def dist_acc(dists, thr=0.5):
    cal = dists != -1
    n = cal.sum()
    return (dists[cal] < thr).sum() * 1.0 / n if n > 0 else -1


def accuracy(outputs, target, hm_type='gaussian', thr=0.5, args=None, cfg=None):
    if hm_type != 'gaussian' or args is not None:
        raise NotImplementedError('only the gaussian heat-map accuracy is on the hot path')
    pred, _ = get_max_preds(outputs)
    gt, _ = get_max_preds(target)
    h, w = outputs.shape[2], outputs.shape[3]
    norm = np.ones((pred.shape[0], 2)) * np.array([h, w]) / 10
    dists = calc_dists(pred, gt, norm)
    J = dists.shape[0]
    acc = np.zeros(J + 1)
    avg_acc, cnt = 0, 0
    for i in range(J):
        acc[i + 1] = dist_acc(dists[i], thr)
        if acc[i + 1] >= 0:
            avg_acc += acc[i + 1]
            cnt += 1
    avg_acc = avg_acc / cnt if cnt != 0 else 0
    if cnt != 0:
        acc[0] = avg_acc
    return acc, avg_acc, cnt, pred
