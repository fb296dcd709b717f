"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's validation path.

numpy / plain-PyTorch restatement of
  * ``flip_back``                      lib/utils/transforms.py:16-41 (heat-map branch)
  * the flip-test merge                lib/core/function.py:240-261
  * ``get_max_preds``                  lib/core/inference.py:22-49
  * ``get_final_preds``                lib/core/inference.py:52-95 (heat-map branch)
  * ``transform_preds`` & friends      lib/utils/transforms.py:57-120
  * one ``validate`` batch             lib/core/function.py:223-300
  * rescoring + OKS-NMS per image      lib/dataset/coco.py:318-371

Pinning: tests/golden/validate.npz is produced by oracle/gen_golden.py from the REAL reference
(``validate`` loop, ``get_final_preds``, ``flip_back``, ``COCODataset.evaluate``'s rescoring loop via
``oks_nms``) and tests/test_oracle_golden.py holds this file to it.  One dependency of that path is
absent from the container and from /root/reference: **OpenCV** (``cv2.getAffineTransform``, imported at
transforms.py:12; requirements.txt does not pin a version).  Its published algorithm - build the 6x6
linear system of the three point pairs in double precision and solve it by LU decomposition
(``cv::solve`` default) - is restated in ``cv_get_affine_transform`` below and injected into the stubbed
``cv2`` module when the fixtures are generated, so for that one function parity is **unpinned against
OpenCV itself**; everything around it (point construction, float32 roundings, the double-precision
multiply, the store into float32) is the reference's own code in the fixtures.
"""
import math

import numpy as np
import torch

from . import nms as onms
from .loss import joints_loss, accuracy
from .posenet import posenet_forward


# ---- OpenCV stand-in ------------------------------------------------------------------------------------------

def cv_get_affine_transform(src, dst):
    """cv2.getAffineTransform(src[3,2] f32, dst[3,2] f32) -> 2x3 float64 mapping src -> dst.
    OpenCV (imgproc, imgwarp.cpp ``getAffineTransform``): rows 2i / 2i+1 of a 6x6 double system hold
    (x, y, 1, 0, 0, 0) and (0, 0, 0, x, y, 1) of src[i], the right-hand side dst[i]; solved by LU."""
    src = np.asarray(src, dtype=np.float32).astype(np.float64)
    dst = np.asarray(dst, dtype=np.float32).astype(np.float64)
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[2 * i, 0:3] = (src[i, 0], src[i, 1], 1.0)
        A[2 * i + 1, 3:6] = (src[i, 0], src[i, 1], 1.0)
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(A, b).reshape(2, 3)


# ---- transforms.py --------------------------------------------------------------------------------------------

def third_point(a, b):
    """transforms.py:107-109: b + rot90(a - b), float32."""
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def affine_from_box(center, scale, rot, output_size, inv):
    """transforms.py:64-98 (shift = 0).  center, scale: float32 [2]; output_size (w, h)."""
    scale = np.asarray(scale)
    scale_px = scale * 200.0                                            # :73
    src_w = scale_px[0]
    dst_w, dst_h = output_size[0], output_size[1]
    ang = np.pi * rot / 180
    sn, cs = np.sin(ang), np.cos(ang)                                   # get_dir, :112-120
    p = [0, src_w * -0.5]
    src_dir = [p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs]
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center                                                  # + scale_px * shift, shift = 0
    src[1, :] = center + src_dir
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2, :] = third_point(src[0, :], src[1, :])
    dst[2, :] = third_point(dst[0, :], dst[1, :])
    return cv_get_affine_transform(dst, src) if inv else cv_get_affine_transform(src, dst)


def transform_preds(coords, center, scale, output_size):
    """transforms.py:57-62: heat-map coordinates -> image coordinates (float64 result)."""
    out = np.zeros(coords.shape)
    t = affine_from_box(center, scale, 0, output_size, inv=1)
    for p in range(coords.shape[0]):
        out[p, 0:2] = np.dot(t, np.array([coords[p, 0], coords[p, 1], 1.]).T)[:2]     # affine_transform :101-104
    return out


def flip_back(maps, pairs):
    """transforms.py:16-41, 4-D branch: reverse W, then swap each (left, right) joint pair in turn."""
    out = np.ascontiguousarray(maps[..., ::-1]).copy()
    for a, b in pairs:
        keep = out[:, a].copy()
        out[:, a] = out[:, b]
        out[:, b] = keep
    return out


# ---- inference.py ---------------------------------------------------------------------------------------------

def get_max_preds(hm):
    """inference.py:22-49."""
    B, J, H, W = hm.shape
    flat = hm.reshape(B, J, -1)
    idx = np.argmax(flat, 2).reshape(B, J, 1)
    maxvals = np.amax(flat, 2).reshape(B, J, 1)
    preds = np.tile(idx, (1, 1, 2)).astype(np.float32)
    preds[:, :, 0] = preds[:, :, 0] % W
    preds[:, :, 1] = np.floor(preds[:, :, 1] / W)
    preds *= np.tile(np.greater(maxvals, 0.0), (1, 1, 2)).astype(np.float32)
    return preds, maxvals


def get_final_preds(hm, center, scale, post_process):
    """inference.py:52-95 with cal_hm_coord=True, coord=None.  Returns (preds f32 [B,J,2], maxvals [B,J,1],
    heat-map-space coords f32 [B,J,2])."""
    coords, maxvals = get_max_preds(hm)
    H, W = hm.shape[2], hm.shape[3]
    if post_process:                                                    # :64-76
        for n in range(coords.shape[0]):
            for p in range(coords.shape[1]):
                m = hm[n][p]
                px = int(math.floor(coords[n][p][0] + 0.5))
                py = int(math.floor(coords[n][p][1] + 0.5))
                if 1 < px < W - 1 and 1 < py < H - 1:
                    diff = np.array([m[py][px + 1] - m[py][px - 1], m[py + 1][px] - m[py - 1][px]])
                    coords[n][p] += np.sign(diff) * .25
    preds = coords.copy()
    for i in range(coords.shape[0]):                                    # :80-84
        preds[i] = transform_preds(coords[i], center[i], scale[i], [W, H])
    return preds, maxvals, coords


# ---- function.py::validate, one batch -------------------------------------------------------------------------

def flip_test_merge(output, output_flipped, pairs, shift):
    """function.py:249-261 on numpy arrays."""
    f = flip_back(output_flipped, pairs)
    if shift:
        f[:, :, :, 1:] = f.copy()[:, :, :, 0:-1]
    return (output + f) * np.float32(0.5)


def validate_batch(net, extra, P, x, target, tw, pairs, flip_test, shift, use_target_weight=True):
    """function.py:223-276: eval forward (+ flip test), loss, accuracy.  Returns (output np [B,J,H,W],
    loss float, avg_acc, cnt)."""
    with torch.no_grad():
        out = posenet_forward(net, P, x, extra, False)
        if flip_test:
            of = posenet_forward(net, P, x.flip(3), extra, False)
            out = torch.from_numpy(flip_test_merge(out.numpy(), of.numpy(), pairs, shift))
        loss = joints_loss(out, target, tw, use_target_weight)
    _, avg, cnt, _ = accuracy(out, target)
    return out.numpy(), float(loss), avg, cnt


def collect(all_out, centers, scales, scores, post_process):
    """function.py:283-297: the all_preds / all_boxes rows of the given batches."""
    preds, maxvals, _ = get_final_preds(all_out, centers, scales, post_process)
    N, J = preds.shape[:2]
    all_preds = np.zeros((N, J, 3), dtype=np.float32)
    all_boxes = np.zeros((N, 6))
    all_preds[:, :, 0:2] = preds[:, :, 0:2]
    all_preds[:, :, 2:3] = maxvals
    all_boxes[:, 0:2] = centers[:, 0:2]
    all_boxes[:, 2:4] = scales[:, 0:2]
    all_boxes[:, 4] = np.prod(scales * 200, 1)
    all_boxes[:, 5] = scores
    return all_preds, all_boxes


# ---- coco.py::evaluate, rescoring + OKS-NMS -------------------------------------------------------------------

def rescore_and_nms(all_preds, all_boxes, image_ids, in_vis_thre, oks_thre, soft):
    """coco.py:318-371.  Returns [(image_id, [(row index into all_preds, new score), ...kept...]), ...]."""
    groups = {}
    for n, img in enumerate(image_ids):
        groups.setdefault(img, []).append(n)
    out = []
    for img, rows in groups.items():
        db = []
        for n in rows:
            acc, cnt = 0, 0
            for j in range(all_preds.shape[1]):                         # :343-350, float32 running sum
                t = all_preds[n][j][2]
                if t > in_vis_thre:
                    acc = acc + t
                    cnt = cnt + 1
            if cnt != 0:
                acc = acc / cnt
            db.append({'keypoints': all_preds[n], 'area': all_boxes[n][4], 'score': acc * all_boxes[n][5]})
        keep = (onms.soft_oks_nms if soft else onms.oks_nms)(db, oks_thre)
        keep = list(range(len(db))) if len(keep) == 0 else [int(k) for k in keep]
        out.append((img, [(rows[k], float(db[k]['score'])) for k in keep]))
    return out
