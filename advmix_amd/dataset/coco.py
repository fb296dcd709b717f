"""The part of lib/dataset/coco.py::COCODataset.evaluate that sits between ``validate`` and the
result file (coco.py:318-371): group predictions per image, rescore each person by the mean of its
confident joint scores, then OKS-NMS per image.  This is ``lib/nms``'s live call site
(SURVEY.md 8 f1); annotation loading, json writing and pycocotools scoring stay outside the path.

``COCO_FLIP_PAIRS`` is coco.py:71-72 (data, needed by the flip test)."""
from collections import defaultdict

import numpy as np

from ..nms.nms import oks_nms, soft_oks_nms

COCO_FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]


def image_index(path):
    """coco.py:321: the 12-digit COCO id in front of the extension."""
    return int(path[-16:-4])


def rescore(preds, box_scores, in_vis_thre):
    """coco.py:340-353 for all persons at once: box score x mean of the joint scores above
    ``in_vis_thre``.  The joint scores are float32 and the reference adds them one by one in joint
    order, so the sum is accumulated joint-by-joint (vectorised over persons, same rounding)."""
    conf = np.asarray(preds)[:, :, 2]
    total = np.zeros(conf.shape[0], dtype=conf.dtype)
    count = np.zeros(conf.shape[0], dtype=conf.dtype)
    for j in range(conf.shape[1]):
        sel = conf[:, j] > in_vis_thre
        total = np.where(sel, total + conf[:, j], total)
        count = count + sel.astype(conf.dtype)
    mean = np.where(count != 0, total / np.maximum(count, 1), total)
    return mean * np.asarray(box_scores)


def rescore_and_nms(preds, all_boxes, img_path, num_joints, in_vis_thre, oks_thre, soft_nms=False):
    """coco.py:318-371.  preds [N,J,3] (x, y, maxval), all_boxes [N,6] (center, scale, area, score),
    img_path: N image paths.  Returns, per image in first-seen order, the kept person dicts
    (keys keypoints/center/scale/area/score/image like the reference's) after OKS-NMS."""
    preds = np.asarray(preds)[:, :num_joints]
    all_boxes = np.asarray(all_boxes)
    scores = rescore(preds, all_boxes[:, 5], in_vis_thre)
    per_image = defaultdict(list)
    for n, path in enumerate(img_path):
        img = image_index(path)
        per_image[img].append({'keypoints': preds[n], 'center': all_boxes[n, 0:2], 'scale': all_boxes[n, 2:4],
                               'area': all_boxes[n, 4], 'score': scores[n], 'image': img})
    suppress = soft_oks_nms if soft_nms else oks_nms
    out = []
    for persons in per_image.values():
        keep = suppress(persons, oks_thre)
        out.append(persons if len(keep) == 0 else [persons[k] for k in keep])
    return out
