"""TEST INFRASTRUCTURE ONLY - CPU restatement of the three-view input pipeline.

  * ``to_tensor_normalize``  tools/train.py:116-126 - torchvision's ToTensor + Normalize.  torchvision is not
    installed here and not vendored in /root/reference (requirements.txt does not even list it), so its
    published behaviour is restated: HWC uint8 -> CHW float32 ``.div(255)``, then ``sub_(mean).div_(std)``
    with float32 mean / std.  Parity for this piece is therefore **unpinned against torchvision itself**.
  * ``grid_mask`` / ``grid_aug``  lib/dataset/advaug.py:111-170 with MixCombine's arguments (:189-202)
  * ``generate_target``      lib/dataset/JointsDataset.py:412-491 (gaussian branch)

Pinning: tests/golden/inputpipe.npz holds outputs of the REAL ``grid_aug`` and ``generate_target``
(oracle/gen_golden.py::gen_inputpipe) and tests/test_oracle_golden.py holds this file to them."""
import numpy as np
import torch

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def to_tensor_normalize(img_u8, mean=MEAN, std=STD):
    """img_u8: numpy uint8 [H,W,3] -> float32 tensor [3,H,W]."""
    t = torch.from_numpy(np.ascontiguousarray(img_u8.transpose(2, 0, 1))).to(torch.float32).div(255)
    m = torch.as_tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.as_tensor(std, dtype=torch.float32)[:, None, None]
    return t.sub_(m).div_(s)


def grid_draws(h, w, ratio, prob, rotate, rng):
    """advaug.py:112-140: the random numbers in call order; None = image left alone."""
    if rng.rand() > prob:
        return None
    d = rng.randint(2, min(h, w))
    l = rng.randint(1, d) if ratio == 1 else min(max(int(d * ratio + 0.5), 1), d - 1)
    st_h, st_w = rng.randint(d), rng.randint(d)
    r = rng.randint(rotate)
    assert r == 0
    return d, l, st_h, st_w


def grid_mask(h, w, d, l, st_h, st_w, mode=1):
    """advaug.py:116-149 with use_h = use_w = True, no rotation: float32 [h,w] of 0/1."""
    hh, ww = int(1.5 * h), int(1.5 * w)
    mask = np.ones((hh, ww), np.float32)
    for i in range(hh // d):
        s = d * i + st_h
        mask[s:min(s + l, hh), :] *= 0
    for i in range(ww // d):
        s = d * i + st_w
        mask[:, s:min(s + l, ww)] *= 0
    mask = np.asarray(np.uint8(mask))                      # PIL round trip, rotate(0)
    mask = mask[(hh - h) // 2:(hh - h) // 2 + h, (ww - w) // 2:(ww - w) // 2 + w].astype(np.float32)
    return 1 - mask if mode == 1 else mask


def grid_aug(img, joints, joints_vis, draws, num_joints):
    """advaug.py:150-170: apply the mask to a [3,h,w] tensor and hide the joints it covers."""
    if draws is None:
        return img, joints_vis, None
    h, w = img.shape[1], img.shape[2]
    mask = grid_mask(h, w, *draws)
    out = img * torch.from_numpy(mask).expand_as(img)
    vis = joints_vis.copy()
    for j in range(num_joints):
        tx = max(min(int(joints[j][0]), w - 1), 0)
        ty = max(min(int(joints[j][1]), h - 1), 0)
        if mask[ty, tx] == 0:
            vis[j][0] = 0
            vis[j][1] = 0
    return out, vis, mask


def generate_target(joints, joints_vis, image_size, heatmap_size, sigma, joints_weight=None):
    """JointsDataset.py:412-491.  image_size / heatmap_size: (w, h).  Returns (target [J,Hh,Wh] f32,
    target_weight [J,1] f32)."""
    J = joints.shape[0]
    image_size, heatmap_size = np.array(image_size), np.array(heatmap_size)
    tw = np.ones((J, 1), dtype=np.float32)
    tw[:, 0] = joints_vis[:, 0]
    target = np.zeros((J, heatmap_size[1], heatmap_size[0]), dtype=np.float32)
    tmp = sigma * 3
    for j in range(J):
        fs = image_size / heatmap_size
        mu_x = int(joints[j][0] / fs[0] + 0.5)
        mu_y = int(joints[j][1] / fs[1] + 0.5)
        ul = [int(mu_x - tmp), int(mu_y - tmp)]
        br = [int(mu_x + tmp + 1), int(mu_y + tmp + 1)]
        if ul[0] >= heatmap_size[0] or ul[1] >= heatmap_size[1] or br[0] < 0 or br[1] < 0:
            tw[j] = 0
            continue
        size = 2 * tmp + 1
        x = np.arange(0, size, 1, np.float32)
        y = x[:, np.newaxis]
        x0 = y0 = size // 2
        g = np.exp(- ((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
        g_x = max(0, -ul[0]), min(br[0], heatmap_size[0]) - ul[0]
        g_y = max(0, -ul[1]), min(br[1], heatmap_size[1]) - ul[1]
        i_x = max(0, ul[0]), min(br[0], heatmap_size[0])
        i_y = max(0, ul[1]), min(br[1], heatmap_size[1])
        if tw[j] > 0.5:
            target[j][i_y[0]:i_y[1], i_x[0]:i_x[1]] = g[g_y[0]:g_y[1], g_x[0]:g_x[1]]
    if joints_weight is not None:
        tw = np.multiply(tw, joints_weight)
    return target, tw


def synth_samples(tag, B, J, H, W):
    """Philox-keyed uint8 crops, float64 joints (some outside the image) and 0/1 visibilities."""
    from . import detinit
    base = (detinit.uniform(tag + '.base', (B, H, W, 3)).numpy() * 256).astype(np.uint8)
    aug = (detinit.uniform(tag + '.aug', (B, H, W, 3)).numpy() * 256).astype(np.uint8)
    jt = np.zeros((B, J, 3), dtype=np.float64)
    jt[:, :, 0] = detinit.uniform(tag + '.jx', (B, J)).numpy().astype(np.float64) * (W + 40) - 20
    jt[:, :, 1] = detinit.uniform(tag + '.jy', (B, J)).numpy().astype(np.float64) * (H + 40) - 20
    vis = np.zeros((B, J, 3), dtype=np.float64)
    v = (detinit.uniform(tag + '.vis', (B, J)).numpy() < 0.8).astype(np.float64)
    vis[:, :, 0] = v
    vis[:, :, 1] = v
    return base, aug, jt, vis
