"""The authored experiment YAMLs load through the yacs-compatible config and build the models
the benchmark configs name (C3 HRNet-W32 256x192, C4 HRNet-W48 384x288)."""
import os
import types

from advmix_amd.config import cfg, update_config
from advmix_amd import models

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(rel):
    c = cfg.clone()
    update_config(c, types.SimpleNamespace(cfg=os.path.join(ROOT, rel), opts=['MODEL.PRETRAINED', '']))
    return c


def test_w32_coco_yaml():
    c = _load('experiments/coco/hrnet/w32_256x192_adam_lr1e-3_advmix.yaml')
    assert c.GPUS == tuple(range(8)) and c.MODEL.IMAGE_SIZE == [192, 256] and c.TRAIN.BATCH_SIZE_PER_GPU == 32
    m = models.pose_hrnet.get_pose_net(c, is_train=True)
    assert sum(p.numel() for p in m.parameters()) == 28536113


def test_w48_coco_yaml():
    c = _load('experiments/coco/hrnet/w48_384x288_adam_lr1e-3_advmix.yaml')
    assert c.MODEL.EXTRA.STAGE4.NUM_CHANNELS == [48, 96, 192, 384]
    m = models.pose_hrnet.get_pose_net(c, is_train=True)
    assert sum(p.numel() for p in m.parameters()) == 63595745
