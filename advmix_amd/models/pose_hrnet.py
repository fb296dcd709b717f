"""HRNet pose network on the HIP plan executor.  Mirrors the interface of
lib/models/pose_hrnet.py: ``get_pose_net(cfg, is_train, **kw)`` -> nn.Module with
``forward([B,3,H,W]) -> [B,J,H/4,W/4]``, ``init_weights(pretrained)`` and the reference's
state-dict keys/shapes (e.g. ``stage3.2.fuse_layers.1.0.0.0.weight``)."""
import logging

from ..plan import PlanNet, hrnet_plan
from ._init_utils import normal_init_, load_pretrained, check_pretrained

logger = logging.getLogger(__name__)


class PoseHighResolutionNet(PlanNet):
    def __init__(self, cfg, **kwargs):
        extra = cfg['MODEL']['EXTRA']
        super().__init__(hrnet_plan(extra, cfg['MODEL']['NUM_JOINTS']))
        self.pretrained_layers = list(extra.get('PRETRAINED_LAYERS', ['*']))

    def init_weights(self, pretrained=''):
        """pose_hrnet.py:462-492: N(0, 1e-3) conv/deconv, zero conv biases, BN (1, 0), then an
        optional ImageNet checkpoint filtered by PRETRAINED_LAYERS."""
        logger.info('=> init weights from normal distribution')
        normal_init_(self, 0.001, zero_conv_bias=True)
        if check_pretrained(pretrained):
            keep = self.pretrained_layers
            load_pretrained(self, pretrained,
                            lambda k: keep[0] == '*' or k.split('.')[0] in keep)


def get_pose_net(cfg, is_train, **kwargs):
    model = PoseHighResolutionNet(cfg, **kwargs)
    if is_train and cfg['MODEL']['INIT_WEIGHTS']:
        model.init_weights(cfg['MODEL']['PRETRAINED'])
    return model
