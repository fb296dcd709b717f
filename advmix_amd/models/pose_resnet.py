"""SimpleBaseline (ResNet + 3 deconvs) on the HIP plan executor.  Mirrors
lib/models/pose_resnet.py: ``get_pose_net(cfg, is_train, **kw)``; state-dict keys such as
``deconv_layers.0.weight`` ([Cin, Cout, 4, 4]) and ``final_layer.{weight,bias}``."""
import logging

from ..plan import PlanNet, resnet_plan
from ._init_utils import normal_init_, load_pretrained, check_pretrained

logger = logging.getLogger(__name__)


def _extra(cfg):
    e = cfg['MODEL']['EXTRA']
    return e


class PoseResNet(PlanNet):
    def __init__(self, cfg, **kwargs):
        extra = _extra(cfg)
        super().__init__(resnet_plan(extra, cfg['MODEL']['NUM_JOINTS']))
        self.deconv_with_bias = bool(extra.get('DECONV_WITH_BIAS', False))

    def init_weights(self, pretrained=''):
        """pose_resnet.py:209-249.  Without a checkpoint: every conv/deconv N(0, 1e-3), BN (1, 0);
        conv biases keep their default init (the reference leaves ``nn.init.constant_(m.bias, 0)``
        commented out at :241) but deconv biases are zeroed."""
        logger.info('=> init weights from normal distribution')
        normal_init_(self, 0.001, zero_conv_bias=False)
        if self.deconv_with_bias:
            for name, _, kind in self.plan.params:
                if name.startswith('deconv_layers') and kind.startswith('bias'):
                    self.get_parameter(name).data.zero_()
        if check_pretrained(pretrained):
            self.get_parameter('final_layer.bias').data.zero_()
            load_pretrained(self, pretrained)


def get_pose_net(cfg, is_train, **kwargs):
    model = PoseResNet(cfg, **kwargs)
    if is_train and cfg['MODEL']['INIT_WEIGHTS']:
        model.init_weights(cfg['MODEL']['PRETRAINED'])
    return model
