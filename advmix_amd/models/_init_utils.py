import os
import logging
import torch

logger = logging.getLogger(__name__)


def normal_init_(net, std=0.001, zero_conv_bias=True):
    """conv / deconv weights ~ N(0, std), BN (1, 0)  (pose_hrnet.py:464-479, pose_resnet.py:238-249)."""
    with torch.no_grad():
        for name, shape, kind in net.plan.params:
            p = net.get_parameter(name)
            if kind in ('conv', 'deconv'):
                p.copy_(torch.empty(shape).normal_(0, std).to(p.device))
            elif kind.startswith('bias') and zero_conv_bias:
                p.zero_()
            elif kind == 'bn_w':
                p.fill_(1)
            elif kind == 'bn_b':
                p.zero_()


def load_pretrained(net, path, layer_filter=None):
    sd = torch.load(path, map_location='cpu')
    if layer_filter is not None:
        sd = {k: v for k, v in sd.items() if layer_filter(k)}
    logger.info('=> loading pretrained model %s', path)
    net.load_state_dict(sd, strict=False)


def check_pretrained(path):
    if path and not os.path.isfile(path):
        logger.error('=> please download pre-trained models first!')
        raise ValueError('{} is not exist!'.format(path))
    return bool(path)
