"""Model EXTRA dicts used by the oracle, fixtures and tests (data only).

W32 values: experiments/mpii/hrnet/w32_256x256_adam_lr1e-3_advmix.yaml:52-90.
W48 = upstream HRNet widths 48/96/192/384 (not a reference YAML).
RES50: experiments/coco/resnet/res50_256x192_d256x3_adam_lr1e-3_advmix.yaml:31-44.
TINY_*: shrunken widths for second-scale CPU tests (same topology).
"""


def hrnet_extra(widths, modules=(1, 4, 3), blocks=4):
    ex = {'FINAL_CONV_KERNEL': 1, 'PRETRAINED_LAYERS': ['*']}
    for i, st in enumerate((2, 3, 4)):
        nb = st
        ex['STAGE%d' % st] = {'NUM_MODULES': modules[i], 'NUM_BRANCHES': nb, 'BLOCK': 'BASIC',
                              'NUM_BLOCKS': [blocks] * nb, 'NUM_CHANNELS': list(widths[:nb]),
                              'FUSE_METHOD': 'SUM'}
    return ex


HRNET_W32 = hrnet_extra((32, 64, 128, 256))
HRNET_W48 = hrnet_extra((48, 96, 192, 384))
HRNET_TINY = hrnet_extra((8, 16, 32, 64), modules=(1, 2, 1), blocks=2)

RES50 = {'FINAL_CONV_KERNEL': 1, 'DECONV_WITH_BIAS': False, 'NUM_DECONV_LAYERS': 3,
         'NUM_DECONV_FILTERS': [256, 256, 256], 'NUM_DECONV_KERNELS': [4, 4, 4],
         'NUM_LAYERS': 50}
RES18_TINY = {'FINAL_CONV_KERNEL': 1, 'DECONV_WITH_BIAS': False, 'NUM_DECONV_LAYERS': 3,
              'NUM_DECONV_FILTERS': [32, 32, 32], 'NUM_DECONV_KERNELS': [4, 4, 4], 'NUM_LAYERS': 18}
