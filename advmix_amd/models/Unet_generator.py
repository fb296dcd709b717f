"""AdvMix augmentation generator (pix2pix U-Net) on the HIP plan executor.  Mirrors
lib/models/Unet_generator.py: ``UnetGenerator(input_nc, output_nc, num_downs, ngf=64, ...)``
with the nested state-dict keys ``model.model.1.model.3...``; default torch init (the
reference applies no custom init, tools/train.py:67)."""
from ..plan import PlanNet, unet_plan


class UnetGenerator(PlanNet):
    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_layer=None, use_dropout=False,
                 with_tanh=False):
        if use_dropout or with_tanh:
            raise NotImplementedError('AdvMix builds UnetGenerator(9, 3, n) without dropout/tanh')
        if num_downs < 5:
            raise ValueError('num_downs must be >= 5')
        super().__init__(unet_plan(input_nc, output_nc, num_downs, ngf))
        self.num_downs = num_downs

    def forward(self, x):
        f = 1 << self.num_downs
        if x.shape[2] % f or x.shape[3] % f:
            raise ValueError('UnetGenerator with %d downs needs H, W divisible by %d (got %dx%d)'
                             % (self.num_downs, f, x.shape[2], x.shape[3]))
        return super().forward(x)
