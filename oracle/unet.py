"""Functional CPU restatement of the augmentation generator (oracle; test infra).

Follows lib/models/Unet_generator.py:13-112.  Quirk reproduced:
every non-outermost block starts with an IN-PLACE LeakyReLU(0.2) (:42,61,69) so
the skip branch of ``torch.cat([x, model(x)], 1)`` (:83) carries LeakyReLU(x).
InstanceNorm2d is affine=False, no running stats, eps 1e-5 -> no state-dict keys.
"""
import torch
import torch.nn.functional as F


def unet_levels(input_nc, output_nc, num_downs, ngf=64):
    """[(prefix, down_in, down_out, up_in, up_out, down_idx, up_idx)] outer->inner."""
    chans = [(output_nc, ngf, input_nc), (ngf, ngf * 2, None), (ngf * 2, ngf * 4, None),
             (ngf * 4, ngf * 8, None)] + [(ngf * 8, ngf * 8, None)] * (num_downs - 5) \
        + [(ngf * 8, ngf * 8, None)]
    lv, pre = [], 'model'
    n = len(chans)
    for i, (outer, inner, inp) in enumerate(chans):
        inp = outer if inp is None else inp
        innermost = i == n - 1
        up_in = inner if innermost else inner * 2
        if i == 0:
            di, ui, si = 0, 3, 1
        elif innermost:
            di, ui, si = 1, 3, None
        else:
            di, ui, si = 1, 5, 3
        lv.append(dict(pre=pre, din=inp, dout=inner, uin=up_in, uout=outer, di=di, ui=ui))
        if si is not None:
            pre = '%s.model.%d' % (pre, si)
    return lv


def unet_spec(input_nc=9, output_nc=3, num_downs=6, ngf=64):
    s = []
    for L in unet_levels(input_nc, output_nc, num_downs, ngf):
        d = '%s.model.%d' % (L['pre'], L['di'])
        u = '%s.model.%d' % (L['pre'], L['ui'])
        s += [(d + '.weight', (L['dout'], L['din'], 4, 4)), (d + '.bias', (L['dout'],))]
        s += [(u + '.weight', (L['uin'], L['uout'], 4, 4)), (u + '.bias', (L['uout'],))]
    return s


def unet_transposed_names(input_nc=9, output_nc=3, num_downs=6, ngf=64):
    return ['%s.model.%d.weight' % (L['pre'], L['ui'])
            for L in unet_levels(input_nc, output_nc, num_downs, ngf)]


def unet_forward(P, x, input_nc=9, output_nc=3, num_downs=6, ngf=64):
    lv = unet_levels(input_nc, output_nc, num_downs, ngf)

    def down(L, t):
        d = '%s.model.%d' % (L['pre'], L['di'])
        return F.conv2d(t, P[d + '.weight'], P[d + '.bias'], 2, 1)

    def up(L, t):
        u = '%s.model.%d' % (L['pre'], L['ui'])
        return F.conv_transpose2d(t, P[u + '.weight'], P[u + '.bias'], 2, 1)

    def run(i, t):
        L = lv[i]
        if i == 0:                                        # outermost: :51-57, with_tanh False
            return up(L, F.relu(run(1, down(L, t))))
        a = F.leaky_relu(t, 0.2)                          # the in-place downrelu
        if i == len(lv) - 1:                              # innermost :58-65
            u = F.instance_norm(up(L, F.relu(down(L, a))), eps=1e-5)
        else:                                             # :66-77
            s = run(i + 1, F.instance_norm(down(L, a), eps=1e-5))
            u = F.instance_norm(up(L, F.relu(s)), eps=1e-5)
        return torch.cat([a, u], 1)                       # :83

    return run(0, x)
