"""Functional CPU restatement of the two pose networks (oracle; test infra only).

Networks are pure functions of a flat ``{state_dict key: tensor}`` mapping, so
they consume exactly the reference's checkpoints.  ``*_spec`` enumerate the
keys/shapes; ``tests/golden/keys_*.json`` pins them against the reference.

Follows:
  HRNet      lib/models/pose_hrnet.py:28-98 (blocks), :101-265 (HR module),
             :274-460 (net + forward)
  SimpleBaseline  lib/models/pose_resnet.py:62-100 (Bottleneck), :103-207
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOM = 0.1          # pose_hrnet.py:18 / nn.BatchNorm2d default (same value)


class Ctx:
    """Forward-mode switches: train => batch statistics + running-stat update."""

    def __init__(self, P, train, momentum=BN_MOM):
        self.P = P
        self.train = train
        self.momentum = momentum


def _bn(c, x, name):
    P = c.P
    if c.train:
        P[name + '.num_batches_tracked'] += 1
    return F.batch_norm(x, P[name + '.running_mean'], P[name + '.running_var'],
                        P[name + '.weight'], P[name + '.bias'],
                        c.train, c.momentum, BN_EPS)


def _conv(c, x, name, stride=1, pad=0):
    return F.conv2d(x, c.P[name + '.weight'], c.P.get(name + '.bias'), stride, pad)


def _cb(c, x, conv, bn, stride=1, pad=1, relu=True):
    y = _bn(c, _conv(c, x, conv, stride, pad), bn)
    return F.relu(y) if relu else y


# ----------------------------------------------------------------------------
# residual blocks (pose_hrnet.py:28-98 == pose_resnet.py:29-100)
# ----------------------------------------------------------------------------
def basic_block(c, x, pre, stride=1):
    out = _cb(c, x, pre + '.conv1', pre + '.bn1', stride, 1)
    out = _cb(c, out, pre + '.conv2', pre + '.bn2', 1, 1, relu=False)
    res = x
    if (pre + '.downsample.0.weight') in c.P:
        res = _cb(c, x, pre + '.downsample.0', pre + '.downsample.1', stride, 0, relu=False)
    return F.relu(out + res)


def bottleneck(c, x, pre, stride=1):
    out = _cb(c, x, pre + '.conv1', pre + '.bn1', 1, 0)
    out = _cb(c, out, pre + '.conv2', pre + '.bn2', stride, 1)
    out = _cb(c, out, pre + '.conv3', pre + '.bn3', 1, 0, relu=False)
    res = x
    if (pre + '.downsample.0.weight') in c.P:
        res = _cb(c, x, pre + '.downsample.0', pre + '.downsample.1', stride, 0, relu=False)
    return F.relu(out + res)


# ----------------------------------------------------------------------------
# HRNet
# ----------------------------------------------------------------------------
def _hr_module(c, xs, pre, nblocks, multi_scale):
    nb = len(xs)
    xs = list(xs)
    for b in range(nb):                                  # pose_hrnet.py:251-252
        for k in range(nblocks[b]):
            xs[b] = basic_block(c, xs[b], '%s.branches.%d.%d' % (pre, b, k))
    if nb == 1:
        return xs
    outs = []
    for i in range(nb if multi_scale else 1):            # pose_hrnet.py:256-263
        acc = None
        for j in range(nb):
            fp = '%s.fuse_layers.%d.%d' % (pre, i, j)
            if j == i:
                t = xs[j]
            elif j > i:                                   # 1x1 + BN + nearest up (:206)
                t = _cb(c, xs[j], fp + '.0', fp + '.1', 1, 0, relu=False)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode='nearest')
            else:                                         # chain of 3x3 s2 (+ReLU but last)
                t = xs[j]
                for k in range(i - j):
                    t = _cb(c, t, '%s.%d.0' % (fp, k), '%s.%d.1' % (fp, k), 2, 1,
                            relu=(k != i - j - 1))
            acc = t if acc is None else acc + t
        outs.append(F.relu(acc))
    return outs


def _transition(c, ys, pre, n_pre, cur_channels, pre_channels):
    xs = []
    for i in range(len(cur_channels)):                    # pose_hrnet.py:323-356, :433-452
        if i < n_pre:
            if cur_channels[i] != pre_channels[i]:
                xs.append(_cb(c, ys[i], '%s.%d.0' % (pre, i), '%s.%d.1' % (pre, i), 1, 1))
            else:
                xs.append(ys[i])
        else:
            t = ys[-1]
            for j in range(i + 1 - n_pre):
                t = _cb(c, t, '%s.%d.%d.0' % (pre, i, j), '%s.%d.%d.1' % (pre, i, j), 2, 1)
            xs.append(t)
    return xs


def hrnet_forward(P, x, extra, train, momentum=BN_MOM):
    """pose_hrnet.py:425-460.  P is mutated (BN running stats) when train."""
    c = Ctx(P, train, momentum)
    x = _cb(c, x, 'conv1', 'bn1', 2, 1)
    x = _cb(c, x, 'conv2', 'bn2', 2, 1)
    for k in range(4):
        x = bottleneck(c, x, 'layer1.%d' % k)
    ys, pre_ch = [x], [256]
    for s in (2, 3, 4):
        cfg = extra['STAGE%d' % s]
        cur = list(cfg['NUM_CHANNELS'])
        xs = _transition(c, ys, 'transition%d' % (s - 1), len(pre_ch), cur, pre_ch)
        nmod = cfg['NUM_MODULES']
        for m in range(nmod):
            ms = not (s == 4 and m == nmod - 1)           # :405-408
            xs = _hr_module(c, xs, 'stage%d.%d' % (s, m), cfg['NUM_BLOCKS'], ms)
        ys, pre_ch = xs, cur
    k = extra['FINAL_CONV_KERNEL']
    return _conv(c, ys[0], 'final_layer', 1, 1 if k == 3 else 0)


def _bn_spec(name, ch):
    return [(name + '.weight', (ch,)), (name + '.bias', (ch,)),
            (name + '.running_mean', (ch,)), (name + '.running_var', (ch,)),
            (name + '.num_batches_tracked', ())]


def _block_spec(kind, pre, cin, planes, stride=1):
    s = []
    if kind == 'basic':
        cout = planes
        s += [(pre + '.conv1.weight', (planes, cin, 3, 3))] + _bn_spec(pre + '.bn1', planes)
        s += [(pre + '.conv2.weight', (planes, planes, 3, 3))] + _bn_spec(pre + '.bn2', planes)
    else:
        cout = planes * 4
        s += [(pre + '.conv1.weight', (planes, cin, 1, 1))] + _bn_spec(pre + '.bn1', planes)
        s += [(pre + '.conv2.weight', (planes, planes, 3, 3))] + _bn_spec(pre + '.bn2', planes)
        s += [(pre + '.conv3.weight', (cout, planes, 1, 1))] + _bn_spec(pre + '.bn3', cout)
    if stride != 1 or cin != cout:
        s += [(pre + '.downsample.0.weight', (cout, cin, 1, 1))] + _bn_spec(pre + '.downsample.1', cout)
    return s, cout


def hrnet_spec(extra, num_joints):
    s = [('conv1.weight', (64, 3, 3, 3))] + _bn_spec('bn1', 64)
    s += [('conv2.weight', (64, 64, 3, 3))] + _bn_spec('bn2', 64)
    cin = 64
    for k in range(4):
        b, cin = _block_spec('bottleneck', 'layer1.%d' % k, cin, 64)
        s += b
    pre_ch = [256]
    for st in (2, 3, 4):
        cfg = extra['STAGE%d' % st]
        cur = list(cfg['NUM_CHANNELS'])
        tp = 'transition%d' % (st - 1)
        for i in range(len(cur)):
            if i < len(pre_ch):
                if cur[i] != pre_ch[i]:
                    s += [('%s.%d.0.weight' % (tp, i), (cur[i], pre_ch[i], 3, 3))]
                    s += _bn_spec('%s.%d.1' % (tp, i), cur[i])
            else:
                n = i + 1 - len(pre_ch)
                for j in range(n):
                    co = cur[i] if j == n - 1 else pre_ch[-1]
                    s += [('%s.%d.%d.0.weight' % (tp, i, j), (co, pre_ch[-1], 3, 3))]
                    s += _bn_spec('%s.%d.%d.1' % (tp, i, j), co)
        nb = len(cur)
        nmod = cfg['NUM_MODULES']
        for m in range(nmod):
            mp = 'stage%d.%d' % (st, m)
            for b in range(nb):
                for k in range(cfg['NUM_BLOCKS'][b]):
                    blk, _ = _block_spec('basic', '%s.branches.%d.%d' % (mp, b, k), cur[b], cur[b])
                    s += blk
            ms = not (st == 4 and m == nmod - 1)
            for i in range(nb if ms else 1):
                for j in range(nb):
                    fp = '%s.fuse_layers.%d.%d' % (mp, i, j)
                    if j > i:
                        s += [(fp + '.0.weight', (cur[i], cur[j], 1, 1))] + _bn_spec(fp + '.1', cur[i])
                    elif j < i:
                        for k in range(i - j):
                            co = cur[i] if k == i - j - 1 else cur[j]
                            s += [('%s.%d.0.weight' % (fp, k), (co, cur[j], 3, 3))]
                            s += _bn_spec('%s.%d.1' % (fp, k), co)
        pre_ch = cur
    k = extra['FINAL_CONV_KERNEL']
    s += [('final_layer.weight', (num_joints, pre_ch[0], k, k)), ('final_layer.bias', (num_joints,))]
    return s


# ----------------------------------------------------------------------------
# SimpleBaseline (pose_resnet.py)
# ----------------------------------------------------------------------------
RESNET_LAYERS = {18: ('basic', [2, 2, 2, 2]), 34: ('basic', [3, 4, 6, 3]),
                 50: ('bottleneck', [3, 4, 6, 3]), 101: ('bottleneck', [3, 4, 23, 3]),
                 152: ('bottleneck', [3, 8, 36, 3])}     # pose_resnet.py:252-258


def resnet_forward(P, x, extra, train, momentum=BN_MOM):
    """pose_resnet.py:193-207."""
    c = Ctx(P, train, momentum)
    kind, layers = RESNET_LAYERS[extra['NUM_LAYERS']]
    blk = basic_block if kind == 'basic' else bottleneck
    x = _cb(c, x, 'conv1', 'bn1', 2, 3)
    x = F.max_pool2d(x, 3, 2, 1)
    for li, n in enumerate(layers):
        for k in range(n):
            x = blk(c, x, 'layer%d.%d' % (li + 1, k), 2 if (k == 0 and li > 0) else 1)
    for i in range(extra['NUM_DECONV_LAYERS']):          # :160-191, kernel 4 -> pad 1, op 0
        kk = extra['NUM_DECONV_KERNELS'][i]
        pad, op = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[kk]
        x = F.conv_transpose2d(x, P['deconv_layers.%d.weight' % (3 * i)],
                               P.get('deconv_layers.%d.bias' % (3 * i)), 2, pad, op)
        x = F.relu(_bn(c, x, 'deconv_layers.%d' % (3 * i + 1)))
    k = extra['FINAL_CONV_KERNEL']
    return _conv(c, x, 'final_layer', 1, 1 if k == 3 else 0)


def resnet_spec(extra, num_joints):
    kind, layers = RESNET_LAYERS[extra['NUM_LAYERS']]
    s = [('conv1.weight', (64, 3, 7, 7))] + _bn_spec('bn1', 64)
    cin = 64
    for li, n in enumerate(layers):
        planes = 64 * 2 ** li
        for k in range(n):
            b, cin = _block_spec(kind, 'layer%d.%d' % (li + 1, k), cin, planes,
                                 2 if (k == 0 and li > 0) else 1)
            s += b
    for i in range(extra['NUM_DECONV_LAYERS']):
        co = extra['NUM_DECONV_FILTERS'][i]
        kk = extra['NUM_DECONV_KERNELS'][i]
        s += [('deconv_layers.%d.weight' % (3 * i), (cin, co, kk, kk))]
        if extra.get('DECONV_WITH_BIAS', False):
            s += [('deconv_layers.%d.bias' % (3 * i), (co,))]
        s += _bn_spec('deconv_layers.%d' % (3 * i + 1), co)
        cin = co
    k = extra['FINAL_CONV_KERNEL']
    s += [('final_layer.weight', (num_joints, cin, k, k)), ('final_layer.bias', (num_joints,))]
    return s


def posenet_forward(name, P, x, extra, train, momentum=BN_MOM):
    return (hrnet_forward if name == 'pose_hrnet' else resnet_forward)(P, x, extra, train, momentum)


def calibrate(name, P, x, extra):
    """Fixture helper (not reference behaviour): one train-mode pass with BN
    momentum 1.0 so running stats == this batch's stats; keeps eval-mode
    (teacher) activations O(1) under the non-degenerate detinit weights."""
    with torch.no_grad():
        posenet_forward(name, P, x, extra, True, momentum=1.0)


def posenet_spec(name, extra, num_joints):
    return (hrnet_spec if name == 'pose_hrnet' else resnet_spec)(extra, num_joints)


def trainable(P):
    """Names autograd should track: every float tensor that is not a BN buffer."""
    return [k for k, v in P.items()
            if v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))]
