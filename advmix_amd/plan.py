"""Static execution plans for the hot-path networks.

Instead of one nn.Module per layer, a network is compiled ONCE into a flat list of
fused kernel-level steps over numbered activation slots (conv, conv-transpose,
norm+residual+activation, cat+activation, multi-resolution fuse, max-pool).  The same
plan drives parameter registration (names/shapes identical to the reference's
state-dict keys) and execution, and is what a HIP-graph capture replays.

Builders below restate the topologies of lib/models/pose_hrnet.py:274-460,
lib/models/pose_resnet.py:103-207 and lib/models/Unet_generator.py:13-112.
"""
import math
import os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_LEAKY

BN_MOMENTUM = 0.1     # pose_hrnet.py:18; fuse/transition BNs use the torch default (also 0.1)
BN_EPS = 1e-5


class Plan:
    def __init__(self, in_channels):
        self.params = []          # (name, shape, kind) kind in conv|deconv|bias|bn_w|bn_b
        self.buffers = []         # (name, shape, kind) kind in rm|rv|nbt
        self.steps = []
        self.ch = [in_channels]   # channels per slot; slot 0 is the input
        self.out = None
        self.tag = None           # launch-chain label given to the steps created next (None = standalone)
        self.slot_tag = [None]    # per slot: the label of the step that produces it

    # -- slot helpers ------------------------------------------------------------------
    def _new(self, c):
        self.ch.append(c)
        self.slot_tag.append(self.tag)
        return len(self.ch) - 1

    def conv(self, x, name, cout, k, stride, pad, bias=False):
        self.params.append((name + '.weight', (cout, self.ch[x], k, k), 'conv'))
        if bias:
            self.params.append((name + '.bias', (cout,), 'bias:%d' % (self.ch[x] * k * k)))
        y = self._new(cout)
        self.steps.append(('conv', name, x, y, stride, pad, bias))
        return y

    def deconv(self, x, name, cout, k, stride, pad, bias=False):
        self.params.append((name + '.weight', (self.ch[x], cout, k, k), 'deconv'))
        if bias:
            self.params.append((name + '.bias', (cout,), 'bias:%d' % (cout * k * k)))
        y = self._new(cout)
        self.steps.append(('deconv', name, x, y, stride, pad, bias))
        return y

    def bn(self, x, name, act=ACT_NONE, res=None):
        c = self.ch[x]
        self.params += [(name + '.weight', (c,), 'bn_w'), (name + '.bias', (c,), 'bn_b')]
        self.buffers += [(name + '.running_mean', (c,), 'rm'), (name + '.running_var', (c,), 'rv'),
                         (name + '.num_batches_tracked', (), 'nbt')]
        y = self._new(c)
        self.steps.append(('bn', name, x, y, res, act))
        return y

    def conv_bn(self, x, cname, bname, cout, k, stride, pad, act, res=None):
        return self.bn(self.conv(x, cname, cout, k, stride, pad), bname, act, res)

    def inorm(self, x, act=ACT_NONE):
        y = self._new(self.ch[x])
        self.steps.append(('inorm', x, y, act))
        return y

    def act(self, x, act):
        y = self._new(self.ch[x])
        self.steps.append(('act', x, y, act))
        return y

    def catact(self, a, b, act=ACT_NONE):
        y = self._new(self.ch[a] + self.ch[b])
        self.steps.append(('catact', a, b, y, act))
        return y

    def fuse(self, xs, shifts, act=ACT_RELU):
        y = self._new(self.ch[xs[shifts.index(0)]])
        self.steps.append(('fuse', list(xs), list(shifts), y, act))
        return y

    def maxpool(self, x):
        y = self._new(self.ch[x])
        self.steps.append(('maxpool', x, y))
        return y

    # -- residual blocks (pose_hrnet.py:28-98 / pose_resnet.py:29-100) ------------------
    def block(self, kind, x, pre, planes, stride=1):
        cin = self.ch[x]
        if kind == 'BASIC':
            cout = planes
            o = self.conv_bn(x, pre + '.conv1', pre + '.bn1', planes, 3, stride, 1, ACT_RELU)
            o = self.conv(o, pre + '.conv2', planes, 3, 1, 1)
            last_bn = pre + '.bn2'
        else:
            cout = planes * 4
            o = self.conv_bn(x, pre + '.conv1', pre + '.bn1', planes, 1, 1, 0, ACT_RELU)
            o = self.conv_bn(o, pre + '.conv2', pre + '.bn2', planes, 3, stride, 1, ACT_RELU)
            o = self.conv(o, pre + '.conv3', cout, 1, 1, 0)
            last_bn = pre + '.bn3'
        res = x
        if stride != 1 or cin != cout:
            res = self.conv_bn(x, pre + '.downsample.0', pre + '.downsample.1', cout, 1, stride, 0, ACT_NONE)
        return self.bn(o, last_bn, ACT_RELU, res)


EXPANSION = {'BASIC': 1, 'BOTTLENECK': 4}


def hrnet_plan(extra, num_joints):
    P = Plan(3)
    # Launch chains (see PlanNet): the stem is one chain; inside a stage, chain (module m, branch b) is
    # [fuse sum that produced the branch input, the branch's blocks, the fuse convolutions that read the
    # branch output] - everything a branch does between two exchanges with the other branches.
    P.tag = 'stem'
    x = P.conv_bn(0, 'conv1', 'bn1', 64, 3, 2, 1, ACT_RELU)
    x = P.conv_bn(x, 'conv2', 'bn2', 64, 3, 2, 1, ACT_RELU)
    for k in range(4):
        x = P.block('BOTTLENECK', x, 'layer1.%d' % k, 64)
    cur = [x]
    for st in (2, 3, 4):
        cfg = extra['STAGE%d' % st]
        kind = cfg['BLOCK']
        widths = [c * EXPANSION[kind] for c in cfg['NUM_CHANNELS']]
        nb = cfg['NUM_BRANCHES']
        assert nb == len(widths) == len(cfg['NUM_BLOCKS'])
        tp = 'transition%d' % (st - 1)
        nxt = []
        for i in range(nb):                               # pose_hrnet.py:323-356 + :433-452
            # a transition conv opens its branch's chain - except where ONE tensor feeds several of them (layer1's output
            # under transition1's two convs): those ride at the tail of the producer's chain, where the second input gradient
            # takes the first as its epilogue's addend; from two chains the autograd engine adds them itself with a torch
            # kernel on the critical path (100 MB tensors at 64x48x256, B = 32: tools/fanin_report.py)
            P.tag = 's%d.m0.b%d' % (st, i) if len(cur) > 1 else 'stem'
            if i < len(cur):
                if P.ch[cur[i]] != widths[i]:
                    nxt.append(P.conv_bn(cur[i], '%s.%d.0' % (tp, i), '%s.%d.1' % (tp, i),
                                         widths[i], 3, 1, 1, ACT_RELU))
                else:
                    nxt.append(cur[i])
            else:
                t, n = cur[-1], i + 1 - len(cur)
                for j in range(n):
                    co = widths[i] if j == n - 1 else P.ch[cur[-1]]
                    t = P.conv_bn(t, '%s.%d.%d.0' % (tp, i, j), '%s.%d.%d.1' % (tp, i, j), co, 3, 2, 1, ACT_RELU)
                nxt.append(t)
        cur = nxt
        nmod = cfg['NUM_MODULES']
        for m in range(nmod):
            mp = 'stage%d.%d' % (st, m)
            multi = not (st == 4 and m == nmod - 1)       # :405-408
            for b in range(nb):
                P.tag = 's%d.m%d.b%d' % (st, m, b)
                for k in range(cfg['NUM_BLOCKS'][b]):
                    cur[b] = P.block(kind, cur[b], '%s.branches.%d.%d' % (mp, b, k), cfg['NUM_CHANNELS'][b])
            if nb == 1:
                continue
            fused = []
            for i in range(nb if multi else 1):           # :196-232, :254-265
                srcs, shifts = [], []
                for j in range(nb):
                    fp = '%s.fuse_layers.%d.%d' % (mp, i, j)
                    P.tag = 's%d.m%d.b%d' % (st, m, j)     # fuse convs ride at the tail of their SOURCE branch
                    if j == i:
                        srcs.append(cur[j]); shifts.append(0)
                    elif j > i:
                        srcs.append(P.conv_bn(cur[j], fp + '.0', fp + '.1', P.ch[cur[i]], 1, 1, 0, ACT_NONE))
                        shifts.append(j - i)
                    else:
                        t = cur[j]
                        for k in range(i - j):
                            last = k == i - j - 1
                            t = P.conv_bn(t, '%s.%d.0' % (fp, k), '%s.%d.1' % (fp, k),
                                          P.ch[cur[i]] if last else P.ch[cur[j]], 3, 2, 1,
                                          ACT_NONE if last else ACT_RELU)
                        srcs.append(t); shifts.append(0)
                # the sum opens the next module's chain of branch i; at a stage boundary it stands alone so
                # that the new branch's transition does not have to wait for a whole sibling chain
                P.tag = ('s%d.m%d.b%d' % (st, m + 1, i)) if m + 1 < nmod else ('head' if st == 4 else None)
                fused.append(P.fuse(srcs, shifts, ACT_RELU))
            cur = fused
    P.tag = 'head'
    k = extra['FINAL_CONV_KERNEL']
    P.out = P.conv(cur[0], 'final_layer', num_joints, k, 1, 1 if k == 3 else 0, bias=True)
    return P


RESNET_SPEC = {18: ('BASIC', [2, 2, 2, 2]), 34: ('BASIC', [3, 4, 6, 3]), 50: ('BOTTLENECK', [3, 4, 6, 3]),
               101: ('BOTTLENECK', [3, 4, 23, 3]), 152: ('BOTTLENECK', [3, 8, 36, 3])}


def resnet_plan(extra, num_joints):
    kind, layers = RESNET_SPEC[extra['NUM_LAYERS']]
    P = Plan(3)
    P.tag = 'all'                                          # a sequential network: one launch chain
    x = P.conv_bn(0, 'conv1', 'bn1', 64, 7, 2, 3, ACT_RELU)
    x = P.maxpool(x)
    for li, n in enumerate(layers):
        for k in range(n):
            x = P.block(kind, x, 'layer%d.%d' % (li + 1, k), 64 * 2 ** li, 2 if (k == 0 and li > 0) else 1)
    for i in range(extra['NUM_DECONV_LAYERS']):           # pose_resnet.py:145-191
        kk = extra['NUM_DECONV_KERNELS'][i]
        if kk != 4:
            raise NotImplementedError('only 4x4 stride-2 deconv heads are on the hot path')
        x = P.deconv(x, 'deconv_layers.%d' % (3 * i), extra['NUM_DECONV_FILTERS'][i], 4, 2, 1,
                     bias=bool(extra.get('DECONV_WITH_BIAS', False)))
        x = P.bn(x, 'deconv_layers.%d' % (3 * i + 1), ACT_RELU)
    k = extra['FINAL_CONV_KERNEL']
    P.out = P.conv(x, 'final_layer', num_joints, k, 1, 1 if k == 3 else 0, bias=True)
    return P


def unet_plan(input_nc, output_nc, num_downs, ngf=64):
    """U-Net with the reference's quirks folded into fused steps: the in-place LeakyReLU that
    also feeds the skip (Unet_generator.py:42,61,69,83) becomes norm+leaky producing the skip
    tensor; the parent's in-place ReLU on the concatenation becomes cat+relu."""
    inner = [ngf, ngf * 2, ngf * 4, ngf * 8] + [ngf * 8] * (num_downs - 4)
    P = Plan(input_nc)
    P.tag = 'all'                                          # a sequential network: one launch chain
    pre = ['model']
    for i in range(1, num_downs):
        pre.append(pre[-1] + ('.model.1' if i == 1 else '.model.3'))

    def names(i):
        if i == 0:
            return pre[0] + '.model.0', pre[0] + '.model.3'
        if i == num_downs - 1:
            return pre[i] + '.model.1', pre[i] + '.model.3'
        return pre[i] + '.model.1', pre[i] + '.model.5'

    def seg(i, up):
        """Launch chain of block i's down (up) half: outer / middle / inner thirds of the U.  One network pass runs
        them one after the other either way; three chains instead of one give the data-parallel backward two points
        where a finished range of the flat gradient buffer can be all-reduced while the rest still runs."""
        third = 0 if i < num_downs // 3 else (1 if i < 2 * num_downs // 3 else 2)
        return 'u%d' % (4 - third if up else third)           # d0 d1 | d2 d3 | d4 d5 u5 u4 | u3 u2 | u1 u0

    def level(i, a):
        """a = LeakyReLU'd input of block i (i >= 1); returns ReLU(cat([a, up-branch]))."""
        dn, un = names(i)
        P.tag = seg(i, False)
        d = P.conv(a, dn, inner[i], 4, 2, 1, bias=True)
        if i == num_downs - 1:
            r = P.act(d, ACT_RELU)
        else:
            # the skip tensor is made in the chain of its FIRST consumer (block i + 1's conv): made a chain earlier it would
            # leave that chain for two others - the next down chain and the up chain that concatenates it - and the autograd
            # engine would add their two gradients with a torch kernel (50 MB tensors at 64x48x128, B = 32: tools/fanin_report.py)
            P.tag = seg(i + 1, False)
            r = level(i + 1, P.inorm(d, ACT_LEAKY))
        P.tag = seg(i, True)
        u = P.inorm(P.deconv(r, un, P.ch[a], 4, 2, 1, bias=True), ACT_NONE)
        return P.catact(a, u, ACT_RELU)

    dn, un = names(0)
    P.tag = seg(0, False)
    d = P.conv(0, dn, inner[0], 4, 2, 1, bias=True)
    r = level(1, P.act(d, ACT_LEAKY))
    P.tag = seg(0, True)
    P.out = P.deconv(r, un, output_nc, 4, 2, 1, bias=True)
    return P


# ---------------------------------------------------------------------------------------------
CHAINS = os.environ.get('ADVMIX_CHAINS', '1') != '0'


class _Slot:
    def __init__(self, slot):
        self.slot = slot


class _SlotRefs:
    """Stands in for the slot table while a chain's static description is built."""

    def __getitem__(self, s):
        return _Slot(s)


class _Name:
    def __init__(self, name):
        self.name = name


class _NameRefs:
    """Stands in for the parameter table: records which named tensor a sub-member wants."""

    def __getitem__(self, n):
        return _Name(n)


class PlanNet(nn.Module):
    """nn.Module whose parameters/buffers are registered under the reference's state-dict
    keys and whose forward interprets a Plan with the HIP ops."""

    def __init__(self, plan):
        super().__init__()
        self.plan = plan
        for name, shape, kind in plan.params:
            self._register(name, nn.Parameter(self._default_init(shape, kind)), False)
        for name, shape, kind in plan.buffers:
            if kind == 'nbt':
                t = torch.zeros((), dtype=torch.int64)
            else:
                t = torch.zeros(shape) if kind == 'rm' else torch.ones(shape)
            self._register(name, t, True)
        # conv -> BatchNorm pairs become one fused member (ops.ConvBN) when the conv has no bias and its
        # output feeds only that BatchNorm
        steps = self._fuse_conv_bn(plan.steps)
        # Launch chains: dependent runs of steps that execute back to back on one lane (ops.Chain).
        # Chains of one level are mutually independent (HRNet's branches with their fuse convolutions)
        # and are launched as ONE concurrent group; lanes only join between levels (once per HRNet
        # module instead of once per conv+BN).  ADVMIX_CHAINS=0 restores one step per chain.
        steps = self._build_chains(steps, plan.out, plan.slot_tag) if CHAINS else steps
        level = {0: 0}
        groups = {}
        for st in steps:
            lvl = 1 + max(level[s] for s in self._srcs(st))
            for d in self._dsts(st):
                level[d] = lvl
            groups.setdefault(lvl, []).append(st)
        self._levels = [groups[k] for k in sorted(groups)]
        self._last_use, self._use_levels = {}, {}
        for li, sts in enumerate(self._levels):
            for st in sts:
                for s in self._srcs(st):
                    self._last_use[s] = li
                    self._use_levels.setdefault(s, set()).add(li)
        self._cache = None
        self._chain_meta = {}
        # execution order of the parameters (FlatAdam lays its flat buffers out in it) and, per level, how many
        # parameter elements the levels up to it own: dp.GradSync cuts the backward pass there
        rank, self._level_elems = 0, []
        plist = dict(self.named_parameters())
        acc = 0
        for sts in self._levels:
            for st in sts:
                for sub in (st[1] if st[0] == 'chain' else (st,)):
                    for nm in self._param_names(sub):
                        plist[nm]._flat_rank = rank
                        rank += 1
                        acc += (plist[nm].numel() + 3) // 4 * 4
            self._level_elems.append(acc)
        # Levels after which the NEXT forward() cuts the autograd graph: every activation alive across the boundary is
        # replaced by a detached twin (same memory) for the levels above, and (original, twin) is recorded in
        # ``last_cuts``.  The backward pass then runs piece by piece (core.function._backward_pieces): the part above a
        # cut leaves its gradients in the twins' .grad, the next piece starts from the originals with those gradients.
        self.cut_levels = ()
        self.last_cuts = []
        # fp64 statistics slots (ops.StatArena): a forward and a backward set per conv + BatchNorm pair
        self._arena = ops.StatArena()
        self._stat_off = {}
        # conv + BN outputs whose ONLY consumer is a fuse sum (HRNet's fuse layers): the sum's backward produces their
        # BatchNorm-backward channel sums (ops.FuseSum.bwd)
        uses, fuse_srcs = {}, set()
        for st in plan.steps:
            for s_ in self._srcs(st):
                uses[s_] = uses.get(s_, 0) + 1
            if st[0] == 'fuse':
                fuse_srcs.update(st[1])
        self._fuse_only = {st[1] for st in plan.steps if st[0] == 'bn' and st[3] in fuse_srcs and uses.get(st[3], 0) == 1
                           and st[3] != plan.out and st[5] == ACT_NONE}
        ch_of = {name: shape[0] for name, shape, kind in plan.params if kind == 'bn_w'}
        for st in plan.steps:
            if st[0] == 'bn':
                c = ch_of[st[1] + '.weight']
                self._stat_off[st[1]] = (self._arena.reserve(c), self._arena.reserve(c))

    @staticmethod
    def _default_init(shape, kind):
        """torch.nn defaults (kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)); BN (1, 0))."""
        if kind in ('conv', 'deconv'):
            fan_in = shape[1] * shape[2] * shape[3]
            bound = 1.0 / math.sqrt(fan_in)
            w = torch.empty(shape).uniform_(-bound, bound)
            return w.contiguous(memory_format=torch.channels_last)
        if kind.startswith('bias'):
            bound = 1.0 / math.sqrt(int(kind.split(':')[1]))
            return torch.empty(shape).uniform_(-bound, bound)
        return torch.ones(shape) if kind == 'bn_w' else torch.zeros(shape)

    def _register(self, dotted, tensor, is_buffer):
        mod = self
        parts = dotted.split('.')
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        if is_buffer:
            mod.register_buffer(parts[-1], tensor)
        else:
            mod.register_parameter(parts[-1], tensor)

    @classmethod
    def _fuse_conv_bn(cls, steps):
        uses = {}
        for st in steps:
            for s in cls._srcs(st):
                uses[s] = uses.get(s, 0) + 1
        bn_of = {st[2]: st for st in steps if st[0] == 'bn'}          # conv-output slot -> its bn step
        out, fused_bn = [], set()
        for st in steps:
            if st[0] == 'conv' and not st[6] and uses.get(st[3], 0) == 1 and st[3] in bn_of:
                _, cname, x, y, stride, pad, _hb = st
                _, bname, _y, z, res, act = bn_of[y]
                out.append(('convbn', cname, bname, x, z, stride, pad, res, act))
                fused_bn.add(bname)
            elif st[0] == 'bn' and st[1] in fused_bn:
                continue
            else:
                out.append(st)
        # a fused step sits where its conv was; a residual produced later in program order (the
        # downsample path) is still earlier in LEVEL order, which is all the scheduler uses
        return cls._toposort(out)

    @classmethod
    def _build_chains(cls, steps, out_slot, slot_tag):
        """Contract the steps that carry the same chain label (given by the plan builder) into one
        ('chain', steps, external source slots, output slots) super-step.  The contracted graph must
        stay acyclic (``_toposort`` raises otherwise); levels are then taken over super-steps."""
        by_tag, chains = {}, []                            # chains: [None, [steps]] in first-step order
        for st in steps:
            tag = slot_tag[cls._dst(st)]
            if tag is None or tag not in by_tag:
                chains.append([None, [st]])
                if tag is not None:
                    by_tag[tag] = len(chains) - 1
            else:
                chains[by_tag[tag]][1].append(st)
        consumers = {}
        for ci, (_, sts) in enumerate(chains):
            for st in sts:
                for s in cls._srcs(st):
                    consumers.setdefault(s, set()).add(ci)
        out = []
        for ci, (_, sts) in enumerate(chains):
            if len(sts) == 1:
                out.append(sts[0])
                continue
            made = [cls._dst(st) for st in sts]
            ext, seen = [], set(made)
            for st in sts:
                for s in cls._srcs(st):
                    if s not in seen:
                        seen.add(s)
                        ext.append(s)
            outs = [d for d in made if d == out_slot or (consumers.get(d, set()) - {ci})]
            out.append(('chain', tuple(sts), tuple(ext), tuple(outs)))
        return cls._toposort(out)

    @classmethod
    def _toposort(cls, steps):
        ready, done, out = {0}, set(), []
        pending = list(steps)
        while pending:
            rest = []
            for st in pending:
                if all(s in ready for s in cls._srcs(st)):
                    out.append(st)
                    ready.update(cls._dsts(st))
                else:
                    rest.append(st)
            if len(rest) == len(pending):
                raise RuntimeError('plan has a dependency cycle')
            pending = rest
        return out

    @staticmethod
    def _param_names(st):
        k = st[0]
        if k == 'convbn':
            return [st[1] + '.weight', st[2] + '.weight', st[2] + '.bias']
        if k in ('conv', 'deconv'):
            return [st[1] + '.weight'] + ([st[1] + '.bias'] if st[6] else [])
        if k == 'bn':
            return [st[1] + '.weight', st[1] + '.bias']
        return []

    def plan_cuts(self, pieces=3):
        """Choose ``pieces - 1`` level boundaries that split the flat parameter buffer into roughly equal ranges and
        returns [(level, first flat element owned by the levels above it)], ascending.  With ``cut_levels`` set to those
        levels for a forward pass, the backward run piecewise (core.function._backward_pieces) finishes the gradient
        range [elem, end) when the piece above the cut is done."""
        total = self._level_elems[-1]
        cuts = []
        for k in range(1, pieces):
            want = total * k / pieces
            li = min(range(len(self._levels) - 1), key=lambda i: abs(self._level_elems[i] - want), default=None)
            if li is not None and (li, self._level_elems[li]) not in cuts and 0 < self._level_elems[li] < total:
                cuts.append((li, self._level_elems[li]))
        cuts.sort()
        return cuts

    @staticmethod
    def _dst(st):
        k = st[0]
        if k == 'convbn':
            return st[4]
        if k in ('conv', 'deconv', 'bn', 'catact', 'fuse'):
            return st[3]
        return st[2]                                       # inorm / act / maxpool

    @classmethod
    def _dsts(cls, st):
        return list(st[3]) if st[0] == 'chain' else [cls._dst(st)]

    @staticmethod
    def _srcs(st):
        k = st[0]
        if k == 'chain':
            return list(st[2])
        if k == 'convbn':
            return [st[3]] + ([st[7]] if st[7] is not None else [])
        if k in ('conv', 'deconv'):
            return [st[2]]
        if k == 'bn':
            return [st[2]] + ([st[4]] if st[4] is not None else [])
        if k in ('inorm', 'act', 'maxpool'):
            return [st[1]]
        if k == 'catact':
            return [st[1], st[2]]
        if k == 'fuse':
            return list(st[1])
        raise ValueError(k)

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._cache = None
        for p in self.parameters():                       # keep conv weights [O][R][S][I] in memory
            if p.dim() == 4 and not p.is_contiguous(memory_format=torch.channels_last):
                p.data = p.data.contiguous(memory_format=torch.channels_last)
        return r

    def _tensors(self):
        if self._cache is None:
            d = dict(self.named_parameters())
            d.update(dict(self.named_buffers()))
            self._cache = d
        return self._cache

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self._cache = None
        self.invalidate_filter_images()
        return r

    def train(self, mode=True):
        """nn.Module.train; a network that (re-)enters training mode is no longer the frozen teacher core.function marked
        ``wino_static``: its filters will change through raw kernels (FlatAdam) that torch's version counters do not see, so
        the mark and the cached images' key are dropped."""
        if mode and getattr(self, 'wino_static', False):
            self.wino_static = False
            self.invalidate_filter_images()
        return super().train(mode)

    def invalidate_filter_images(self):
        """Have the next forward pass re-make the Winograd / small-map filter images even for a network marked
        ``wino_static`` (see _wino_refresh).  Needed only after writing a frozen network's filters in a way torch's version
        counters do not see: through ``param.data``, a collective, a raw kernel."""
        bank = getattr(self, '_wino_bank', None)
        if bank is not None:
            bank.static_key = None

    def _chain_member(self, st, slots, T, train):
        """ops.Chain member of a ('chain', steps, ext_slots, out_slots) super-step.  The static part
        (which tensor of which sub-member is a slot, which a parameter) is built once per mode."""
        key = (id(st), train)
        cached = self._chain_meta.get(key)
        if cached is None:
            _, sts, ext, outs = st
            ref = _SlotRefs()
            subs, names = [], []
            for sub in sts:
                op, tensors, meta = self._member(sub, ref, _NameRefs(), train)
                refs = []
                for t in tensors:
                    if t is None:
                        refs.append(None)
                    elif isinstance(t, _Slot):
                        refs.append(('s', t.slot))
                    else:
                        refs.append(('i', len(ext) + len(names)))
                        names.append(t.name)
                subs.append((op, tuple(refs), meta, self._dst(sub)))
            subs = tuple(subs)
            cached = ((subs, tuple(ext), tuple(outs), ops.Chain.bnb_plan(subs), ops.Chain.inbn_plan(subs, tuple(outs))), tuple(names))
            self._chain_meta[key] = cached
        meta, names = cached
        return (ops.Chain, tuple(slots[s] for s in meta[1]) + tuple(T[n] for n in names), meta)

    def _member(self, st, slots, T, train):
        k = st[0]
        if k == 'chain':
            return self._chain_member(st, slots, T, train)
        if k == 'convbn':
            _, cname, bname, s, d, stride, pad, res, act = st
            return (ops.ConvBN,
                    (slots[s], T[cname + '.weight'], T[bname + '.weight'], T[bname + '.bias'],
                     T[bname + '.running_mean'], T[bname + '.running_var'], T[bname + '.num_batches_tracked'],
                     slots[res] if res is not None else None),
                    (stride, pad, act, train, BN_MOMENTUM, BN_EPS, self._arena) + self._stat_off[bname]
                    + (bname in self._fuse_only,))
        if k in ('conv', 'deconv'):
            _, name, s, d, stride, pad, hb = st
            return (ops.Conv if k == 'conv' else ops.Deconv,
                    (slots[s], T[name + '.weight'], T[name + '.bias'] if hb else None), (stride, pad))
        if k == 'bn':
            _, name, s, d, res, act = st
            return (ops.BatchNorm,
                    (slots[s], T[name + '.weight'], T[name + '.bias'], T[name + '.running_mean'],
                     T[name + '.running_var'], T[name + '.num_batches_tracked'],
                     slots[res] if res is not None else None), (act, train, BN_MOMENTUM, BN_EPS))
        if k == 'inorm':
            return (ops.InstanceNorm, (slots[st[1]],), (st[3], 1e-5))
        if k == 'act':
            return (ops.Act, (slots[st[1]],), st[3])
        if k == 'catact':
            return (ops.CatAct, (slots[st[1]], slots[st[2]]), st[4])
        if k == 'fuse':
            return (ops.FuseSum, tuple(slots[s] for s in st[1]), (st[4], list(st[2])))
        if k == 'maxpool':
            return (ops.MaxPool, (slots[st[1]],), None)
        raise ValueError(k)

    def _wino_refresh(self, device):
        """Winograd path (ops.WinoBank, csrc/conv_wino.hip): the 3x3 / stride 1 / pad 1 conv + BatchNorm steps with 32, 48, 64, 96 or
        128 input channels (and the 256 -> 256 ones: csrc/conv_smap.hip's re-laid filters) keep transformed images of their filters in a side buffer; ONE launch re-computes all of them at the
        start of every forward pass (the filters change once per optimizer step; ~10 us per launch), on the caller's stream
        before any lane forks.  The bank is rebuilt when the parameters have moved (``.to(device)``, FlatAdam's flat
        buffer, ``load_state_dict`` of differently placed tensors)."""
        if not ops.WINO:
            return
        names = getattr(self, '_wino_names', None)
        if names is None:
            names, self._wino_first_level = [], None
            for li, sts in enumerate(self._levels):
                for st in sts:
                    for sub in (st[1] if st[0] == 'chain' else (st,)):
                        if (sub[0] == 'convbn' and sub[5] == 1 and sub[6] in (0, 1)) or \
                                (ops.WINO4 and sub[0] in ('conv', 'deconv') and sub[4] == 2 and sub[5] == 1):   # stride 1: 3x3 / pad 1, or 1x1 / pad 0; the U-Net's 4x4 / stride 2 / pad 1 (csrc/conv_wino4.hip)
                            names.append(sub[1] + '.weight')
                            if self._wino_first_level is None:
                                self._wino_first_level = li    # (the stem's strided convs come first)
            self._wino_names = names
        T = self._tensors()
        ws = [T[n] for n in names]
        ws = [w for w in ws if w.is_cuda and w.shape[0] % 16 == 0 and (
              (tuple(w.shape[2:]) == (3, 3) and (w.shape[1] in (32, 48, 64, 96, 128) or (ops.SMAP and tuple(w.shape[:2]) == (ops.SMAP_C, ops.SMAP_C))))   # (256 -> 256: conv_smap's images)
              or (tuple(w.shape[2:]) == (4, 4) and ops.WINO4 and w.shape[1] % 16 == 0)                                                              # (conv_wino4: forward-form image; the 3-channel ends of the U-Net stay direct)
              or (tuple(w.shape[2:]) == (1, 1) and ops.PW and tuple(w.shape[:2]) in ((256, 64), (64, 256))))]                                       # (conv_pw: forward of 64 -> 256, input gradient of 256 -> 64)
        if not ws:
            return
        bank = getattr(self, '_wino_bank', None)
        if bank is None or not bank.matches(ws):
            if bank is not None:
                bank.release()
                self._wino_bank = None
            if torch.cuda.is_current_stream_capturing():   # (building allocates and copies tables: not inside a capture -
                return                                     #  without images these convs take the direct kernels)
            bank = self._wino_bank = ops.WinoBank(ws)
        if getattr(self, 'wino_static', False) and not self.training:
            # a FROZEN eval-mode network (the AdvMix teacher: core.function marks it): its filters change only through torch
            # (load_state_dict, param.copy_ / mul_ ...) - which bumps the parameters' version counters - so the images are re-made
            # only then; writes torch does not see (param.data, collectives, raw kernels): invalidate_filter_images()
            key = tuple((w._version, w.data_ptr()) for w in ws)      # (re-pointed storage - FlatAdam, .to() - counts as a change too)
            if getattr(bank, 'static_key', None) == key:
                return
            bank.static_key = key
        else:
            bank.static_key = None
        first = self._wino_first_level or 0
        if ops.WINO_ASYNC and first > 0:
            # the ~125 us of filter transforms run on a launch-lane stream BESIDE the levels that need no image (the stem);
            # PlanRun joins it before the first level with such a conv.  (The side stream first waits for the caller's:
            # the optimizer step that wrote the filters, and every earlier reader of the images, are behind it.)
            cur = torch.cuda.current_stream(device)
            side = ops._lanes(device, 1)[0]
            side.wait_stream(cur)
            bank.refresh(ops.ctypes.c_void_p(side.cuda_stream))
            self._wino_pending = (side, first)
        else:
            bank.refresh()

    def begin(self, x):
        """Start a level-by-level execution (see PlanRun)."""
        return PlanRun(self, x)

    def forward(self, x):
        run = self.begin(x)
        while not run.done:
            run.consume(ops.run_group(run.members()))
        return run.result


class PlanRun:
    """One in-flight forward of a PlanNet, advanced one level (= one concurrent launch group)
    at a time."""

    def __init__(self, net, x):
        self.net, self.T, self.train = net, net._tensors(), net.training
        self.slots = [None] * len(net.plan.ch)
        self.slots[0] = x
        self.li = 0
        net.last_cuts = []
        if self.train and torch.is_tensor(x) and x.is_cuda:
            net._arena.begin_pass(x.device)                # zero the statistics slots once, before any lane forks
        if torch.is_tensor(x) and x.is_cuda:
            net._wino_refresh(x.device)                    # transformed filters of the Winograd convs, one launch

    @property
    def done(self):
        return self.li >= len(self.net._levels)

    def members(self):
        pend = getattr(self.net, '_wino_pending', None)
        if pend is not None and self.li >= pend[1]:         # the filter images are needed from this level on (see _wino_refresh)
            torch.cuda.current_stream(pend[0].device).wait_stream(pend[0])
            self.net._wino_pending = None
        return [self.net._member(st, self.slots, self.T, self.train) for st in self.net._levels[self.li]]

    def consume(self, outs):
        net, sts = self.net, self.net._levels[self.li]
        for st, o in zip(sts, outs):
            if st[0] == 'chain':
                for d, t in zip(st[3], o):
                    self.slots[d] = t
            else:
                self.slots[net._dst(st)] = o
                if ops.SLOT_TAP is not None:
                    ops.SLOT_TAP(net, net._dst(st), o, None)
        for st in sts:                                     # drop dead activations early
            for s in net._srcs(st):
                if net._last_use[s] == self.li and s != net.plan.out:
                    self.slots[s] = None
        if self.li in net.cut_levels:                      # cut the autograd graph at this level boundary
            pairs = []
            for s_, t in enumerate(self.slots):
                if t is not None and s_ != 0 and net._last_use.get(s_, -1) > self.li and torch.is_tensor(t) \
                        and t.requires_grad:
                    twin = t.detach().requires_grad_(True)
                    self.slots[s_] = twin
                    pairs.append((t, twin))
            net.last_cuts.append(pairs)
        self.li += 1

    @property
    def result(self):
        return self.slots[self.net.plan.out]
