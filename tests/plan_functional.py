"""A plan.Plan read functionally on torch CPU (any dtype): the truth of the pinned-mask parity tests.

``interpret(P, W, x)`` evaluates the plan's steps with train-mode BatchNorm; each activation uses its OWN mask, or - ``pin`` -
the mask of another evaluation of the same network (the device's: ``SlotRecorder``), which makes the function the same
piecewise-linear piece the device computed on, so that gradients can be compared element by element (profiles/EXPERIMENTS.md
K2: un-pinned, one pre-activation within rounding of zero flips a ReLU mask in ANY fp32 evaluation and moves whole gradient
tensors by per cent).  tests/test_plan_interpreter_cpu.py holds this reading to the oracle's forward passes to 1e-12.
Test infrastructure: nothing under advmix_amd/ imports it."""
import torch
import torch.nn.functional as F

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def activated_slots(P):
    """[(slot, activation)] of every slot of Plan ``P`` that an activation produces, in step order."""
    out = []
    for st in P.steps:
        k = st[0]
        hit = {'bn': lambda: (st[3], st[5]), 'fuse': lambda: (st[3], st[4]), 'inorm': lambda: (st[2], st[3]),
               'act': lambda: (st[2], st[3]), 'catact': lambda: (st[3], st[4])}.get(k)
        if hit is not None and hit()[1] != ACT_NONE:
            out.append(hit())
    return out


def pooled_slots(P):
    """[(source slot, destination slot)] of the plan's max-pool steps."""
    return [(st[1], st[2]) for st in P.steps if st[0] == 'maxpool']


def interpret(P, W, x, pin=None, pool_src=None):
    """{slot: value} of the steps of ``P`` in x's dtype.  ``pin``: {activated slot: that slot's value in the evaluation whose
    masks are adopted} (sign > 0 = the pass-through side of ReLU / LeakyReLU(0.2)); ``pool_src``: {max-pool destination slot:
    the INPUT of that pool in the adopted evaluation} - the window element that wins there is the one gathered here."""
    def act(t, a, d_):
        if a == ACT_NONE:
            return t
        if pin is None:
            return F.relu(t) if a == ACT_RELU else F.leaky_relu(t, 0.2)
        on = pin[d_] > 0
        return t * on.to(t.dtype) if a == ACT_RELU else torch.where(on, t, t * 0.2)         # (LeakyReLU(0.2): Unet_generator.py:42)
    val = {0: x}
    for st in P.steps:
        k = st[0]
        if k in ('conv', 'deconv'):
            _, name, s_, d_, stride, pad, hb = st
            val[d_] = (F.conv2d if k == 'conv' else F.conv_transpose2d)(val[s_], W[name + '.weight'], W[name + '.bias'] if hb else None,
                                                                         stride, pad)
        elif k == 'bn':
            _, name, s_, d_, res, a = st
            pre = F.batch_norm(val[s_], None, None, W[name + '.weight'], W[name + '.bias'], True, 0.1, 1e-5)
            val[d_] = act(pre if res is None else pre + val[res], a, d_)
        elif k == 'inorm':
            val[st[2]] = act(F.instance_norm(val[st[1]], eps=1e-5), st[3], st[2])
        elif k == 'act':
            val[st[2]] = act(val[st[1]], st[3], st[2])
        elif k == 'catact':
            val[st[3]] = act(torch.cat([val[st[1]], val[st[2]]], 1), st[4], st[3])
        elif k == 'fuse':
            _, xs, shifts, d_, a = st
            val[d_] = act(sum(val[s_] if sh == 0 else F.interpolate(val[s_], scale_factor=2 ** sh, mode='nearest')
                              for s_, sh in zip(xs, shifts)), a, d_)
        elif k == 'maxpool':
            src = val[st[1]]
            if pool_src is None:
                val[st[2]] = F.max_pool2d(src, 3, 2, 1)
            else:                                           # the adopted evaluation's winners (pose_resnet.py:121: MaxPool2d(3, 2, 1))
                _, idx = F.max_pool2d(pool_src[st[2]], 3, 2, 1, return_indices=True)
                val[st[2]] = src.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
        else:
            raise ValueError(k)
    return val


class SlotRecorder:
    """Context manager around ops.SLOT_TAP: records, on the CPU, every slot the plan steps of the watched networks produce
    while it is active - from the forward passes the PRODUCT itself runs (launch chains, lanes and all), not from a re-run.
    ``of(net)`` -> [{slot: value}] one dict per forward pass of that network, in order."""

    def __init__(self):
        self.events = []

    def __enter__(self):
        from advmix_amd import ops
        assert ops.SLOT_TAP is None
        ops.SLOT_TAP = self._tap
        return self

    def __exit__(self, *exc):
        from advmix_amd import ops
        ops.SLOT_TAP = None

    def _tap(self, owner, slot, tensor, stream):
        torch.cuda.synchronize()                            # the producing kernel ran on a launch lane's stream
        self.events.append((id(owner), slot, tensor.detach().cpu().contiguous().clone()))

    def of(self, net):
        owners = {id(net)} | {id(v[0][0]) for v in net._chain_meta.values()}
        passes = [{}]
        for o, slot, val in self.events:
            if o not in owners:
                continue
            if slot in passes[-1]:                          # the same slot again: the next forward pass of this network
                passes.append({})
            passes[-1][slot] = val
        return [p for p in passes if p]


def pins_of(P, slots):
    """(pin, pool_src) for ``interpret`` from one recorded forward pass {slot: value}."""
    pin = {s_: slots[s_] for s_, _ in activated_slots(P)}
    pool = {dst: slots[src] for src, dst in pooled_slots(P)}
    return pin, pool
