#!/usr/bin/env python3
"""Where does autograd add gradients itself (at::native add kernels on the caller's stream)?
Builds the headline step's two autograd graphs (D step, G step) eagerly and lists every output of a launch-group node that
has more than one consumer edge - the engine sums those gradients with torch kernels; inside a launch chain the sums are
the dgrad epilogues' addends / advmix_add.   python tools/fanin_report.py [workload] [B]"""
import collections
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_common import WORKLOADS, build_models, synth      # noqa: E402


def walk(root, label):
    edges = collections.Counter()
    shapes = {}
    seen, stack = set(), [root]
    while stack:
        fn = stack.pop()
        if fn is None or id(fn) in seen:
            continue
        seen.add(id(fn))
        for nxt, idx in fn.next_functions:
            if nxt is None:
                continue
            edges[(id(nxt), idx)] += 1
            shapes[(id(nxt), idx)] = (type(nxt).__name__, type(fn).__name__)
            stack.append(nxt)
    multi = {k: c for k, c in edges.items() if c > 1 and 'AccumulateGrad' not in shapes[k][0]}
    print('%s: %d nodes, %d outputs with more than one consumer' % (label, len(seen), len(multi)))
    for k, c in sorted(multi.items(), key=lambda kv: -kv[1]):
        print('   x%d  producer %s output %d (one consumer: %s)' % (c, shapes[k][0], k[1], shapes[k][1]))
    return multi


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'hrnet_w32'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    from advmix_amd import ops
    from advmix_amd.core.function import set_require_grad
    net, extra, J, H, W, downs, _ = WORKLOADS[wl]
    dev = torch.device('cuda:0')
    cfg, D, G, T, crit, optD, optG = build_models(wl, dev)
    views, tgt, tw = synth(B, J, H, W, dev, 7)
    # what a member's outputs are: wrap GroupFn.apply so that every output tensor remembers its member kinds / shape
    info = {}
    real = ops.GroupFn.apply

    def apply(spec, *flat):
        outs = real(spec, *flat)
        kinds = [getattr(op, '__name__', str(op)) for op, _c, _m in spec]
        for i, o in enumerate(outs if isinstance(outs, tuple) else (outs,)):
            if torch.is_tensor(o) and o.grad_fn is not None:
                info[(id(o.grad_fn), i)] = (kinds, tuple(o.shape))
        return outs
    ops.GroupFn.apply = apply
    gi = ops.cat_views([v.contiguous() for v in views])
    logits = G(gi)
    set_require_grad(D, True)
    tmp = ops.softmax_mix(logits, views)
    out = D(tmp.detach())
    with torch.no_grad():
        teach = T(views[0])
    loss_D = crit(out, tgt, tw) * 0.9 + crit(out, teach, tw) * 0.1
    for lab, root in (('D step', loss_D.grad_fn),):
        for k, c in walk(root, lab).items():
            print('      ', info.get(k))
    loss_D.backward()
    set_require_grad(D, False)
    out2 = D(tmp)
    loss_G = -crit(out2, tgt, tw)
    for k, c in walk(loss_G.grad_fn, 'G step').items():
        print('      ', info.get(k))
    loss_G.backward()
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
