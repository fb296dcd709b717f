#!/usr/bin/env python3
"""Wall time of the AdvMix step's phases, each captured as its own HIP graph and replayed (so launch
overhead is out and the lanes overlap as in the real step): where do the 66 ms go?"""
import sys, os, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from advmix_amd import ops
from advmix_amd.core.function import set_require_grad

dev = torch.device('cuda:0')
cfg, D, G, T, crit, optD, optG = bench.build_models(sys.argv[1] if len(sys.argv) > 1 else 'hrnet_w32', dev)
views, tgt, tw = bench.synth(32, 17, bench.WORKLOADS['hrnet_w32'][3], bench.WORKLOADS['hrnet_w32'][4], dev, 1234)
x = views[0]


def timed(name, fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print('%-46s %7.2f ms' % (name, ms), flush=True)
    return ms


def g_fwd():
    with torch.no_grad():
        return ops.softmax_mix(G(ops.cat_views(views)), views)


def d_fwd():
    with torch.no_grad():
        return D(x)


def t_fwd():
    with torch.no_grad():
        return T(x)


def d_fwd_bwd(frozen):
    def f():
        set_require_grad(D, not frozen)
        xin = x.detach().requires_grad_(frozen)
        optD.zero_grad()
        crit(D(xin), tgt, tw).backward()
    return f


def g_fwd_bwd():
    optG.zero_grad()
    out = ops.softmax_mix(G(ops.cat_views(views)), views)
    out.backward(torch.ones_like(out))


D.train(); G.train(); T.eval()
a = timed('G forward + mix (1 chain)', g_fwd)
b = timed('D forward, train-mode BN (no autograd state)', d_fwd)
c = timed('teacher forward, eval (conv+BN+ReLU fused)', t_fwd)
d = timed('D forward + backward (dgrad + wgrad)', d_fwd_bwd(False))
e = timed('D forward + backward, frozen (dgrad only)', d_fwd_bwd(True))
f = timed('G forward + mix + backward', g_fwd_bwd)
print('sum of the step\'s parts: G fwd %.1f + D fwd/bwd %.1f + T %.1f + frozen D fwd/bwd %.1f + G bwd %.1f = %.1f ms'
      % (a, d, c, e, f - a, a + d + c + e + f - a))
