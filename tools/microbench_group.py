#!/usr/bin/env python3
"""One grouped launch (advmix_conv_group) against the same problems launched one by one on one stream and on four
streams: HRNet-W32's four branch convs (3x3, C = 32 / 64 / 128 / 256 at 64x48 / 32x24 / 16x12 / 8x6), B = 32.
usage: microbench_group.py [fwd_stats|fwd_eval|dgrad_add|dgrad_bnb] [branches=4] [B=32]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call, lib, ConvProblem

mode = sys.argv[1] if len(sys.argv) > 1 else 'fwd_stats'
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device('cuda:0')
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
A = lambda t: 0 if t is None else t.data_ptr()
shapes = [(32, 64, 48), (64, 32, 24), (128, 16, 12), (256, 8, 6)][:nb]
T = []
for (C, H, W) in shapes:
    t = dict(C=C, H=H, W=W, x=torch.randn(B, H, W, C, device=dev), w=torch.randn(C, 3, 3, C, device=dev) * 0.05,
             y=torch.empty(B, H, W, C, device=dev), y2=torch.empty(B, H, W, C, device=dev), res=torch.randn(B, H, W, C, device=dev),
             yy=torch.randn(B, H, W, C, device=dev), mk=torch.randint(0, 16, (B * H * W * C // 4,), device=dev, dtype=torch.uint8), cc=torch.randn(B, H, W, C, device=dev),
             g=torch.rand(C, device=dev) + 0.5, b=torch.randn(C, device=dev), rm=torch.randn(C, device=dev) * 0.1,
             rv=torch.rand(C, device=dev) + 0.5, mean=torch.zeros(C, device=dev), invstd=torch.ones(C, device=dev),
             slots=torch.zeros(2 * C * 64, device=dev, dtype=torch.float64), slots2=torch.zeros(2 * C * 64, device=dev, dtype=torch.float64))
    T.append(t)
flops = sum(2.0 * B * t['H'] * t['W'] * t['C'] * t['C'] * 9 for t in T)
streams = [torch.cuda.Stream() for _ in T]


def single(t, st, out='y', slots='slots'):
    C, H, W = t['C'], t['H'], t['W']
    geom = (B, H, W, C, H, W, C, 3, 3, 1, 1)
    nbg = ctypes.c_int(0)
    if mode == 'fwd_stats':
        call('advmix_conv_fwd_ex', P(t['x']), P(t['w']), None, P(t[out]), *geom, None, None, None, None, 0.0, None, 0,
             P(t[slots]), ctypes.byref(nbg), st)
    elif mode == 'fwd_eval':
        call('advmix_conv_fwd_ex', P(t['x']), P(t['w']), None, P(t[out]), *geom, P(t['g']), P(t['b']), P(t['rm']), P(t['rv']),
             1e-5, P(t['res']), 1, None, None, st)
    elif mode == 'dgrad_add':
        call('advmix_conv_tr_w_add', P(t['x']), P(t['w']), P(t['res']), P(t[out]), *geom, st)
    else:
        call('advmix_conv_tr_w_bnb', P(t['x']), P(t['w']), P(t['res']), P(t[out]), *geom, P(t['mk']), P(t['cc']), P(t['mean']),
             P(t['invstd']), None, None, 1, P(t[slots]), ctypes.byref(nbg), st)
    return nbg.value


def problems(out='y', slots='slots'):
    arr = (ConvProblem * len(T))()
    for i, t in enumerate(T):
        C, H, W = t['C'], t['H'], t['W']
        q = arr[i]
        q.x, q.w, q.bias, q.y = A(t['x']), A(t['w']), 0, A(t[out])
        q.N, q.Hx, q.Wx, q.Cx, q.Hy, q.Wy, q.Cy, q.R, q.S, q.stride, q.pad = B, H, W, C, H, W, C, 3, 3, 1, 1
        if mode == 'fwd_stats':
            q.stats, q.stats_ns = A(t[slots]), 0
        elif mode == 'fwd_eval':
            q.bn_gamma, q.bn_beta, q.bn_rm, q.bn_rv, q.bn_eps, q.residual, q.act = A(t['g']), A(t['b']), A(t['rm']), A(t['rv']), 1e-5, A(t['res']), 1
        elif mode == 'dgrad_add':
            q.residual = A(t['res'])
        else:
            q.residual, q.stats, q.stats_ns = A(t['res']), A(t[slots]), 0
            q.bnb_mask, q.bnb_c, q.bnb_mean, q.bnb_invstd, q.bnb_act = A(t['mk']), A(t['cc']), A(t['mean']), A(t['invstd']), 1
    return arr


kind = 0 if mode.startswith('fwd') else 1
cur = torch.cuda.current_stream()
st0 = ctypes.c_void_p(cur.cuda_stream)

# ---- numerics: grouped == one by one -------------------------------------------------------------------
for t in T:
    single(t, st0)
arr = problems('y2', 'slots2')
rc = lib.advmix_conv_group(kind, len(T), arr, st0)
assert rc == 0, rc
torch.cuda.synchronize()
for i, t in enumerate(T):
    assert torch.equal(t['y'], t['y2']), 'branch %d differs' % i
    if mode in ('fwd_stats', 'dgrad_bnb'):
        ns = arr[i].stats_ns
        a = t['slots'][:2 * t['C'] * ns].view(2, ns, t['C']).sum(1)
        b = t['slots2'][:2 * t['C'] * ns].view(2, ns, t['C']).sum(1)
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-6), 'branch %d sums differ' % i
print('grouped launch == single launches (outputs bit-identical)')


def timeit(run, iters=200):
    for _ in range(20):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def seq():
    for t in T:
        single(t, st0)


def par():
    for t, s in zip(T, streams):
        s.wait_stream(cur)
        single(t, ctypes.c_void_p(s.cuda_stream))
    for s in streams:
        cur.wait_stream(s)


arr = problems()


def grp():
    for q in arr:
        q.stats_ns = 0
    lib.advmix_conv_group(kind, len(T), arr, st0)


for name, fn in (('one by one, one stream', seq), ('one by one, %d streams + joins' % len(T), par), ('one grouped launch', grp)):
    us = timeit(fn)
    print('%-34s %6.1f us  %5.1f TFLOP/s  %.3f of fp32 MFMA peak' % (name, us, flops / us / 1e6, flops / us / 1e6 / 157.3))
