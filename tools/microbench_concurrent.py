#!/usr/bin/env python3
"""Aggregate throughput of the dominant conv when 1 / 2 / 4 HIP streams launch it concurrently
(what the launch lanes do), and of the BatchNorm-backward trio beside it.
usage: microbench_concurrent.py [B]"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call, lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
H, W, C = 64, 48, 32
rows = B * H * W


def make():
    x = torch.randn(B, H, W, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    y = torch.empty(B, H, W, C, device=dev)
    dy = torch.randn(rows, C, device=dev); yy = torch.relu(torch.randn(rows, C, device=dev))
    mean = torch.zeros(C, device=dev); invstd = torch.ones(C, device=dev); gamma = torch.ones(C, device=dev)
    dx = torch.empty(rows, C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    ws = torch.zeros(lib.advmix_norm_ws_bytes(1, C) // 4 + 16, device=dev)
    return dict(x=x, w=w, y=y, dy=dy, yy=yy, mean=mean, invstd=invstd, gamma=gamma, dx=dx, dg=dg, db=db, ws=ws)


def conv(t, st):
    call('advmix_conv_fwd', P(t['x']), P(t['w']), None, P(t['y']), B, H, W, C, H, W, C, 3, 3, 1, 1, st)


def bnbwd(t, st):
    call('advmix_norm_bwd', P(t['dy']), P(t['yy']), C, P(t['x']), P(t['mean']), P(t['invstd']), P(t['gamma']),
         P(t['dx']), None, P(t['dg']), P(t['db']), 1, rows, C, 1, P(t['ws']), st)


def run(kinds, iters=200):
    streams = [torch.cuda.Stream() for _ in kinds]
    bufs = [make() for _ in kinds]
    hs = [ctypes.c_void_p(s.cuda_stream) for s in streams]
    for k, t, h in zip(kinds, bufs, hs):
        for _ in range(5):
            k(t, h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for k, t, h in zip(kinds, bufs, hs):
            k(t, h)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


fl = 2.0 * rows * C * C * 9
only_conv = len(sys.argv) > 2 and sys.argv[2] == 'conv'
for n in ((1, 4) if only_conv else (1, 2, 3, 4, 6)):
    us = run([conv] * n)
    print('%d streams x conv 3x3 32->32 @64x48 B%d: %.1f us per round -> %.1f TFLOP/s aggregate (%.1f%% of 157.3)' % (
        n, B, us, n * fl / us / 1e6, n * fl / us / 1e6 / 157.3 * 100), flush=True)
if only_conv:
    sys.exit(0)
for n in (1, 2, 4):
    us = run([bnbwd] * n)
    print('%d streams x BN backward (3 kernels, rows %d C %d): %.1f us per round (%.1f us each)' % (n, rows, C, us, us / n), flush=True)
us = run([conv, conv, bnbwd, bnbwd])
print('2 conv + 2 BN-backward streams: %.1f us per round' % us, flush=True)
us = run([conv, conv, conv, bnbwd])
print('3 conv + 1 BN-backward streams: %.1f us per round' % us, flush=True)
