#!/usr/bin/env python3
"""The small weight gradients of HRNet-W32's fuse layers / transitions (B = 32): one advmix_conv_wgrad launch each against
advmix_conv_wgrad_multi launches of up to 16 mixed geometries, launch to launch through the C ABI.
usage: microbench_wgrad_multi.py [B=32] [iters=50]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call, lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device('cuda:0')
# (Ci, H, W, Co, k, s, p): one stage-4 fuse layer's convs (pose_hrnet.py:172-247) + a transition
PROBS = [(32, 64, 48, 32, 3, 2, 1), (32, 64, 48, 64, 3, 2, 1), (32, 32, 24, 32, 3, 2, 1), (32, 32, 24, 128, 3, 2, 1),
         (32, 16, 12, 256, 3, 2, 1), (64, 32, 24, 128, 3, 2, 1), (64, 32, 24, 64, 3, 2, 1), (64, 16, 12, 256, 3, 2, 1),
         (128, 16, 12, 256, 3, 2, 1), (64, 32, 24, 32, 1, 1, 0), (128, 16, 12, 32, 1, 1, 0), (128, 16, 12, 64, 1, 1, 0),
         (256, 8, 6, 32, 1, 1, 0), (256, 8, 6, 64, 1, 1, 0), (256, 8, 6, 128, 1, 1, 0), (128, 16, 12, 256, 3, 2, 1)]
xs, dys, dws, geoms = [], [], [], []
for Ci, H, W, Co, k, s, p in PROBS:
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    xs.append(torch.randn(B, H, W, Ci, device=dev)); dys.append(torch.randn(B, Ho, Wo, Co, device=dev))
    dws.append(torch.zeros(Co, k, k, Ci, device=dev)); geoms.append((B, Ho, Wo, Co, H, W, Ci, k, k, s, p))
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
n = len(PROBS)
arr = ctypes.c_void_p * n
A, Bp, D = arr(*[t.data_ptr() for t in dys]), arr(*[t.data_ptr() for t in xs]), arr(*[t.data_ptr() for t in dws])
G = (ctypes.c_int * (11 * n))(*[v for g in geoms for v in g])


def timed(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def singles():
    for i in range(n):
        call('advmix_conv_wgrad', P(dys[i]), P(xs[i]), P(dws[i]), *geoms[i], st)


def multi():
    rc = lib.advmix_conv_wgrad_multi(n, A, Bp, D, G, st)
    assert rc == 0, rc


fl = sum(2.0 * g[0] * g[1] * g[2] * g[3] * g[6] * g[7] * g[8] for g in geoms)
t1, t2 = timed(singles), timed(multi)
print('%d small weight gradients, B = %d, %.2f GFLOP: one launch each %.1f us (%.1f TFLOP/s); one mixed launch %.1f us (%.1f TFLOP/s)  x%.2f'
      % (n, B, fl / 1e9, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6, t1 / t2))
for i in range(n):
    t = timed(lambda: call('advmix_conv_wgrad', P(dys[i]), P(xs[i]), P(dws[i]), *geoms[i], st))
    print('   %-28s alone %.1f us' % ('%dx%d s%d %d->%d @%dx%d' % (PROBS[i][4], PROBS[i][4], PROBS[i][5], PROBS[i][0], PROBS[i][3],
                                                                  geoms[i][1], geoms[i][2]), t))
