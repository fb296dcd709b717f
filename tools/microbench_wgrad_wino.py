#!/usr/bin/env python3
"""Winograd weight gradient (csrc/wgrad_wino.hip) against the grouped direct kernels, eight problems of one geometry per
launch as the step issues them.  usage: microbench_wgrad_wino.py [B=32] [iters=50]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device('cuda:0')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(run):
    for _ in range(5):
        run()
    vals = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        vals.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(vals)[2]


for C, H, W in ((32, 64, 48), (64, 32, 24), (128, 16, 12), (64, 64, 48)):
    for n in (8, 1):
        xs = [torch.randn(B, H, W, C, device=dev) for _ in range(n)]
        dys = [torch.randn(B, H, W, C, device=dev) for _ in range(n)]
        dws = [torch.zeros(C, 3, 3, C, device=dev) for _ in range(n)]
        arr = ctypes.c_void_p * n
        a, b, g = arr(*[t.data_ptr() for t in dys]), arr(*[t.data_ptr() for t in xs]), arr(*[t.data_ptr() for t in dws])
        tw = timed(lambda: lib.advmix_conv3x3_wgrad_wino_group(n, a, b, g, B, H, W, C, C, st))
        if n > 1:
            td = timed(lambda: lib.advmix_conv_wgrad_group(n, a, b, g, B, H, W, C, H, W, C, 3, 3, 1, 1, st))
        else:
            td = timed(lambda: lib.advmix_conv_wgrad(a[0], b[0], g[0], B, H, W, C, H, W, C, 3, 3, 1, 1, st))
        fl = 2.0 * n * B * H * W * C * C * 9
        print('wgrad 3x3 %d->%d @%dx%d B=%d x%d: wino %7.1f us (%.1f per problem, %.3f of peak on direct FLOPs)   direct %7.1f us (%.1f, %.3f)   x%.2f' % (
            C, C, H, W, B, n, tw, tw / n, fl / tw / 1e6 / 157.3, td, td / n, fl / td / 1e6 / 157.3, td / tw))
