#!/usr/bin/env python3
"""Winograd F(2x2,3x3) kernel (csrc/conv_wino.hip) against the direct kernel, launch to launch through the C ABI, for the
roles the step uses.  usage: microbench_wino.py [B=32] [iters=200] [only=C:role, e.g. 32:fwd+sums - 20 launches of that one Winograd variant, for a counter pass]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd import ops
from advmix_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
only = sys.argv[3].split(':') if len(sys.argv) > 3 else None
dev = torch.device('cuda:0')
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(run):
    for _ in range(20):
        run()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(best)[len(best) // 2]


for C, H, W in ((32, 64, 48), (64, 32, 24), (128, 16, 12), (64, 64, 48), (256, 8, 6)):     # (256 @8x6: csrc/conv_smap.hip)
    if only and int(only[0]) != C:
        continue
    x = torch.randn(B, H, W, C, device=dev)
    y = torch.empty(B, H, W, C, device=dev)
    w = (torch.randn(C, 3, 3, C, device=dev) * 0.05).permute(0, 3, 1, 2)
    res, cc = torch.randn_like(x), torch.randn_like(x)
    gam, bet, rm, rv = (torch.rand(C, device=dev) + 0.5 for _ in range(4))
    mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mk = torch.randint(0, 16, (x.numel() // 4,), device=dev, dtype=torch.uint8)
    slots = torch.zeros(2 * C * 64, device=dev, dtype=torch.float64)
    bank = ops.WinoBank([w])
    bank.refresh()
    uf, ud = bank.images(w)
    geom = (B, H, W, C, H, W, C, 3, 3, 1, 1)
    ns = ctypes.c_int(0)
    kname = ('smapw' if ops.SMAP_WINO else 'smap') if C == 256 else 'wino'       # (ADVMIX_SMAP_WINO=0: the direct form of the image-per-workgroup kernel)
    k_fwd, k_dg = getattr(lib, 'advmix_conv3x3_%s_fwd' % kname), getattr(lib, 'advmix_conv3x3_%s_dgrad' % kname)

    def z():
        ns.value = 0
    runs = {
        'fwd+sums': (lambda: (z(), k_fwd(P(x), uf, P(y), B, H, W, C, C, None, None, None, None, 0.0, None, 0, P(slots), ctypes.byref(ns), st)),
                     lambda: (z(), lib.advmix_conv_fwd_ex(P(x), P(w), None, P(y), *geom, None, None, None, None, 0.0, None, 0, P(slots), ctypes.byref(ns), st))),
        'fwd+bn_eval+res+relu': (lambda: k_fwd(P(x), uf, P(y), B, H, W, C, C, P(gam), P(bet), P(rm), P(rv), 1e-5, P(res), 1, None, None, st),
                                 lambda: lib.advmix_conv_fwd_ex(P(x), P(w), None, P(y), *geom, P(gam), P(bet), P(rm), P(rv), 1e-5, P(res), 1, None, None, st)),
        'dgrad+addend+bnb(mask)': (lambda: (z(), k_dg(P(x), ud, P(res), P(y), B, H, W, C, C, P(mk), P(cc), P(mean), P(invstd), None, None, 1, P(slots), ctypes.byref(ns), st)),
                                   lambda: (z(), lib.advmix_conv_tr_w_bnb(P(x), P(w), P(res), P(y), B, H, W, C, H, W, C, 3, 3, 1, 1, P(mk), P(cc), P(mean), P(invstd), None, None, 1, P(slots), ctypes.byref(ns), st))),
        'dgrad+bnb(sign from c)': (lambda: (z(), k_dg(P(x), ud, None, P(y), B, H, W, C, C, None, P(cc), P(mean), P(invstd), P(gam), P(bet), 1, P(slots), ctypes.byref(ns), st)),
                                   lambda: (z(), lib.advmix_conv_tr_w_bnb(P(x), P(w), None, P(y), B, H, W, C, H, W, C, 3, 3, 1, 1, None, P(cc), P(mean), P(invstd), P(gam), P(bet), 1, P(slots), ctypes.byref(ns), st))),
    }
    if only:
        for _ in range(20):
            runs[only[1]][0]()
        torch.cuda.synchronize()
        print('ran 20 launches of', C, only[1])
        break
    fl = 2.0 * B * H * W * C * C * 9
    t_tr = timed(lambda: bank.refresh(st))
    print('3x3 %d->%d @%dx%d B=%d   (weight transform launch, 2 images: %.1f us)' % (C, C, H, W, B, t_tr))
    for name, (wino, direct) in runs.items():
        tw, td = timed(wino), timed(direct)
        print('  %-26s %s %6.1f us (%.3f of peak on direct FLOPs)   direct %6.1f us (%.3f)   x%.2f' % (
            name, kname, tw, fl / tw / 1e6 / 157.3, td, fl / td / 1e6 / 157.3, td / tw))
    bank.release()
