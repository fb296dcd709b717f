#!/usr/bin/env python3
"""The streaming 1x1 kernel (csrc/conv_pw.hip: 64 -> 256 channels) against conv_direct, launch to launch through the C ABI, for
the roles the step uses.  usage: microbench_pw.py [B=32] [iters=200] [H=64] [W=48]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd import ops
from advmix_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
H = int(sys.argv[3]) if len(sys.argv) > 3 else 64
W = int(sys.argv[4]) if len(sys.argv) > 4 else 48
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


x64, y256 = torch.randn(B, H, W, 64, device=dev), torch.empty(B, H, W, 256, device=dev)
res, cc = torch.randn_like(y256), torch.randn_like(y256)
w_up = (torch.randn(256, 1, 1, 64, device=dev) * 0.1).permute(0, 3, 1, 2)        # 64 -> 256 (forward side)
w_dn = (torch.randn(64, 1, 1, 256, device=dev) * 0.1).permute(0, 3, 1, 2)        # 256 -> 64 (its input gradient writes 256 channels)
gam, bet, rm, rv = (torch.rand(256, device=dev) + 0.5 for _ in range(4))
mean, invstd = torch.zeros(256, device=dev), torch.ones(256, device=dev)
mk = torch.randint(0, 16, (y256.numel() // 4,), device=dev, dtype=torch.uint8)
slots = torch.zeros(2 * 256 * 64, device=dev, dtype=torch.float64)
bank = ops.WinoBank([w_up, w_dn])
bank.refresh()
uf = bank.images(w_up)[0]
ud = bank.images(w_dn)[1]
ns = ctypes.c_int(0)


def z():
    ns.value = 0


g_up = (B, H, W, 64, H, W, 256, 1, 1, 1, 0)
runs = {
    'fwd+sums': (lambda: (z(), lib.advmix_conv1x1_pw_fwd(P(x64), uf, P(y256), B, H, W, 64, 256, None, None, None, None, 0.0, None, 0, P(slots), ctypes.byref(ns), st)),
                 lambda: (z(), lib.advmix_conv_fwd_ex(P(x64), P(w_up), None, P(y256), *g_up, None, None, None, None, 0.0, None, 0, P(slots), ctypes.byref(ns), st))),
    'fwd+bn_eval+res+relu': (lambda: lib.advmix_conv1x1_pw_fwd(P(x64), uf, P(y256), B, H, W, 64, 256, P(gam), P(bet), P(rm), P(rv), 1e-5, P(res), 1, None, None, st),
                             lambda: lib.advmix_conv_fwd_ex(P(x64), P(w_up), None, P(y256), *g_up, P(gam), P(bet), P(rm), P(rv), 1e-5, P(res), 1, None, None, st)),
    'dgrad+addend+bnb(mask)': (lambda: (z(), lib.advmix_conv1x1_pw_dgrad(P(x64), ud, P(res), P(y256), B, H, W, 64, 256, P(mk), P(cc), P(mean), P(invstd), None, None, 1, P(slots), ctypes.byref(ns), st)),
                               lambda: (z(), lib.advmix_conv_tr_w_bnb(P(x64), P(w_dn), P(res), P(y256), B, H, W, 64, H, W, 256, 1, 1, 1, 0, P(mk), P(cc), P(mean), P(invstd), None, None, 1, P(slots), ctypes.byref(ns), st))),
    'dgrad+bnb(sign from c)': (lambda: (z(), lib.advmix_conv1x1_pw_dgrad(P(x64), ud, None, P(y256), B, H, W, 64, 256, None, P(cc), P(mean), P(invstd), P(gam), P(bet), 1, P(slots), ctypes.byref(ns), st)),
                               lambda: (z(), lib.advmix_conv_tr_w_bnb(P(x64), P(w_dn), None, P(y256), B, H, W, 64, H, W, 256, 1, 1, 1, 0, None, P(cc), P(mean), P(invstd), P(gam), P(bet), 1, P(slots), ctypes.byref(ns), st))),
}
rows = B * H * W
fl = 2.0 * rows * 64 * 256
print('1x1 64->256 @%dx%d B=%d  (%.1f MB per 256-channel tensor)' % (H, W, B, rows * 256 * 4 / 1e6))
for name, (new, direct) in runs.items():
    assert new()[-1] == 0 if isinstance(new(), tuple) else new() == 0, name
    tn, td = timed(new), timed(direct)
    nt = {'fwd+sums': 1.25, 'fwd+bn_eval+res+relu': 2.25, 'dgrad+addend+bnb(mask)': 3.31, 'dgrad+bnb(sign from c)': 2.25}[name]   # 256-channel tensors moved
    nbytes = nt * rows * 256 * 4
    print('  %-26s conv_pw %6.1f us (%.2f TB/s, %.3f of the fp32 matrix peak)   direct %6.1f us (%.2f TB/s)   x%.2f' % (
        name, tn, nbytes / tn / 1e6, fl / tn / 1e6 / 157.3, td, nbytes / td / 1e6, td / tn))
bank.release()
