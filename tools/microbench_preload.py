#!/usr/bin/env python3
"""What BatchNorm + ReLU applied to the conv's A fragments ON LOAD costs (DESIGN.md section 8, item 2 b): needs the
measurement build  tools/build_variant.sh pre conv_direct -DCD_PRELOAD  and ADVMIX_SO=tools/_dbg/libpre.so.
conv'(c1; scale, shift) against conv(relu(c1 * scale + shift)): same result, launch-to-launch time with and without."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call, lib

dev = torch.device('cuda:0')
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.advmix_dbg_preload.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
B = 32


def timeit(run, iters=300):
    for _ in range(20):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters):
            run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for (C, H, W) in ((32, 64, 48), (64, 32, 24), (128, 16, 12), (256, 8, 6)):
    c1 = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
    y1 = torch.relu(c1 * sc + sh)
    ya, yb = torch.empty_like(c1), torch.empty_like(c1)
    slots = torch.zeros(2 * C * 64, device=dev, dtype=torch.float64)
    nbg = ctypes.c_int(0)
    geom = (B, H, W, C, H, W, C, 3, 3, 1, 1)

    def conv(x, y):
        nbg.value = 0
        call('advmix_conv_fwd_ex', P(x), P(w), None, P(y), *geom, None, None, None, None, 0.0, None, 0, P(slots),
             ctypes.byref(nbg), st)
    lib.advmix_dbg_preload(None, None)
    conv(y1, ya)
    t_plain = timeit(lambda: conv(y1, ya))
    lib.advmix_dbg_preload(P(sc), P(sh))
    conv(c1, yb)
    t_pre = timeit(lambda: conv(c1, yb))
    lib.advmix_dbg_preload(None, None)
    torch.cuda.synchronize()
    err = float((ya - yb).abs().max()) / float(ya.abs().max())
    print('3x3 %3d->%-3d @%2dx%-2d: conv+sums %.1f us, with BN+ReLU on load %.1f us (+%.1f); max rel diff %.1e' % (
        C, C, H, W, t_plain, t_pre, t_pre - t_plain, err), flush=True)
