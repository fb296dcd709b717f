#!/usr/bin/env python3
"""conv + train-mode BatchNorm at the HRNet-W32 branch shapes (B = 32): what the statistics cost.

forward : conv(+column sums in the epilogue) -> [finalize] -> apply          (old: 3 launches, new: 2)
backward: [statistics pass -> finalize ->] apply -> input-gradient conv       (old: 4 launches; new: the
          consumer's input-gradient conv carries the statistics epilogue, then ONE apply launch)
Prints launch-to-launch microseconds per sequence (HIP events on the launching stream).
usage: microbench_bn.py [B] [slots ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call, lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SLOTS = [int(v) for v in sys.argv[2:]] or [0, 16, 64]     # 0 = the kernel's own choice
dev = torch.device('cuda:0')
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(run, iters=50):
    for _ in range(5):
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


for (C, H, W) in ((32, 64, 48), (64, 32, 24), (128, 16, 12), (256, 8, 6)):
    rows = B * H * W
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    c = torch.empty(B, H, W, C, device=dev)
    y = torch.empty_like(c)
    res = torch.randn_like(c)
    amask = torch.randint(0, 16, (c.numel() // 4,), device=c.device, dtype=torch.uint8)     # the activation bit mask
    dy = torch.randn_like(c)
    dc = torch.empty_like(c)
    dx = torch.empty_like(c)
    g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws = torch.zeros(int(lib.advmix_norm_ws_bytes(1, C)) // 4 + 16, device=dev)
    geom = (B, H, W, C, H, W, C, 3, 3, 1, 1)
    flops = 2.0 * rows * C * C * 9
    out = ['C%-3d %2dx%-2d rows %6d | conv alone %.1f us' % (C, H, W, rows, timeit(
        lambda: call('advmix_conv_fwd', P(x), P(w), None, P(c), *geom, st)))]
    for ns in SLOTS:
        slots = torch.zeros(2 * C * 64, device=dev, dtype=torch.float64)
        nbg = ctypes.c_int(ns)

        def fwd_old():
            nbg.value = ns
            call('advmix_conv_fwd_ex', P(x), P(w), None, P(c), *geom, None, None, None, None, 0.0, None, 0, P(slots),
                 ctypes.byref(nbg), st)
            call('advmix_norm_finalize', P(slots), nbg.value, rows, C, 1e-5, P(mean), P(invstd), P(rm), P(rv), P(nbt), 0.1, P(amask), st)
            call('advmix_norm_apply', P(c), P(mean), P(invstd), P(g), P(b), P(res), P(y), C, 1, rows, C, 1, st)

        def fwd_new():
            nbg.value = ns
            call('advmix_conv_fwd_ex', P(x), P(w), None, P(c), *geom, None, None, None, None, 0.0, None, 0, P(slots),
                 ctypes.byref(nbg), st)
            call('advmix_norm_apply_slots', P(c), P(slots), nbg.value, rows, C, 1e-5, P(g), P(b), P(res), P(y), 1,
                 P(mean), P(invstd), P(rm), P(rv), P(nbt), 0.1, P(amask), st)

        def conv_stats():
            nbg.value = ns
            call('advmix_conv_fwd_ex', P(x), P(w), None, P(c), *geom, None, None, None, None, 0.0, None, 0, P(slots),
                 ctypes.byref(nbg), st)

        def bwd_old():
            call('advmix_norm_bwd', P(dy), P(y), C, P(c), P(mean), P(invstd), P(g), P(dc), None, P(dg), P(db), 1, rows, C, 1,
                 P(ws), st)
            call('advmix_conv_tr_w_add', P(dc), P(w), P(res), P(dx), *geom, st)

        conv_stats()
        nsu = nbg.value                                       # slots actually used

        def bwd_new():
            nbg.value = ns
            call('advmix_norm_bwd_apply_slots', P(dy), P(c), P(mean), P(invstd), P(g), P(slots), nsu, rows, C, P(dc), P(dg),
                 P(db), st)
            call('advmix_conv_tr_w_bnb', P(dc), P(w), P(res), P(dx), *geom, P(amask), P(c), P(mean), P(invstd), None, None, 1, P(slots),
                 ctypes.byref(nbg), st)
        out.append('ns %2d(%2d): conv+sums %.1f | fwd old %.1f new %.1f | bwd old %.1f new %.1f' % (
            ns, nsu, timeit(conv_stats), timeit(fwd_old), timeit(fwd_new), timeit(bwd_old), timeit(bwd_new)))
    print('\n   '.join(out), flush=True)
    apply_bytes = rows * C * 4 * 3
    t = timeit(lambda: call('advmix_norm_apply', P(c), P(mean), P(invstd), P(g), P(b), P(res), P(y), C, 1, rows, C, 1, st))
    t2 = timeit(lambda: call('advmix_conv_tr_w_add', P(dc), P(w), P(res), P(dx), *geom, st))
    print('   apply alone %.1f us (%.2f TB/s) | dgrad+addend alone %.1f us (%.1f TFLOP/s)' % (
        t, apply_bytes / t / 1e6, t2, flops / t2 / 1e6), flush=True)
