#!/usr/bin/env python3
"""advmix_norm_apply_slots (with and without residual) and advmix_norm_bwd_apply_slots against the 8 TB/s HBM peak over
the tensor sizes of the step (rows x C); bytes = the 2 or 3 full passes each kernel makes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call, lib
dev = torch.device('cuda:0')
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(run, iters=100):
    for _ in range(10): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
for (rows, C) in ((98304, 32), (98304, 64), (98304, 128), (98304, 256), (393216, 64), (24576, 512), (24576, 64), (6144, 128), (1536, 256)):
    c = torch.randn(rows, C, device=dev); y = torch.empty_like(c); res = torch.randn_like(c); amask = torch.zeros(rows * C // 4, device=dev, dtype=torch.uint8); g0 = torch.randn_like(c); dx = torch.empty_like(c)
    gm, bt = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ns = 16
    slots = torch.randn(2 * C * ns, device=dev, dtype=torch.float64).abs() * rows
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    t1 = timeit(lambda: call('advmix_norm_apply_slots', P(c), P(slots), ns, rows, C, 1e-5, P(gm), P(bt), None, P(y), 1, P(mean), P(invstd), None, None, None, 0.1, None, st))
    t2 = timeit(lambda: call('advmix_norm_apply_slots', P(c), P(slots), ns, rows, C, 1e-5, P(gm), P(bt), P(res), P(y), 1, P(mean), P(invstd), None, None, None, 0.1, P(amask), st))
    t3 = timeit(lambda: call('advmix_norm_bwd_apply_slots', P(g0), P(c), P(mean), P(invstd), P(gm), P(slots), ns, rows, C, P(dx), P(dg), P(db), st))
    mb = rows * C * 4 / 1e6
    print('rows %6d C %4d (%5.1f MB): apply %6.1f us %.2f TB/s | +res %6.1f us %.2f TB/s | bwd %6.1f us %.2f TB/s' % (rows, C, mb, t1, 2 * mb / t1, t2, 3 * mb / t2, t3, 3 * mb / t3))
