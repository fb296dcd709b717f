"""The U-Net generator's non-conv kernels at its three largest activations (B = 32): InstanceNorm forward (statistics + apply),
InstanceNorm backward (partial sums + finalize + apply), activation copy / backward, bias gradient - us per call and the HBM
rate over the passes each one makes.  These run on ONE lane with nothing beside them (the generator is a sequential net)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call, lib

d = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr() if t is not None else None


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, C, H, W) in [(32, 64, 128, 96), (32, 128, 64, 48), (32, 256, 32, 24), (32, 512, 16, 12)]:
    HW, rows = H * W, B * H * W
    x = torch.randn(rows, C, device=d); dy = torch.randn(rows, C, device=d); y = torch.empty_like(x); dx = torch.empty_like(x)
    mean = torch.empty(B * C, device=d); invstd = torch.empty(B * C, device=d)
    ws = torch.zeros(lib.advmix_norm_ws_bytes(B, C) // 4 + 16, device=d)
    db = torch.zeros(C, device=d)
    mb = rows * C * 4 / 1e6
    t = {}
    t['in stats (1 pass)'] = (timeit(lambda: call('advmix_norm_stats', p(x), B, HW, C, 1e-5, p(mean), p(invstd), None, None, None, 0.0, p(ws), st)), 1)
    t['in apply (2)'] = (timeit(lambda: call('advmix_norm_apply', p(x), p(mean), p(invstd), None, None, None, p(y), C, B, HW, C, 2, st)), 2)
    t['in bwd (7)'] = (timeit(lambda: call('advmix_norm_bwd', p(dy), p(y), C, p(x), p(mean), p(invstd), None, p(dx), None, None, None, B, HW, C, 2, p(ws), st)), 7)
    t['act copy (2)'] = (timeit(lambda: call('advmix_act_copy', p(x), C, p(y), C, rows, C, 2, st)), 2)
    t['act bwd (3)'] = (timeit(lambda: call('advmix_act_bwd', p(dy), C, p(y), C, p(dx), C, rows, C, 2, st)), 3)
    t['bias grad (1)'] = (timeit(lambda: call('advmix_bias_grad', p(dy), p(db), rows, C, st)), 1)
    print('B %d C %d @%dx%d (%.0f MB per tensor): ' % (B, C, H, W, mb) + '; '.join('%s %.1f us %.1f TB/s' % (k, v[0], mb * v[1] / v[0]) for k, v in t.items()), flush=True)
