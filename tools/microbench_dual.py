#!/usr/bin/env python3
"""Student (train mode: conv + BatchNorm column sums) and teacher (eval mode: conv + BN + residual + ReLU epilogue) of the SAME
layer as ONE two-problem launch (advmix_conv_group) against the two launches back to back: HRNet-W32's four branch convs, B = 32.
The only independent same-shape work inside an AdvMix step (function.py:146-149).  usage: microbench_dual.py [B=32]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call, lib, ConvProblem

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
A = lambda t: 0 if t is None else t.data_ptr()
cur = torch.cuda.current_stream()
st0 = ctypes.c_void_p(cur.cuda_stream)


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


for C, H, W in [(32, 64, 48), (64, 32, 24), (128, 16, 12), (256, 8, 6)]:
    mk = lambda: torch.randn(B, H, W, C, device=dev)
    xs, xt, ys, yt, ys2, yt2, res = mk(), mk(), mk(), mk(), mk(), mk(), mk()
    ws, wt = (torch.randn(C, 3, 3, C, device=dev) * 0.05 for _ in range(2))
    g, b, rm = (torch.randn(C, device=dev) for _ in range(3))
    rv = torch.rand(C, device=dev) + 0.5
    slots, slots2 = (torch.zeros(2 * C * 64, device=dev, dtype=torch.float64) for _ in range(2))
    geom = (B, H, W, C, H, W, C, 3, 3, 1, 1)
    nbg = ctypes.c_int(0)

    def two(o_s=ys, o_t=yt, sl=slots):
        nbg.value = 0
        call('advmix_conv_fwd_ex', P(xs), P(ws), None, P(o_s), *geom, None, None, None, None, 0.0, None, 0, P(sl), ctypes.byref(nbg), st0)
        call('advmix_conv_fwd_ex', P(xt), P(wt), None, P(o_t), *geom, P(g), P(b), P(rm), P(rv), 1e-5, P(res), 1, None, None, st0)
    arr = (ConvProblem * 2)()
    for q, (x, w, y) in zip(arr, ((xs, ws, ys2), (xt, wt, yt2))):
        q.x, q.w, q.bias, q.y = A(x), A(w), 0, A(y)
        q.N, q.Hx, q.Wx, q.Cx, q.Hy, q.Wy, q.Cy, q.R, q.S, q.stride, q.pad = B, H, W, C, H, W, C, 3, 3, 1, 1
    arr[0].stats, arr[0].stats_ns = A(slots2), 0
    t = arr[1]
    t.bn_gamma, t.bn_beta, t.bn_rm, t.bn_rv, t.bn_eps, t.residual, t.act = A(g), A(b), A(rm), A(rv), 1e-5, A(res), 1

    def grp():
        arr[0].stats_ns = 0
        return lib.advmix_conv_group(0, 2, arr, st0)
    two(); rc = grp(); torch.cuda.synchronize()
    assert rc == 0, rc
    assert torch.equal(ys, ys2) and torch.equal(yt, yt2), 'grouped != single'
    a, bb = timeit(two), timeit(grp)
    fl = 2 * 2.0 * B * H * W * C * C * 9
    print('C%-3d %2dx%-2d  student + teacher: two launches %5.1f us (%.3f of peak)   one grouped launch %5.1f us (%.3f)   %+.1f %%' % (
        C, H, W, a, fl / a / 1e6 / 157.3, bb, fl / bb / 1e6 / 157.3, 100 * (bb / a - 1)))
