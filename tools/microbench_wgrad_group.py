"""Eight weight gradients of one geometry: eight advmix_conv_wgrad launches back to back against ONE advmix_conv_wgrad_group
launch (HRNet-W32's 3x3 C -> C branch convs at B = 32; 1.81 GFLOP each).  ADVMIX_WGRAD_GROUP_BLOCKS=<n> to sweep the grid."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call, lib

dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, n = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 8
check = len(sys.argv) > 2 and sys.argv[2] == 'check'     # also compare the grouped launch's result with the singles'


def timeit(fn, iters=50):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for C, H, W in ((32, 64, 48), (64, 32, 24), (128, 16, 12), (256, 8, 6), (64, 64, 48)):
    dys = [torch.randn(B, H, W, C, device=dev) for _ in range(n)]
    xs = [torch.randn(B, H, W, C, device=dev) for _ in range(n)]
    dws = [torch.zeros(C, 3, 3, C, device=dev) for _ in range(n)]
    geom = (B, H, W, C, H, W, C, 3, 3, 1, 1)
    arr = ctypes.c_void_p * n
    A, Bp, D = arr(*[t.data_ptr() for t in dys]), arr(*[t.data_ptr() for t in xs]), arr(*[t.data_ptr() for t in dws])

    def singles():
        for i in range(n):
            call('advmix_conv_wgrad', P(dys[i]), P(xs[i]), P(dws[i]), *geom, st)

    def grouped():
        assert lib.advmix_conv_wgrad_group(n, A, Bp, D, *geom, st) == 0
    if check:
        for d in dws: d.zero_()
        singles()
        ref = [d.clone() for d in dws]
        for d in dws: d.zero_()
        grouped()
        torch.cuda.synchronize()
        worst = max(float((d - r).abs().max() / r.abs().max()) for d, r in zip(dws, ref))
        assert worst < 2e-5, worst
    t1, t2 = timeit(singles), timeit(grouped)
    fl = 2.0 * B * H * W * C * C * 9 * n
    print('3x3 %3d->%-3d @%dx%d x%d: singles %.1f us (%.1f us each, %.2f of peak) | grouped %.1f us (%.1f us each, %.2f of peak)' % (
        C, C, H, W, n, t1, t1 / n, fl / t1 / 1e6 / 157.3, t2, t2 / n, fl / t2 / 1e6 / 157.3), flush=True)
