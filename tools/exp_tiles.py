"""Time + checksum of one conv variant (see exp_tiles.sh): usage exp_tiles.py B C H W mode iters"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call, lib
B, C, H, W = [int(v) for v in sys.argv[1:5]]
mode, iters = sys.argv[5], int(sys.argv[6])
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
x = torch.randn(B, H, W, C, generator=g).to(dev)
w = (torch.randn(C, 3, 3, C, generator=g) * 0.05).to(dev)
y = torch.zeros(B, H, W, C, device=dev)
x2, yy, cc = (torch.randn(B, H, W, C, generator=g).to(dev) for _ in range(3))
mk = torch.randint(0, 16, (B * H * W * C // 4,), generator=g, dtype=torch.uint8).to(dev)      # the activation bit mask
mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
slots = torch.zeros(2 * C * 64, device=dev, dtype=torch.float64)
nbg = ctypes.c_int(0)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
if mode == 'fwd_stats':
    def run():
        nbg.value = 0
        call('advmix_conv_fwd_ex', P(x), P(w), None, P(y), B, H, W, C, H, W, C, 3, 3, 1, 1, None, None, None, None, 0.0, None, 0,
             P(slots), ctypes.byref(nbg), st)
else:
    def run():
        nbg.value = 0
        call('advmix_conv_tr_w_bnb', P(x), P(w), P(x2), P(y), B, H, W, C, H, W, C, 3, 3, 1, 1, P(mk), P(cc), P(mean), P(invstd), None, None, 1,
             P(slots), ctypes.byref(nbg), st)
slots.zero_(); run(); torch.cuda.synchronize()
ck = (float(y.double().sum()), float(y.double().abs().sum()), float(slots.view(2, -1, C).sum(1).abs().sum()))
for _ in range(20):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(iters):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
fl = 2.0 * B * H * W * C * C * 9
print('C%d %dx%d %-9s cfg %d: %.1f us  %.3f of peak   y sum %.6e abs %.6e stats %.6e' % (
    C, H, W, mode, lib.advmix_conv_direct_config(0 if mode == 'fwd_stats' else 1, B, H, W, C, C, 3, 3, 1), us,
    fl / us / 1e6 / 157.3, ck[0], ck[1], ck[2]))
