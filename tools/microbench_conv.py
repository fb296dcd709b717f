#!/usr/bin/env python3
"""Time one conv configuration through the C ABI directly (ctypes, ~3 us/call of host overhead,
so the HIP-event average is the kernel's launch-to-launch time, not Python's).
usage: microbench_conv.py B Ci H W Co k stride pad [mode=fwd|fwd_stats|dgrad|dgrad_narrow|dgrad_bt|dgrad_bt_add|dgrad_bnb|wgrad] [iters]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call

B, Ci, H, W, Co, k, s, p = [int(v) for v in sys.argv[1:9]]
mode = sys.argv[9] if len(sys.argv) > 9 else 'fwd'
iters = int(sys.argv[10]) if len(sys.argv) > 10 else 50
dev = torch.device('cuda:0')
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
x = torch.randn(B, H, W, Ci, device=dev)
w = torch.randn(Co, k, k, Ci, device=dev) * 0.05
wt = torch.randn(Ci, k, k, Co, device=dev) * 0.05
y = torch.randn(B, Ho, Wo, Co, device=dev)
dw = torch.zeros(Co, k, k, Ci, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
if mode == 'fwd_stats':      # the variant the training step launches: conv + BN column sums in the epilogue
    slots = torch.zeros(2 * Co * 64, device=dev, dtype=torch.float64)
    nbg = ctypes.c_int(0)
    run = lambda: call('advmix_conv_fwd_ex', P(x), P(w), None, P(y), B, H, W, Ci, Ho, Wo, Co, k, k, s, p,
                       None, None, None, None, 0.0, None, 0, P(slots), ctypes.byref(nbg), st)
elif mode == 'fwd':
    run = lambda: call('advmix_conv_fwd', P(x), P(w), None, P(y), B, H, W, Ci, Ho, Wo, Co, k, k, s, p, st)
elif mode == 'dgrad':        # weights pre-transposed to [Ci][R][S][Co] (one re-layout launch per use, or a mirror per step)
    run = lambda: call('advmix_conv_tr', P(y), P(wt), None, P(x), B, Ho, Wo, Co, H, W, Ci, k, k, s, p, st)
elif mode == 'dgrad_narrow':  # <= 4 input channels (a network's first conv): one thread per pixel
    run = lambda: call('advmix_conv_tr_narrow', P(y), P(w), P(x), B, Ho, Wo, Co, H, W, Ci, k, k, s, p, st)
elif mode == 'dgrad_bt':     # weights in their own layout, scattered into the LDS image (what the step launches)
    run = lambda: call('advmix_conv_tr_w_add', P(y), P(w), None, P(x), B, Ho, Wo, Co, H, W, Ci, k, k, s, p, st)
elif mode == 'dgrad_bt_add':
    x2 = torch.randn_like(x)
    run = lambda: call('advmix_conv_tr_w_add', P(y), P(w), P(x2), P(x), B, Ho, Wo, Co, H, W, Ci, k, k, s, p, st)
elif mode == 'dgrad_bnb':    # + addend + BatchNorm-backward epilogue of the producer
    x2, yy, cc = torch.randn_like(x), torch.randn_like(x), torch.randn_like(x)
    mk = torch.randint(0, 16, (x.numel() // 4,), device=x.device, dtype=torch.uint8)       # the activation bit mask
    mean, invstd = torch.zeros(Ci, device=dev), torch.ones(Ci, device=dev)
    slots = torch.zeros(2 * Ci * 64, device=dev, dtype=torch.float64)
    nbg = ctypes.c_int(0)
    def run():
        nbg.value = 0
        call('advmix_conv_tr_w_bnb', P(y), P(w), P(x2), P(x), B, Ho, Wo, Co, H, W, Ci, k, k, s, p, P(mk), P(cc), P(mean),
             P(invstd), None, None, 1, P(slots), ctypes.byref(nbg), st)
else:
    run = lambda: call('advmix_conv_wgrad', P(y), P(x), P(dw), B, Ho, Wo, Co, H, W, Ci, k, k, s, p, st)
for _ in range(5):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * B * Ho * Wo * Co * Ci * k * k
print('%-5s B%d Ci%d %dx%d Co%d k%d s%d [%s]: %.1f us/launch  %.1f TFLOP/s (%.1f%% of 157.3)' % (
    mode, B, Ci, H, W, Co, k, s, os.environ.get('ADVMIX_CONV', 'direct'), ms * 1e3, fl / ms / 1e9,
    fl / ms / 1e9 / 157.3 * 100))
