#!/usr/bin/env python3
"""What the batched GEMMs of a non-fused Winograd F(3x3,2x2)-per-phase form of the U-Net's 4x4 / stride-2 convs would cost
on the direct kernel (1x1 conv, 16 'images' = the 16 positions xi, rows = 3x3 output tiles, K' = 4 Cin, N = Cout), beside
the direct 4x4 conv itself.  usage: microbench_wino4_gemm.py [B=32] [iters=30]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(run):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def conv(Bn, H, W, Ci, Co, k, s, p, mode):
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(Bn, H, W, Ci, device=dev)
    w = torch.randn(Co, k, k, Ci, device=dev) * 0.05
    y = torch.randn(Bn, Ho, Wo, Co, device=dev)
    dw = torch.zeros(Co, k, k, Ci, device=dev)
    if mode == 'fwd':
        return timeit(lambda: call('advmix_conv_fwd', P(x), P(w), None, P(y), Bn, H, W, Ci, Ho, Wo, Co, k, k, s, p, st))
    return timeit(lambda: call('advmix_conv_wgrad', P(y), P(x), P(dw), Bn, Ho, Wo, Co, H, W, Ci, k, k, s, p, st))


# (Cin, Cout, output map) of the forward-form 4x4 / stride-2 launches of UnetGenerator(9, 3, 6) at 256x192
for Ci, Co, Ho, Wo in [(64, 256, 64, 48), (128, 512, 32, 24), (256, 1024, 16, 12), (512, 1024, 8, 6),
                       (64, 128, 64, 48), (128, 256, 32, 24), (256, 512, 16, 12), (512, 512, 8, 6)]:
    th, tw = -(-Ho // 3), -(-Wo // 3)
    tiles = B * th * tw
    rows = -(-tiles // 128) * 128
    fl_direct = 2.0 * B * Ho * Wo * Co * Ci * 16
    t_dir = conv(B, 2 * Ho, 2 * Wo, Ci, Co, 4, 2, 1, 'fwd')
    t_dirw = conv(B, 2 * Ho, 2 * Wo, Ci, Co, 4, 2, 1, 'wgrad')
    # the GEMM as a 1x1 conv over 16 images of rows / 128 x 128 'pixels'
    t_g = conv(16, rows // 128, 128, 4 * Ci, Co, 1, 1, 0, 'fwd')
    t_gw = conv(16, rows // 128, 128, 4 * Ci, Co, 1, 1, 0, 'wgrad')
    fl_g = 2.0 * 16 * rows * 4 * Ci * Co
    v_mb = 16 * rows * 4 * Ci * 4 / 1e6
    m_mb = 16 * rows * Co * 4 / 1e6
    print('%4d->%4d @%dx%d: direct fwd %.1f us (%.3f of peak)  wgrad %.1f us | tiles %d (rows %d) K\' %d: GEMM fwd %.1f us '
          '(%.3f of peak on its own flops)  wgrad-GEMM %.1f us | V %.0f MB  M %.0f MB -> transforms >= %.1f us at 5 TB/s'
          % (Ci, Co, Ho, Wo, t_dir, fl_direct / t_dir / 1e6 / 157.3, t_dirw, tiles, rows, 4 * Ci, t_g,
             fl_g / t_g / 1e6 / 157.3, t_gw, v_mb, m_mb, 2 * (v_mb + m_mb) / 5.0), flush=True)
