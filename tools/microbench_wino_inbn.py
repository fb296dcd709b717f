#!/usr/bin/env python3
"""VERDICT r5 next 4, launch to launch: BatchNorm + ReLU applied to the input patch WHILE conv_wino stages it
(advmix_conv3x3_wino_fwd_inbn: every workgroup reduces the producer's 16 statistics slots itself, relu(fma((c - mean) * invstd,
gamma, beta)) on every staged element, workgroup (0, 0) publishes mean / invstd / running statistics) against what the step did
before for an inner edge conv1 -> BN1 -> ReLU -> conv2 of a BasicBlock (lib/models/pose_hrnet.py:41-57): norm_apply_slots (its
own launch, y written and re-read) + the plain Winograd conv.  Per shape: plain conv, fused conv, norm_apply_slots alone, the
pair back to back - and the fused result against the pair's.   usage: microbench_wino_inbn.py [B=32] [iters=200]
(The first measurement, profiles/r06d_microbench_wino_inbn.log, was a knock-in variant with precomputed scale / shift.)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd import ops
from advmix_amd._lib import lib, call

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
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


for C, H, W in ((32, 64, 48), (64, 32, 24), (128, 16, 12), (64, 64, 48)):
    rows = B * H * W
    c1 = torch.randn(B, H, W, C, device=dev)                      # conv1's raw output
    y1, y2, yf = torch.empty_like(c1), torch.empty_like(c1), torch.empty_like(c1)
    w = (torch.randn(C, 3, 3, C, device=dev) * 0.05).permute(0, 3, 1, 2)
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean2, invstd2 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    NS = 16
    sl = torch.zeros(2, NS, C, device=dev, dtype=torch.float64)   # conv1's statistics as its epilogue leaves them (spread over 16 slots)
    cc = c1.double().reshape(NS, -1, C)
    sl[0], sl[1] = cc.sum(1), (cc ** 2).sum(1)
    slots = sl.reshape(-1).contiguous()
    slots2 = torch.zeros(2 * C * 64, device=dev, dtype=torch.float64)
    bank = ops.WinoBank([w])
    bank.refresh()
    uf, ud = bank.images(w)
    ns = ctypes.c_int(0)

    def apply():
        call('advmix_norm_apply_slots', P(c1), P(slots), NS, rows, C, 1e-5, P(gam), P(bet), None, P(y1), 1, P(mean), P(invstd),
             P(rm), P(rv), P(nbt), 0.1, None, st)

    def conv(x, y):
        ns.value = 0
        call('advmix_conv3x3_wino_fwd', P(x), uf, P(y), B, H, W, C, C, None, None, None, None, 0.0, None, 0, P(slots2), ctypes.byref(ns), st)

    def fused():
        ns.value = 0
        call('advmix_conv3x3_wino_fwd_inbn', P(c1), uf, P(yf), B, H, W, C, C, P(slots), NS, P(gam), P(bet), 1e-5, P(mean2), P(invstd2),
             None, None, None, 0.1, P(slots2), ctypes.byref(ns), st)
    apply(); conv(y1, y2); fused()
    torch.cuda.synchronize()
    err = float((yf - y2).abs().max() / y2.abs().max())
    assert torch.equal(mean, mean2) or float((mean - mean2).abs().max()) < 1e-6, 'published statistics differ'
    t_fused, t_plain, t_apply = timed(fused), timed(lambda: conv(y1, y2)), timed(apply)
    t_pair = timed(lambda: (apply(), conv(y1, y2)))
    print('3x3 %d->%d @%dx%d B=%d: plain conv %5.1f us | fused (slots reduced + BN + ReLU on load) %5.1f us (+%.1f) | norm_apply_slots %5.1f us | '
          'pair back to back %5.1f us -> fused saves %.1f us per edge; fused vs pair result: rel err %.1e'
          % (C, C, H, W, B, t_plain, t_fused, t_fused - t_plain, t_apply, t_pair, t_pair - t_fused, err))
    bank.release()
