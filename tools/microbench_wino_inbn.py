#!/usr/bin/env python3
"""VERDICT r5 next 4, measured where it was asked: BatchNorm + ReLU applied to the input patch WHILE conv_wino stages it
(relu(fma(c, scale, shift)) on every staged element, ~1.3x per input element) against what the step does today for an inner
edge conv1 -> BN1 -> ReLU -> conv2 of a BasicBlock (lib/models/pose_hrnet.py:41-57): norm_apply_slots (its own launch, y
written and re-read) + the plain Winograd conv.  Needs the measurement library:
    tools/build_variant.sh inbn conv_wino -DWN_INBN ;  ADVMIX_SO=tools/_dbg/libinbn.so python tools/microbench_wino_inbn.py [B] [iters]
Prints per shape: plain conv, fused conv, norm_apply_slots alone, and the back-to-back pair - and checks the fused result
against the pair's."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd import ops
from advmix_amd._lib import lib, call

assert lib.advmix_build_flags() & 64, 'run with ADVMIX_SO=tools/_dbg/libinbn.so (tools/build_variant.sh inbn conv_wino -DWN_INBN)'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device('cuda:0')
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.advmix_dbg_wino_inbn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]


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
    y1 = torch.empty_like(c1)
    y2 = torch.empty_like(c1)
    yf = torch.empty_like(c1)
    w = (torch.randn(C, 3, 3, C, device=dev) * 0.05).permute(0, 3, 1, 2)
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    slots = torch.zeros(2 * C * 64, device=dev, dtype=torch.float64)
    slots2 = torch.zeros(2 * C * 64, device=dev, dtype=torch.float64)
    bank = ops.WinoBank([w])
    bank.refresh()
    uf, ud = bank.images(w)
    ns = ctypes.c_int(0)
    # conv1's statistics in the slots, as its epilogue leaves them (16 slots used)
    NS = 16
    s1 = c1.double().sum((0, 1, 2))
    s2 = (c1.double() ** 2).sum((0, 1, 2))
    sl = torch.zeros(2, NS, C, device=dev, dtype=torch.float64)
    sl[0, 0], sl[1, 0] = s1, s2
    slots[:2 * NS * C] = sl.reshape(-1)
    keep = slots.clone()

    def apply():
        slots.copy_(keep)                                         # (norm_apply_slots clears the slots it reads)
        call('advmix_norm_apply_slots', P(c1), P(slots), NS, rows, C, 1e-5, P(gam), P(bet), None, P(y1), 1, P(mean), P(invstd),
             P(rm), P(rv), P(nbt), 0.1, None, st)

    def apply_only():
        call('advmix_norm_apply_slots', P(c1), P(slots), NS, rows, C, 1e-5, P(gam), P(bet), None, P(y1), 1, P(mean), P(invstd),
             P(rm), P(rv), P(nbt), 0.1, None, st)

    def conv(x, y):
        ns.value = 0
        lib.advmix_conv3x3_wino_fwd(P(x), uf, P(y), B, H, W, C, C, None, None, None, None, 0.0, None, 0, P(slots2), ctypes.byref(ns), st)
    lib.advmix_dbg_wino_inbn(None, None)
    apply()
    conv(y1, y2)
    torch.cuda.synchronize()
    scale = (gam * invstd).contiguous()
    shift = (bet - mean * gam * invstd).contiguous()
    lib.advmix_dbg_wino_inbn(P(scale), P(shift))
    conv(c1, yf)
    torch.cuda.synchronize()
    err = float((yf - y2).abs().max() / y2.abs().max())
    t_fused = timed(lambda: conv(c1, yf))
    lib.advmix_dbg_wino_inbn(None, None)
    t_plain = timed(lambda: conv(y1, y2))
    t_apply = timed(apply_only)
    t_pair = timed(lambda: (apply_only(), conv(y1, y2)))
    print('3x3 %d->%d @%dx%d B=%d: plain conv %5.1f us | fused (BN + ReLU on load) %5.1f us (+%.1f) | norm_apply_slots %5.1f us | pair back to back '
          '%5.1f us -> fused saves %.1f us per edge; fused vs pair result: rel err %.1e'
          % (C, C, H, W, B, t_plain, t_fused, t_fused - t_plain, t_apply, t_pair, t_pair - t_fused, err))
    bank.release()
