#!/usr/bin/env python3
"""conv_wino4.hip (4x4 / stride 2 / pad 1 as Winograd F(3x3,2x2) per input phase: input transform + 16 GEMMs in one launch +
output transform) against the direct kernel, launch to launch through the C ABI, on the forward-form 4x4 convs of
UnetGenerator(9, 3, 6) at 256x192.  Prints the error of both kernels against torch fp64 on a small case first.
usage: microbench_wino4.py [B=32] [iters=30]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from advmix_amd._lib import call, lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
REC = np.dtype([('w', '<u8'), ('u', '<u8'), ('Cn', '<i4'), ('Ck', '<i4'), ('role', '<i4'), ('blk0', '<i4')])


def transform(w):
    Co, R, S, Ci = w.shape
    u = torch.empty(lib.advmix_wino4_u_floats(Co, Ci), device=dev)
    nb = Co * 4 * Ci // 256
    ents = torch.from_numpy(np.array([(w.data_ptr(), u.data_ptr(), Co, Ci, 0, 0)], dtype=REC).view(np.uint8).copy()).to(dev)
    own = torch.zeros(nb, dtype=torch.int32, device=dev)
    call('advmix_w4_weights', P(ents), P(own), nb, st)
    torch.cuda.synchronize()
    return u


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


def problem(Bn, Ci, Co, H, W, bias=False, relu_input=False):
    x = torch.randn(Bn, H, W, Ci, device=dev)
    if relu_input:                                         # (what the U-Net's up path feeds its transposed convs: non-negative, mean 0.8 sigma)
        x = x.abs()
    w = torch.randn(Co, 4, 4, Ci, device=dev) * (1.0 / (16 * Ci) ** 0.5)
    b = torch.randn(Co, device=dev) if bias else None
    y0 = torch.empty(Bn, H // 2, W // 2, Co, device=dev)
    y1 = torch.empty_like(y0)
    u = transform(w)
    wsf = lib.advmix_conv4x4s2_wino_ws_floats(Bn, H, W, Ci, Co)
    ws = torch.empty(wsf, device=dev)
    direct = lambda: call('advmix_conv_fwd', P(x), P(w), P(b) if bias else None, P(y0), Bn, H, W, Ci, H // 2, W // 2, Co, 4, 4, 2, 1, st)
    wino = lambda: call('advmix_conv4x4s2_wino_fwd', P(x), P(u), P(b) if bias else None, P(y1), P(ws), wsf, Bn, H, W, Ci, Co, st)
    return x, w, b, y0, y1, direct, wino


# numerics: small ragged cases against torch fp64 (CPU)
# (+ two layers of the 6-down U-Net at 512x512 / B = 2 - the case whose gradient-median bound EXPERIMENTS K is about - with rms errors)
for (Bn, Ci, Co, H, W, bias) in [(2, 16, 32, 16, 12, False), (3, 32, 64, 20, 28, True), (1, 64, 128, 8, 6, False), (2, 128, 64, 4, 2, True),
                                 (2, 64, 128, 256, 256, True), (2, 256, 512, 64, 64, True), (-2, 64, 128, 256, 256, True), (-2, 256, 512, 64, 64, True)]:
    relu_in, Bn = Bn < 0, abs(Bn)                          # (negative batch = the same case on a rectified input)
    x, w, b, y0, y1, direct, wino = problem(Bn, Ci, Co, H, W, bias, relu_in)
    direct(); wino(); torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2),
                                     b.double().cpu() if bias else None, stride=2, padding=1).permute(0, 2, 3, 1)
    sc = ref.abs().max().item()
    e0, e1 = y0.double().cpu() - ref, y1.double().cpu() - ref
    print('numerics' + (' |x|' if relu_in else '') + ' B%d %d->%d %dx%d bias=%d: direct %.2e  wino4 %.2e of scale (|ref| max %.3f); rms error / rms value: direct %.2e  wino4 %.2e' % (
        Bn, Ci, Co, H, W, bias, e0.abs().max().item() / sc, e1.abs().max().item() / sc, sc,
        (e0.pow(2).mean() / ref.pow(2).mean()).sqrt().item(), (e1.pow(2).mean() / ref.pow(2).mean()).sqrt().item()), flush=True)

for Ci, Co, Ho, Wo in [(64, 256, 64, 48), (128, 512, 32, 24), (256, 1024, 16, 12), (512, 1024, 8, 6),
                       (64, 128, 64, 48), (128, 256, 32, 24), (256, 512, 16, 12), (512, 512, 8, 6), (512, 512, 4, 3)]:
    x, w, b, y0, y1, direct, wino = problem(B, Ci, Co, 2 * Ho, 2 * Wo)
    direct(); wino(); torch.cuda.synchronize()
    err = (y0 - y1).abs().max().item() / y0.abs().max().item()
    td, tw = timeit(direct), timeit(wino)
    fl = 2.0 * B * Ho * Wo * Co * Ci * 16
    print('%4d->%4d @%dx%d: direct %.1f us (%.3f of peak)  wino4 %.1f us (%.3f on direct flops)  x%.2f   |direct - wino4| %.1e of scale'
          % (Ci, Co, Ho, Wo, td, fl / td / 1e6 / 157.3, tw, fl / tw / 1e6 / 157.3, td / tw, err), flush=True)
    lo = torch.randn_like(y0)
    # the transposed form with the same filters: lo [Cl = Co] -> hi [Ch = Ci]
    bankw = w.permute(0, 3, 1, 2)
    from advmix_amd import ops as _ops
    bank = _ops.WinoBank([bankw]); bank.refresh(); torch.cuda.synchronize()
    if bankw._wino[2] is not None:
        yh0, yh1 = torch.empty_like(x), torch.empty_like(x)
        wst = lib.advmix_deconv4x4s2_wino_ws_floats(B, Ho, Wo, Co, Ci)
        wtb = torch.empty(wst, device=dev)
        dtr = lambda: call('advmix_conv_tr_w', P(lo), P(w), None, P(yh0), B, Ho, Wo, Co, 2 * Ho, 2 * Wo, Ci, 4, 4, 2, 1, st)
        wtr = lambda: call('advmix_deconv4x4s2_wino_fwd', P(lo), bankw._wino[2], None, None, P(yh1), P(wtb), wst, B, Ho, Wo, Co, Ci, st)
        dtr(); wtr(); torch.cuda.synchronize()
        errt = (yh0 - yh1).abs().max().item() / yh0.abs().max().item()
        t0, t1 = timeit(dtr), timeit(wtr)
        print('      transposed form %d->%d: direct %.1f us (%.3f)  wino4 %.1f us (x%.2f)   |direct - wino4| %.1e of scale'
              % (Co, Ci, t0, fl / t0 / 1e6 / 157.3, t1, t0 / t1, errt), flush=True)
    bank.release()
    # weight gradient: hi = x, lo = a gradient of y
    dw0, dw1 = torch.zeros_like(w), torch.zeros_like(w)
    wsg = [lib.advmix_conv4x4s2_wino_wgrad_ws_floats(B, 2 * Ho, 2 * Wo, Ci, Co, hv) for hv in (0, 1)]
    if wsg[0] > 0:
        wsw = torch.empty(wsg[0], device=dev)
        wsf = lib.advmix_conv4x4s2_wino_ws_floats(B, 2 * Ho, 2 * Wo, Ci, Co)
        vbuf = torch.empty(wsf, device=dev)
        call('advmix_conv4x4s2_wino_fwd', P(x), P(transform(w)), None, P(y1), P(vbuf), wsf, B, 2 * Ho, 2 * Wo, Ci, Co, st)   # leaves V(x) in vbuf
        dwg = lambda: call('advmix_conv_wgrad', P(lo), P(x), P(dw0), B, Ho, Wo, Co, 2 * Ho, 2 * Wo, Ci, 4, 4, 2, 1, st)
        wwg = lambda: call('advmix_conv4x4s2_wino_wgrad', P(x), P(lo), P(dw1), None, P(wsw), wsg[0], B, 2 * Ho, 2 * Wo, Ci, Co, st)
        wwv = lambda: call('advmix_conv4x4s2_wino_wgrad', P(x), P(lo), P(dw1), P(vbuf), P(wsw), wsg[1], B, 2 * Ho, 2 * Wo, Ci, Co, st)
        dwg(); wwg(); torch.cuda.synchronize()
        errw = (dw0 - dw1).abs().max().item() / dw0.abs().max().item()
        t0, t1, t2 = timeit(dwg), timeit(wwg), timeit(wwv)
        print('      weight gradient: direct %.1f us (%.3f)  wino4 %.1f us (x%.2f)  with V handed in %.1f us (x%.2f)   |direct - wino4| %.1e of scale'
              % (t0, fl / t0 / 1e6 / 157.3, t1, t0 / t1, t2, t0 / t2, errw), flush=True)
