#!/usr/bin/env python3
"""Does giving every launch lane its own quarter of the chip beat letting four lanes share all of it?
HRNet-W32's four branch convs (3x3, C = 32 / 64 / 128 / 256 at 64x48 / 32x24 / 16x12 / 8x6, B = 32), each lane a chain of
REPS launches of its conv: (a) all on one stream, (b) four ordinary streams, (c) four streams created with
hipExtStreamCreateWithCUMask, lane l on XCDs {2l, 2l+1} (mask bit i = CU i / 8 of XCD i % 8 - the KFD deals the mask bits
to the XCDs round-robin), (d) each masked lane alone.  Eager launches through the C ABI (no graphs: a graph's kernel
nodes do not keep a stream's CU mask), so (b) is the like-for-like baseline, not the replayed step.
usage: microbench_cumask.py [fwd_stats|dgrad_bnb] [reps=200] [xcds_per_lane=2]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call

mode = sys.argv[1] if len(sys.argv) > 1 else 'fwd_stats'
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
XPL = int(sys.argv[3]) if len(sys.argv) > 3 else 2
B = 32
dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
hip = ctypes.CDLL('libamdhip64.so')
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
shapes = [(32, 64, 48), (64, 32, 24), (128, 16, 12), (256, 8, 6)]
T = []
for (C, H, W) in shapes:
    T.append(dict(C=C, H=H, W=W, x=torch.randn(B, H, W, C, device=dev), w=torch.randn(C, 3, 3, C, device=dev) * 0.05,
                  y=torch.empty(B, H, W, C, device=dev), res=torch.randn(B, H, W, C, device=dev),
                  mk=torch.randint(0, 16, (B * H * W * C // 4,), device=dev, dtype=torch.uint8), cc=torch.randn(B, H, W, C, device=dev),
                  mean=torch.zeros(C, device=dev), invstd=torch.ones(C, device=dev),
                  slots=torch.zeros(2 * C * 64, device=dev, dtype=torch.float64)))
flops = [2.0 * B * t['H'] * t['W'] * t['C'] * t['C'] * 9 for t in T]


def masked_stream(xcds):
    words = (ctypes.c_uint32 * 8)()
    for i in range(256):
        if i % 8 in xcds:
            words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return s


def launch(t, st):
    C, H, W = t['C'], t['H'], t['W']
    geom = (B, H, W, C, H, W, C, 3, 3, 1, 1)
    nbg = ctypes.c_int(0)
    if mode == 'fwd_stats':
        call('advmix_conv_fwd_ex', P(t['x']), P(t['w']), None, P(t['y']), *geom, None, None, None, None, 0.0, None, 0,
             P(t['slots']), ctypes.byref(nbg), st)
    else:
        call('advmix_conv_tr_w_bnb', P(t['x']), P(t['w']), P(t['res']), P(t['y']), *geom, P(t['mk']), P(t['cc']), P(t['mean']),
             P(t['invstd']), None, None, 1, P(t['slots']), ctypes.byref(nbg), st)


def timed(fn, n=5):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


plain = [ctypes.c_void_p(torch.cuda.Stream().cuda_stream) for _ in T]
keep = [torch.cuda.Stream() for _ in T]
plain = [ctypes.c_void_p(s.cuda_stream) for s in keep]
masked = [masked_stream(set(range(XPL * l, XPL * l + XPL))) for l in range(len(T))]


def chains(streams, lanes=range(4)):
    def fn():
        for _ in range(REPS):
            for l in lanes:
                launch(T[l], streams[l])
    return fn


def one_stream():
    for _ in range(REPS):
        for l in range(4):
            launch(T[l], plain[0])


for fn in (one_stream, chains(plain), chains(masked)):
    fn()
tot = sum(flops) * REPS
ser = timed(one_stream)
print('%s, %d launches per lane, %d XCDs per masked lane' % (mode, REPS, XPL))
print('one stream              %7.1f us per level of four  %.3f of peak' % (ser / REPS * 1e6, tot / ser / 157.3e12))
for name, st in (('four plain streams', plain), ('four masked streams', masked)):
    t = timed(chains(st))
    print('%-22s  %7.1f us per level of four  %.3f of peak' % (name, t / REPS * 1e6, tot / t / 157.3e12))
for l in range(4):
    tp = timed(chains(plain, [l]))
    tm = timed(chains(masked, [l]))
    print('lane %d alone (C = %3d)   plain %6.1f us   masked %6.1f us (x %.2f; %d of 8 XCDs)' % (
        l, T[l]['C'], tp / REPS * 1e6, tm / REPS * 1e6, tm / tp, XPL))
