#!/usr/bin/env python3
"""In-kernel shader clock of conv_direct's main loop (diagnostic build: tools/build_variant.sh clk conv_direct -DCD_CLK,
run with ADVMIX_SO=tools/_dbg/libclk.so): delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups, after
~1 s of back-to-back launches on random data (MI355X_MICROARCH.md, DVFS give-back item 6).
usage: conv_clock.py B Ci H W Co k s p [mode=fwd|fwd_stats] [seconds]"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from advmix_amd._lib import call, lib
B, Ci, H, W, Co, k, s, p = [int(v) for v in sys.argv[1:9]]
mode = sys.argv[9] if len(sys.argv) > 9 else 'fwd'
secs = float(sys.argv[10]) if len(sys.argv) > 10 else 1.0
dev = torch.device('cuda:0')
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
x = torch.randn(B, H, W, Ci, device=dev); w = torch.randn(Co, k, k, Ci, device=dev) * 0.05
y = torch.empty(B, Ho, Wo, Co, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
slots = torch.zeros(2 * Co * 64, device=dev, dtype=torch.float64); nbg = ctypes.c_int(0)
if mode == 'fwd_stats':
    def run():
        nbg.value = 0
        call('advmix_conv_fwd_ex', P(x), P(w), None, P(y), B, H, W, Ci, Ho, Wo, Co, k, k, s, p, None, None, None, None, 0.0,
             None, 0, P(slots), ctypes.byref(nbg), st)
else:
    run = lambda: call('advmix_conv_fwd', P(x), P(w), None, P(y), B, H, W, Ci, Ho, Wo, Co, k, k, s, p, st)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(200):
        run()
    n += 200
torch.cuda.synchronize()
dt = time.time() - t0
buf = (ctypes.c_ulonglong * (2 * 8192))()
lib.advmix_dbg_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.advmix_dbg_clk(buf, 2 * 8192) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
a = a[(a[:, 1] > 0)]
ghz = a[:, 0] / a[:, 1] * 0.1
print('%s B%d Ci%d %dx%d Co%d k%d: %.1f us/launch over %d launches; main loop of a workgroup: median %.0f shader cycles = '
      '%.2f us; in-kernel clock median %.3f GHz (p10 %.3f, p90 %.3f) over %d workgroups' % (
          mode, B, Ci, H, W, Co, k, dt / n * 1e6, n, np.median(a[:, 0]), np.median(a[:, 1]) / 100.0, np.median(ghz),
          np.percentile(ghz, 10), np.percentile(ghz, 90), len(a)))
