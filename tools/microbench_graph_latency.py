#!/usr/bin/env python3
"""What one more launch costs a SEQUENTIAL region of the step: N dependent tiny kernels (advmix_fill of 4 floats) on one stream,
eager and replayed from a HIP graph, and the same with a fork / join onto a second stream around every second kernel; then the
same chains with a realistic small kernel (advmix_fill of 4 MB).  us per launch = total / N."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call

dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
P = lambda t: ctypes.c_void_p(t.data_ptr())


def chain(buf, n, st, side=None):
    cur = torch.cuda.current_stream()
    for i in range(n):
        if side is not None and i % 2 == 1:
            side.wait_stream(cur)
            call('advmix_fill', P(buf), float(i), buf.numel(), ctypes.c_void_p(side.cuda_stream))
            cur.wait_stream(side)
        else:
            call('advmix_fill', P(buf), float(i), buf.numel(), st)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


s = torch.cuda.Stream()
side = torch.cuda.Stream()
with torch.cuda.stream(s):
    st = ctypes.c_void_p(s.cuda_stream)
    for nm, numel in (('4 floats', 4), ('4 MB', 1 << 20), ('64 MB', 16 << 20)):
        buf = torch.zeros(numel, device=dev)
        for fork in (False, True):
            sd = side if fork else None
            t_eager = timed(lambda: chain(buf, N, st, sd), 5)
            g = torch.cuda.CUDAGraph()
            chain(buf, N, st, sd); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                chain(buf, N, st, sd)
            t_graph = timed(g.replay)
            print('%-9s %-22s eager %6.2f us / launch   graph replay %6.2f us / launch' % (
                nm, 'fork/join every 2nd' if fork else 'one stream', t_eager / N * 1e6, t_graph / N * 1e6))
