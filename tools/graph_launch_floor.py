#!/usr/bin/env python3
"""What a kernel launch costs inside a replayed HIP graph on this box: chains of tiny / small kernels, 1-4 parallel
branches (the launch lanes of ops.GroupFn), with and without joins.  Prints microseconds per kernel.
usage: graph_launch_floor.py [kernels_per_chain]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())


def bench(label, lanes, elems, join_every=0):
    bufs = [torch.zeros(elems, device=dev) for _ in range(lanes)]
    side = [torch.cuda.Stream() for _ in range(lanes - 1)]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        def body():
            cur = torch.cuda.current_stream()
            streams = [cur] + side
            for st in side:
                st.wait_stream(cur)
            for k in range(n):
                for li, st in enumerate(streams):
                    call('advmix_fill', P(bufs[li]), float(k), elems, ctypes.c_void_p(st.cuda_stream))
                if join_every and (k + 1) % join_every == 0:
                    for st in side:
                        cur.wait_stream(st)
                    for st in side:
                        st.wait_stream(cur)
            for st in side:
                cur.wait_stream(st)
        body()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
            body()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%-46s %7.1f us per replay, %5.2f us per kernel (%d kernels), %5.2f us per chain step' % (
        label, dt * 1e6, dt * 1e6 / (n * lanes), n * lanes, dt * 1e6 / n))


for elems, name in ((256, 'tiny fill (1 KB)'), (65536, 'fill 256 KB'), (3145728, 'fill 12.6 MB')):
    bench('%s, 1 chain' % name, 1, elems)
    bench('%s, 2 parallel chains' % name, 2, elems)
    bench('%s, 4 parallel chains' % name, 4, elems)
    if os.environ.get('FLOOR_JOINS') == '1':               # (event joins inside the capture crashed the HIP runtime once)
        bench('%s, 4 chains, join every 16 kernels' % name, 4, elems, 16)
