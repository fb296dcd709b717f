#!/usr/bin/env python3
"""One HIP graph per launch lane, replayed concurrently on 4 streams, against ONE graph holding the 4 parallel chains.
usage: graph_per_lane.py [kernels_per_chain] [elems]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
for elems in (256, 65536, 3145728):
    lanes = 4
    bufs = [torch.zeros(elems, device=dev) for _ in range(lanes)]
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    graphs = []
    for li in range(lanes):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(streams[li]):
            for k in range(3):
                call('advmix_fill', P(bufs[li]), 0.0, elems, ctypes.c_void_p(streams[li].cuda_stream))
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=streams[li], capture_error_mode='thread_local'):
                for k in range(n):
                    call('advmix_fill', P(bufs[li]), float(k), elems, ctypes.c_void_p(streams[li].cuda_stream))
        graphs.append(g)
    torch.cuda.synchronize()

    def run():
        for li in range(lanes):
            with torch.cuda.stream(streams[li]):
                graphs[li].replay()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%8d floats: 4 single-chain graphs on 4 streams: %7.1f us per round, %.2f us per kernel, %.2f us per chain step' % (
        elems, dt * 1e6, dt * 1e6 / (n * lanes), dt * 1e6 / n))
