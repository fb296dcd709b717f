import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
b = torch.zeros(256, device=dev)
s = torch.cuda.Stream()
s2 = torch.cuda.Stream()
for n in (1, 5, 20, 80):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        call('advmix_fill', P(b), 0.0, 256, ctypes.c_void_p(s.cuda_stream))
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
            for k in range(n):
                call('advmix_fill', P(b), float(k), 256, ctypes.c_void_p(s.cuda_stream))
    torch.cuda.synchronize()
    reps = 200
    with torch.cuda.stream(s):
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print('graph of %3d tiny kernels: %.1f us per replay on the device (%.2f us per kernel), host %.1f us per replay' % (
        n, (t2 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6 / n, (t1 - t0) / reps * 1e6))
# ping-pong between two streams with wait_stream + small graphs (what a tape does at every level)
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
for g, st in ((g1, s), (g2, s2)):
    with torch.cuda.stream(st):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st, capture_error_mode='thread_local'):
            for k in range(5):
                call('advmix_fill', P(b), float(k), 256, ctypes.c_void_p(st.cuda_stream))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    with torch.cuda.stream(s):
        g1.replay()
    s2.wait_stream(s)
    with torch.cuda.stream(s2):
        g2.replay()
    s.wait_stream(s2)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('ping-pong of two 5-kernel graphs across two streams: %.1f us per pair on the device, host %.1f us' % (
    (t2 - t0) / 200 * 1e6, (t1 - t0) / 200 * 1e6))
