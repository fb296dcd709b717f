#!/usr/bin/env python3
"""Device-side hand-offs between lanes replayed as separate single-chain HIP graphs: cost and liveness.
Four lanes, each a sequence of graphs (one per level) of ``k`` tiny kernels; after every level lane 0 waits for the
three others and they wait for lane 0 (a fork / join per level), with signal / wait kernels inside the graphs.
usage: lane_handoff.py [levels] [kernels_per_level]"""
import ctypes, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 60
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
P = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)
L = 4
streams = [torch.cuda.Stream() for _ in range(L)]
bufs = [torch.zeros(256, device=dev) for _ in range(L)]
epoch = torch.zeros(L, dtype=torch.int64, device=dev)
fork = torch.zeros(levels, dtype=torch.int64, device=dev)      # signalled by lane 0
join = torch.zeros(levels, dtype=torch.int64, device=dev)      # signalled by lanes 1..3
err = torch.zeros(1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
graphs = [[] for _ in range(L)]
for lv in range(levels):
    for l in range(L):
        g = torch.cuda.CUDAGraph()
        st = ctypes.c_void_p(streams[l].cuda_stream)
        with torch.cuda.stream(streams[l]):
            with torch.cuda.graph(g, stream=streams[l], capture_error_mode='thread_local'):
                if lv == 0:
                    call('advmix_lane_tick', P(epoch, 8 * l), st)
                if l == 0:
                    call('advmix_lane_signal', P(fork, 8 * lv), st)
                else:
                    call('advmix_lane_wait', P(fork, 8 * lv), P(epoch, 8 * l), 1, P(err), st)
                for i in range(k):
                    call('advmix_fill', P(bufs[l]), float(i), 256, st)
                if l == 0:
                    call('advmix_lane_wait', P(join, 8 * lv), P(epoch, 0), L - 1, P(err), st)
                else:
                    call('advmix_lane_signal', P(join, 8 * lv), st)
        graphs[l].append(g)
torch.cuda.synchronize()


def replay():
    for lv in range(levels):
        for l in range(L):
            with torch.cuda.stream(streams[l]):
                graphs[l][lv].replay()


for _ in range(3):
    replay()
torch.cuda.synchronize()
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
n = levels * L * k
print('%d levels x 4 lanes x %d kernels: %.1f us per level on the device (%.2f us per kernel overall), host %.1f us per level; '
      'err flag %d, epochs %s, join[-1] %d' % (levels, k, (t2 - t0) / reps / levels * 1e6, (t2 - t0) / reps * 1e6 / n,
                                               (t1 - t0) / reps / levels * 1e6, int(err), epoch.tolist(), int(join[-1])))
