#!/usr/bin/env python3
"""Does a HIP graph replay run parallel chains concurrently - and does it depend on the ORDER the chains were captured in?
Four independent chains (one per stream) of L x {conv 3x3 C->C, norm-sized copy}, forked from and joined to the main
stream, captured (a) chain by chain (what ops._run_lanes does: lane 0's whole chain, then lane 1's, ...) and (b) breadth
first (kernel 1 of every lane, kernel 2 of every lane, ...), then replayed; also the same work on ONE stream.
usage: microbench_capture_order.py [L] [B]"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
SHAPES = [(64, 48, 32), (32, 24, 64), (16, 12, 128), (8, 6, 256)]      # HRNet-W32's four branches


def make(H, W, C):
    return dict(x=torch.randn(B, H, W, C, device=dev), w=torch.randn(C, 3, 3, C, device=dev) * 0.05,
                y=torch.empty(B, H, W, C, device=dev), H=H, W=W, C=C)


def conv(t, st):
    call('advmix_conv_fwd', P(t['x']), P(t['w']), None, P(t['y']), B, t['H'], t['W'], t['C'], t['H'], t['W'], t['C'], 3, 3, 1, 1, st)


def conv_back(t, st):                     # second half of a pair: y -> x (keeps the chain a real dependency chain)
    call('advmix_conv_fwd', P(t['y']), P(t['w']), None, P(t['x']), B, t['H'], t['W'], t['C'], t['H'], t['W'], t['C'], 3, 3, 1, 1, st)


bufs = [make(*s) for s in SHAPES]
side = [torch.cuda.Stream() for _ in range(3)]


def body(order):
    cur = torch.cuda.current_stream()
    for s in side:
        s.wait_stream(cur)
    hs = [ctypes.c_void_p(cur.cuda_stream)] + [ctypes.c_void_p(s.cuda_stream) for s in side]
    if order == 'one':
        for t in bufs:
            for k in range(L):
                (conv if k % 2 == 0 else conv_back)(t, hs[0])
    elif order == 'chain':
        for t, h in zip(bufs, hs):
            for k in range(L):
                (conv if k % 2 == 0 else conv_back)(t, h)
    else:
        for k in range(L):
            for t, h in zip(bufs, hs):
                (conv if k % 2 == 0 else conv_back)(t, h)
    for s in side:
        cur.wait_stream(s)


def timed(order, reps=30):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        body(order)
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        body(order)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    # eager, same order
    with torch.cuda.stream(st):
        for _ in range(3):
            body(order)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            body(order)
        torch.cuda.synchronize()
    ue = (time.perf_counter() - t0) / reps * 1e6
    print('%-6s graph replay %8.1f us   eager %8.1f us   (%d kernels)' % (order, us, ue, 4 * L), flush=True)


for o in ('one', 'chain', 'bfs', 'chain', 'bfs'):
    timed(o)
