#!/usr/bin/env python3
"""ONE phase of the step (d_fwd | t_fwd | d_fwd_bwd | d_frozen | g_fwd_bwd) captured as a HIP graph and replayed back to back,
a `cos_` marker kernel between replays - run under `rocprofv3 --kernel-trace` and summarise with
`tools/analyze_trace.py <trace> 0.5 cos_kernel 8`: kernels in flight / idle gaps of that phase alone.
usage: trace_phase.py <phase> [workload]"""
import sys, os
phase = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
sys.path.insert(0, os.getcwd() + '/tools'); sys.path.insert(0, os.getcwd())
import torch
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'phase_times.py')).read()
exec(src[:src.index("D.train(); G.train(); T.eval()")])
D.train(); G.train(); T.eval()
fn = {'d_fwd': d_fwd, 't_fwd': t_fwd, 'd_fwd_bwd': d_fwd_bwd(False), 'd_frozen': d_fwd_bwd(True), 'g_fwd_bwd': g_fwd_bwd}[phase]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        fn()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    fn()
mark = torch.zeros(64, device=dev)
for _ in range(12):
    mark.cos_()
    g.replay()
mark.cos_()
torch.cuda.synchronize()
print('done', phase)
