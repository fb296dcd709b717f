#!/usr/bin/env python3
"""Can the teacher's forward hide behind the generator's / the student's?  Each variant captured as one HIP graph and
replayed (as tools/phase_times.py does).  Measured (MI355X, B = 32): G fwd then T fwd 9.86 ms; T as ONE chain 7.53 ms;
G fwd || T (one chain on an aux stream) 11.14 ms = the sum of the two; (G fwd, D fwd) || T 17.91 vs 17.28 ms one after
the other.  No overlap: a fifth stream shares one of the runtime's four hardware queues, and with eight queues the
multi-branch graph itself is slower (tools/tape_ab.sh).  Capturing T with its four lanes beside G crashes the HIP
runtime (the lanes' streams then belong to two branches of the capture)."""
import sys, os
sys.argv = [sys.argv[0]]
sys.path.insert(0, os.getcwd() + '/tools'); sys.path.insert(0, os.getcwd())
import torch
src = open('tools/phase_times.py').read()
head = src[:src.index("D.train(); G.train(); T.eval()")]
exec(head)
D.train(); G.train(); T.eval()
AUX = torch.cuda.Stream()
def g_then_t():
    g_fwd(); t_fwd()
def t_one_lane():
    old = ops.MAX_LANES
    ops.MAX_LANES = 1
    try:
        t_fwd()
    finally:
        ops.MAX_LANES = old
def g_and_t():
    cur = torch.cuda.current_stream()
    AUX.wait_stream(cur)
    with torch.cuda.stream(AUX):
        t_one_lane()
    g_fwd()
    cur.wait_stream(AUX)
def gd_and_t():
    cur = torch.cuda.current_stream()
    AUX.wait_stream(cur)
    with torch.cuda.stream(AUX):
        t_one_lane()
    g_fwd(); d_fwd()
    cur.wait_stream(AUX)
def gdt():
    g_fwd(); d_fwd(); t_fwd()
timed('G fwd then T fwd (one after the other)', g_then_t)
timed('T fwd as one chain', t_one_lane)
timed('G fwd || T fwd (T as one chain on an aux stream)', g_and_t)
timed('G fwd, D fwd, T fwd (one after the other)', gdt)
timed('(G fwd, D fwd) || T fwd', gd_and_t)
