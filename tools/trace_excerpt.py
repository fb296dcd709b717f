#!/usr/bin/env python3
"""Which hardware queues a graph replay really uses, and an excerpt of its timeline.
usage: trace_excerpt.py <kernel_trace.csv> <marker kernel> [first row of the excerpt in the last step] [rows]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?'))
             for r in rows))
marks = [e[0] for e in ev if sys.argv[2] in e[2]]
lo, hi = marks[-2], marks[-1]
sel = [e for e in ev if lo < e[0] < hi]
print('last step: %d kernels, %.2f ms' % (len(sel), (hi - lo) / 1e6))
q = collections.Counter((e[3], e[4]) for e in sel)
for k, v in sorted(q.items()):
    t = sum(e[1] - e[0] for e in sel if (e[3], e[4]) == k)
    print('  queue %s stream %s: %5d kernels, %.2f ms of kernel time' % (k[0], k[1], v, t / 1e6))
a = int(sys.argv[3]) if len(sys.argv) > 3 else len(sel) // 2
n = int(sys.argv[4]) if len(sys.argv) > 4 else 100
t0 = sel[a][0]
for s, e, name, qu, st in sel[a:a + n]:
    nm = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
    print('%9.1f %7.1f  q%s s%s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, qu, st, nm))
