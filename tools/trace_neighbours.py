#!/usr/bin/env python3
"""Who launches a kernel?  For every launch whose name contains <substring>, the kernels that ran just before / after it
on the SAME queue, counted over the steady-state window (the last whole steps, as analyze_trace.py delimits them).
usage: trace_neighbours.py <kernel_trace.csv> <substring> [step-marker kernel = cat_views_kernel] [steps = 4]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else 'cat_views_kernel'
nst = int(sys.argv[4]) if len(sys.argv) > 4 else 4


def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]


ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'),
             r.get('Grid_Size', '?'), r.get('Workgroup_Size', '?')) for r in rows)
marks = [e[0] for e in ev if marker in e[2]]
lo, hi = marks[-nst - 1], marks[-1]
sel = [e for e in ev if lo <= e[0] < hi]
byq = collections.defaultdict(list)
for e in sel:
    byq[e[3]].append(e)
pairs = collections.Counter()
dur = collections.defaultdict(float)
n = 0
for q, es in byq.items():
    for i, e in enumerate(es):
        if sub in e[2]:
            prev = short(es[i - 1][2]) if i > 0 else '-'
            nxt = short(es[i + 1][2]) if i + 1 < len(es) else '-'
            key = (prev, nxt, e[4], e[5])
            pairs[key] += 1
            dur[key] += (e[1] - e[0]) / 1e3
            n += 1
print('%d launches of *%s* in %d steps (%.1f per step) on %d queues' % (n, sub, nst, n / nst, len(byq)))
for key, c in pairs.most_common(25):
    print('  %5d  avg %5.1f us  grid %s wg %s   after %-60s before %s' % (c, dur[key] / c, key[2], key[3], key[0], key[1]))
