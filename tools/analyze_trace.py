#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: busy fraction, concurrency, per-kernel totals for the
steady-state tail of the run.  usage: analyze_trace.py <kernel_trace.csv> [tail_fraction]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + int((t1 - t0) * (1 - frac))
sel = [e for e in ev if e[0] >= cut]
if len(sys.argv) > 3:
    # steady-state steps only: argv[3] = a kernel that runs exactly once per step, first thing (cat_views_kernel for the
    # AdvMix step).  The window is the last argv[4] (default 4) whole steps - replayed graphs, no capture / warm-up / host
    # set-up gaps in it.
    marks = [e[0] for e in ev if sys.argv[3] in e[2]]
    nst = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    lo, hi = marks[-nst - 1], marks[-1]
    sel = [e for e in ev if lo <= e[0] < hi]
    print('steady-state window: %d steps of %.2f ms' % (nst, (hi - lo) / nst / 1e6))
span = max(e[1] for e in sel) - sel[0][0]
busy, cs, ce = 0, sel[0][0], sel[0][1]
for s, e, _ in sel[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
tot = sum(e[1] - e[0] for e in sel)
print('window %.1f ms  busy(union) %.1f ms (%.1f%%)  sum of kernels %.1f ms  (avg concurrency %.2f)  launches %d' % (
    span / 1e6, busy / 1e6, 100 * busy / span, tot / 1e6, tot / busy, len(sel)))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    n = n.split('(')[0][:80]
    agg[n][0] += e - s; agg[n][1] += 1
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print('%6.2f%%  calls %6d  avg %8.1f us  %s' % (100 * d / tot, c, d / c / 1e3, n))

# concurrency histogram: share of the window with exactly k kernels in flight, and how much of the
# window has no MFMA (conv / wgrad) kernel in flight at all
pts = []
for i, (s, e, n) in enumerate(sel):
    heavy = ('conv_' in n) or ('wgrad' in n)            # every MFMA kernel: conv_direct / conv_wino / conv_smap / conv_igemm / conv_wgrad* / wgrad*
    pts.append((s, 1, heavy, i)); pts.append((e, -1, heavy, i))
pts.sort()
hist = collections.defaultdict(int)
k = h = 0
no_heavy = 0
prev = pts[0][0]
live = set()
alone = collections.defaultdict(int)                        # time a kernel is the ONLY one in flight, by kernel name
for t, d, heavy, i in pts:
    hist[k] += t - prev
    if h == 0 and k > 0:
        no_heavy += t - prev
    if k == 1:
        alone[next(iter(live))] += t - prev
    prev = t
    k += d
    if d > 0:
        live.add(i)
    else:
        live.discard(i)
    if heavy:
        h += d
print('kernels in flight: ' + '  '.join('%d: %.1f%%' % (kk, 100 * v / span) for kk, v in sorted(hist.items()) if v > 0.002 * span))
print('time with kernels in flight but no conv/wgrad kernel among them: %.1f%% of the window' % (100 * no_heavy / span))
by_name = collections.defaultdict(int)
for i, dt in alone.items():
    by_name[sel[i][2].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:80]] += dt
print('time a kernel runs ALONE (nothing else in flight), by kernel:')
for n, dt in sorted(by_name.items(), key=lambda kv: -kv[1])[:16]:
    print('  %6.2f%% of the window  %s' % (100 * dt / span, n))

# idle gaps (no kernel in flight): how long they are, and which kernels bracket them
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    return n[:70]
gaps = []
cur_end, last = sel[0][1], sel[0][2]
for s, e, n in sel[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, short(last), short(n)))
    if e > cur_end:
        cur_end, last = e, n
edges = [0, 1e3, 2e3, 4e3, 6e3, 8e3, 12e3, 20e3, 50e3, 200e3, 1e12]
tot_idle = sum(g[0] for g in gaps)
print('idle gaps: %d, %.1f ms = %.1f%% of the window' % (len(gaps), tot_idle / 1e6, 100 * tot_idle / span))
for lo, hi in zip(edges[:-1], edges[1:]):
    gs = [g[0] for g in gaps if lo <= g[0] < hi]
    if gs:
        print('  %6.0f - %-8s us: %6d gaps  %7.2f ms  (%.1f%% of the window)' % (lo / 1e3, ('%.0f' % (hi / 1e3)) if hi < 1e11 else 'inf',
                                                                               len(gs), sum(gs) / 1e6, 100 * sum(gs) / span))
pair = collections.defaultdict(lambda: [0, 0])
for d, a, b in gaps:
    pair[(a, b)][0] += d; pair[(a, b)][1] += 1
print('idle time by (kernel that ended last -> kernel that started next):')
for (a, b), (d, c) in sorted(pair.items(), key=lambda kv: -kv[1][0])[:25]:
    print('  %6.2f ms  %5d gaps  avg %5.1f us  %s -> %s' % (d / 1e6, c, d / c / 1e3, a, b))
