#!/usr/bin/env python3
"""Per-SHAPE kernel summary of a rocprofv3 kernel trace.

rocprofv3 --stats groups launches by kernel template; four convs of equal FLOPs but different resolution share one
template and one row.  This joins the trace (kernel name + grid) with the launch log the library writes under
ADVMIX_TRACE_SHAPES (bench.py --dump-shapes: kernel template, grid, problem shape, algorithmic FLOPs of every MFMA
launch of ONE step) and prints / writes one row per (kernel, shape): launches, average microseconds, TFLOP/s, fraction of
the 157.3 TFLOP/s fp32 matrix peak, share of the traced kernel time.  Kernels without a shape entry (norm, pointwise,
copies) are listed by name + grid.

usage: kernel_shapes.py <kernel_trace.csv> <shapes.csv> [out.csv] [tail_fraction=0.5]"""
import collections, csv, sys

PEAK = 157.3
trace, shapes = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else None
tail = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5

shape_of, seq_of, ambiguous = {}, collections.defaultdict(list), set()
for r in csv.DictReader(open(shapes)):
    key = (r['kernel'], int(r['grid_x']), int(r['grid_y']), int(r['grid_z']))
    desc = '%s %sx%s s%s %s->%s @%sx%s (B %s)' % (r['kind'], r['R'], r['S'], r['stride'], r['Ci'], r['Co'],
                                                  r['Ho'] if r['kind'].startswith('fwd') or r['kind'].startswith('wgrad') else r['Hi'],
                                                  r['Wo'] if r['kind'].startswith('fwd') or r['kind'].startswith('wgrad') else r['Wi'], r['N'])
    if key in shape_of and shape_of[key] != (desc, float(r['flops'])):
        ambiguous.add(key)                                 # two shapes share template AND grid
    shape_of.setdefault(key, (desc, float(r['flops'])))
    seq_of[r['kernel']].append((key[1:], desc, float(r['flops'])))   # launch order within the template (one step)

rows = list(csv.DictReader(open(trace)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + int((t1 - t0) * (1 - tail))                     # steady state: the tail of the run
agg = collections.defaultdict(lambda: [0, 0.0])
total = 0.0
counters = collections.Counter()
for s, e, r in ev:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    base = name.split('(')[0]
    short = base.split('::')[-1] if '<' not in base else base[base.index('::') + 2:] if base.startswith(('direct::', 'wgd::', 'wino::', 'smap::')) else base
    gx = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
    gy = int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))
    gz = int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_Z']))
    hit = shape_of.get((short, gx, gy, gz))
    if (short, gx, gy, gz) in ambiguous:
        # Same template and grid for several shapes: resolved by the launch ORDER within the template, which a
        # single-stream trace (ADVMIX_LANES=1) preserves - every step, eager or replayed, launches the logged sequence.
        seq = seq_of[short]
        i = counters[short] % len(seq)
        hit = (seq[i][1], seq[i][2]) if seq[i][0] == (gx, gy, gz) else ('AMBIGUOUS ' + hit[0], 0.0)
    counters[short] += 1
    key = (short, hit[0] if hit else 'grid %dx%dx%d' % (gx, gy, gz), hit[1] if hit else 0.0)
    if s < cut:                                            # (counted for the sequence, not for the steady-state table)
        continue
    agg[key][0] += 1
    agg[key][1] += (e - s) / 1e3
    total += (e - s) / 1e3
lines = [('kernel', 'shape', 'launches', 'avg_us', 'tflops', 'frac_of_fp32_mfma_peak', 'share_of_kernel_time')]
mf = mt = 0.0
for (k, d, fl), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tf = fl / (us / n * 1e-6) / 1e12 if fl else 0.0
    if fl:
        mf += fl * n
        mt += us * 1e-6
    lines.append((k, d, n, '%.2f' % (us / n), '%.2f' % tf if fl else '', '%.4f' % (tf / PEAK) if fl else '', '%.4f' % (us / total)))
lines.append(('ALL MFMA KERNELS WITH A SHAPE ENTRY', 'time-weighted', '', '', '%.2f' % (mf / mt / 1e12 if mt else 0),
              '%.4f' % (mf / mt / 1e12 / PEAK if mt else 0), '%.4f' % (mt * 1e6 / total if total else 0)))
if out:
    with open(out, 'w', newline='') as f:
        csv.writer(f).writerows(lines)
for ln in lines[:45] + lines[-1:]:
    print('  '.join(str(x) for x in ln))
