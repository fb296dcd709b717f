#!/usr/bin/env python3
"""MFMA-busy cycles of the AdvMix step from a rocprofv3 --pmc pass over bench.py (tools/pmc_step.sh).
usage: summarize_pmc_step.py <dir> <timed steps> <warm-up steps> <out.json> [step_ms=58.1] [GHz=2.4]
SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's SIMDs (cross-check: the dominant conv, 1.81 GFLOP = 11.5 us of MFMA on
1024 SIMDs, reads 28.3 M busy cycles = 11.5 us x 2.4 GHz x 1024).  Utilisation of the step = busy cycles of ONE step /
(step time x clock x 1024 SIMDs), with the step time of the un-profiled HIP-graph run."""
import collections, csv, glob, json, sys
root, steps, warm, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
step_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 58.1
ghz = float(sys.argv[6]) if len(sys.argv) > 6 else 2.4
f = glob.glob('%s/*/*_counter_collection.csv' % root) + glob.glob('%s/*_counter_collection.csv' % root)
tot = collections.Counter()
per = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    tot[r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
        n = r['Kernel_Name'].replace('void ', '').split('(')[0][:70]
        per[n][0] += float(r['Counter_Value']); per[n][1] += 1
runs = steps + warm + 2 + 0                     # timed + warm-up + the runner-free eager path has no extra capture steps
busy = tot['SQ_VALU_MFMA_BUSY_CYCLES']
res = {'counter_totals_whole_run': dict(tot), 'steps_in_run_assumed': steps + warm,
       'mfma_busy_cycles_per_step': busy / (steps + warm),
       'step_ms_unprofiled': step_ms, 'clock_GHz': ghz,
       'mfma_utilisation_of_step': busy / (steps + warm) / (step_ms * 1e-3 * ghz * 1e9 * 1024),
       'top_kernels_by_mfma_busy': [{'kernel': k, 'share': v[0] / busy, 'dispatches': v[1]}
                                    for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])[:12]]}
# memory-side (beyond-L2) traffic, if the FETCH_SIZE / WRITE_SIZE passes are there (KiB; Infinity-Cache hits are counted; on gfx950
# FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streams at 64 bytes: doubled, MI355X_MICROARCH.md "HBM")
mem = {}
for sub, name in (('_fetch', 'FETCH_SIZE'), ('_write', 'WRITE_SIZE')):
    ff = glob.glob('%s%s/*/*_counter_collection.csv' % (root, sub)) + glob.glob('%s%s/*_counter_collection.csv' % (root, sub))
    if ff:
        mem[name] = sum(float(r['Counter_Value']) for r in csv.DictReader(open(ff[0])) if r['Counter_Name'] == name)
if len(mem) == 2:
    rd = mem['FETCH_SIZE'] * 1024 * 2 / (steps + warm)
    wr = mem['WRITE_SIZE'] * 1024 / (steps + warm)
    res['memory_side_bytes_per_step'] = {'read_corrected': rd, 'written': wr, 'total': rd + wr,
                                         'average_GBps_over_the_step': (rd + wr) / (step_ms * 1e-3) / 1e9,
                                         'frac_of_8TBps': (rd + wr) / (step_ms * 1e-3) / 8e12}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != 'counter_totals_whole_run'}, indent=1)[:1800])
