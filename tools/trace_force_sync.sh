#!/bin/bash
# VERDICT r4 item 4 b: what the gradient exchange launches under ADVMIX_FORCE_SYNC=1 (ONE rank, real RCCL: the only RCCL this
# pool can run - it refuses two ranks on one device) and what it overlaps.  Kernel trace of the seven-graph step with the
# side-stream all-reduces between the replays; every kernel whose name mentions nccl / rccl (or a copy that stands in for
# the one-rank all-reduce) is listed with its grid, its workgroup size, its stream (queue id) and the kernels in flight
# beside it.  NCCL_MAX_NCHANNELS 4 / 8 / 16 is A/B'ed on the step time.   usage (GPU box, repo root): tools/trace_force_sync.sh <tag>
R=$PWD; TAG=$1; OUT=$R/gpurun_out/sync_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ADVMIX_FORCE_SYNC=1
rm -rf $OUT/raw
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o p -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-through-loop > $OUT/bench_trace.log 2>&1
T=$(ls $OUT/raw/*/*kernel_trace.csv $OUT/raw/*kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/analyze_trace.py $T 0.5 cat_views_kernel 4 > $OUT/trace_summary.txt 2>&1
python3 - $T > $OUT/exchange_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r) for r in rows)
marks = [s for s, e, r in ev if 'cat_views_kernel' in r['Kernel_Name']]
lo, hi = marks[-5], marks[-1]
sel = [(s, e, r) for s, e, r in ev if lo <= s < hi]
queues = collections.Counter(r.get('Queue_Id', '?') for s, e, r in sel)
print('steady-state window: 4 steps, %.2f ms each; launches per queue (stream): %s' % ((hi - lo) / 4 / 1e6, dict(queues)))
ex = [(s, e, r) for s, e, r in sel if any(k in r['Kernel_Name'].lower() for k in ('nccl', 'rccl', 'allreduce', 'all_reduce'))]
print('kernels whose name mentions nccl / rccl / allreduce in the window: %d' % len(ex))
agg = collections.defaultdict(list)
for s, e, r in ex:
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    beside = sum(1 for s2, e2, r2 in sel if s2 < e and e2 > s and r2 is not r)
    agg[(r['Kernel_Name'][:90], r.get('Queue_Id', '?'), grid // max(wg, 1), wg)].append(((e - s) / 1e3, beside))
for (name, q, nwg, wg), v in agg.items():
    print('  %-90s queue %s  %d workgroups x %d threads  calls %d  avg %.1f us  other kernels in flight beside it (avg) %.1f' % (
        name, q, nwg, wg, len(v), sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v)))
if not ex:
    # one rank: RCCL's all-reduce of an in-place buffer has nothing to move - list what DOES run on queues other than the busiest four
    side = collections.defaultdict(list)
    main_q = {q for q, _ in queues.most_common(4)}
    for s, e, r in sel:
        if r.get('Queue_Id', '?') not in main_q:
            side[(r['Kernel_Name'][:90], r.get('Queue_Id', '?'))].append((e - s) / 1e3)
    print('kernels on queues other than the four busiest (the launch lanes):')
    for (name, q), v in sorted(side.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print('  %-90s queue %s calls %d avg %.1f us' % (name, q, len(v), sum(v) / len(v)))
PY
rm -rf $OUT/raw
for ch in default 4 8 16; do
  if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
  python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-through-loop 2>/dev/null | tail -1 | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); print('NCCL_MAX_NCHANNELS=$ch  %.1f images/s  %.3f ms/step  verified %s identical %s finite %s capture %s s' % (d['value'], d['ms_per_step'], d.get('grad_exchange_verified'), d.get('replicas_identical'), d.get('all_finite'), d.get('graph_capture_s')))" >> $OUT/nchannels_ab.txt
done
unset NCCL_MAX_NCHANNELS ADVMIX_FORCE_SYNC
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-through-loop 2>/dev/null | tail -1 | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); print('no data-parallel machinery  %.1f images/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" >> $OUT/nchannels_ab.txt
cat $OUT/exchange_kernels.txt $OUT/nchannels_ab.txt; head -3 $OUT/trace_summary.txt
