#!/bin/bash
# Round-6 evidence in one gpurun call (GPU box, repo root): bench lines of the headline and the secondary configurations,
# same-box A/Bs of the round's switches, the step's kernel trace (4 lanes) and per-shape table (1 lane), PMC traffic of the
# roofline's dominant members, MFMA-busy counter of the step - and a MANIFEST of what was made from which tree
# (gpurun_out/<tag>/LATEST.json; tools/adopt_profiles.py copies the set into profiles/ and makes it the one bench.py and
# DESIGN.md quote: VERDICT r5 weak 9).     GRAFT_HEAD=<git rev> tools/collect_r06.sh <tag>  ->  gpurun_out/<tag>/
T=${1:-r06}; O=gpurun_out/$T; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])"; }
# 1. the step's trace + per-shape table first: bench.py's roofline quotes the per-shape table of THIS collection
tools/profile_step.sh $T > $O/profile_step.log 2>&1
cp gpurun_out/prof_$T/stats_4lane.csv $O/stats_4lanes.csv 2>/dev/null
for f in stats_1lane.csv trace_summary_4lanes.txt per_shape_1lane.csv per_shape_1lane.txt; do cp gpurun_out/prof_$T/$f $O/ 2>/dev/null; done
# 2. PMC: traffic of the dominant members; MFMA-busy cycles and traffic of the step (against the unprofiled step time of this box)
tools/pmc_dominant.sh $T > $O/pmc_dominant.log 2>&1; cp gpurun_out/pmc_$T/pmc_dominant.json gpurun_out/pmc_$T/pmc_wino32.json $O/ 2>/dev/null
python bench.py --no-cpu-baseline --no-roofline --no-through-loop > $O/bench_quick.json 2>> $O/bench.err
MS=$(python -c "import json; print(json.loads(open('$O/bench_quick.json').read().strip().splitlines()[-1])['ms_per_step'])")
bash tools/pmc_step.sh $T 3 $MS > $O/pmc_step.log 2>&1; cp gpurun_out/pmc_step_$T.json $O/pmc_step_mfma.json 2>/dev/null
# 3. the manifest BEFORE the headline line is made: bench.py reads profiles/LATEST.json - on this box the collection itself
python - <<PY
import json, os, time
files = {'per_shape_1lane': 'per_shape_1lane.csv', 'trace_summary_4lanes': 'trace_summary_4lanes.txt', 'stats_4lanes': 'stats_4lanes.csv',
         'stats_1lane': 'stats_1lane.csv', 'pmc_dominant': 'pmc_dominant.json', 'pmc_step_mfma': 'pmc_step_mfma.json'}
have = {k: 'profiles/${T}_' + v for k, v in files.items() if os.path.exists('$O/' + v)}
json.dump({'tag': '$T', 'head': os.environ.get('GRAFT_HEAD', 'unknown'), 'collected_at': time.strftime('%Y-%m-%d %H:%M:%S'), 'files': have},
          open('$O/LATEST.json', 'w'), indent=1)
# (on the box: the same set under profiles/, so that THIS run's bench line quotes THIS run's files)
import shutil
for k, v in files.items():
    if os.path.exists('$O/' + v):
        shutil.copy('$O/' + v, 'profiles/${T}_' + v)
shutil.copy('$O/LATEST.json', 'profiles/LATEST.json')
PY
# 4. bench lines
python bench.py > $O/bench_line.json 2>> $O/bench.err
for wl in resnet50 hrnet_w48 hrnet_w32_512; do
  python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $O/${wl}_bench_line.json 2>> $O/bench.err
done
python bench.py --path validate --no-cpu-baseline > $O/validate_bench_line.json 2>> $O/bench.err
python bench.py --path inputs --no-cpu-baseline > $O/inputs_bench_line.json 2>> $O/bench.err
python bench.py --path nms --no-cpu-baseline > $O/nms_bench_line.json 2>> $O/bench.err
ADVMIX_DETERMINISTIC=1 python bench.py --no-cpu-baseline --no-roofline --no-through-loop > $O/deterministic_bench_line.json 2>> $O/bench.err
ADVMIX_FORCE_SYNC=1 python bench.py --no-cpu-baseline --no-roofline > $O/force_sync_1rank_bench_line.json 2>> $O/bench.err
ADVMIX_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/dp2_shared_gpu_bench_line.json 2>> $O/bench.err
# 5. same-box A/Bs of the round's switches (two rounds each, interleaved)
tools/ab_env.sh $O/ab_round6.log hrnet_w32 30 "ADVMIX_INBN=0" "ADVMIX_INBN=1" "ADVMIX_WINO_NC=2" "ADVMIX_WINO=0" > /dev/null 2>> $O/bench.err
python tools/phase_times.py > $O/phase_times.log 2>&1
for f in bench_line resnet50_bench_line hrnet_w48_bench_line hrnet_w32_512_bench_line validate_bench_line inputs_bench_line nms_bench_line deterministic_bench_line force_sync_1rank_bench_line dp2_shared_gpu_bench_line; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r = d.get('roofline') or {}
    print('$f', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'frac_own', r.get('frac_own'), 'mfma_busy', r.get('mfma_busy_frac'), 'dominant', (r.get('dominant') or {}).get('kernel'),
          d.get('step_frac_of_fp32_mfma_peak'), d.get('grad_exchange_verified'), d.get('replicas_identical'), (d.get('ranks') or {}).get('exchange_wait_ms'))
except Exception as e:
    print('$f FAILED', e)
PY
done
cat $O/ab_round6.log; head -3 $O/trace_summary_4lanes.txt; grep "kernels in flight\|idle gaps" $O/trace_summary_4lanes.txt; tail -6 $O/pmc_step.log
