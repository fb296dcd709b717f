#!/bin/bash
# The measurements a round commits under profiles/ (run on the GPU box from the repo root): bench lines of the
# headline and the secondary configurations, phase times, rocprofv3 kernel summaries -> gpurun_out/r2t, gpurun_out/prof_r02f
mkdir -p gpurun_out/r2t; O=gpurun_out/r2t
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --through-loop --no-cpu-baseline --no-roofline > $O/through_loop.json 2>> $O/bench.err
python bench.py --workload resnet50 --no-cpu-baseline --no-roofline > $O/resnet50.json 2>> $O/bench.err
python bench.py --workload hrnet_w48 --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $O/hrnet_w48.json 2>> $O/bench.err
python bench.py --path validate --no-cpu-baseline > $O/validate.json 2>> $O/bench.err
ADVMIX_FORCE_SYNC=1 python bench.py --no-cpu-baseline --no-roofline > $O/force_sync.json 2>> $O/bench.err
ADVMIX_DETERMINISTIC=1 python bench.py --no-cpu-baseline --no-roofline > $O/deterministic.json 2>> $O/bench.err
python tools/phase_times.py > $O/phase_times.log 2>&1
for f in bench through_loop resnet50 hrnet_w48 validate force_sync deterministic; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print('$f FAILED', e)
PY
done
bash tools/profile_step.sh r02f > $O/profile.log 2>&1; tail -5 $O/profile.log
