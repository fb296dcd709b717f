#!/bin/bash
# Round-3 evidence in one GPU call: full GPU test suite, the profiling recipe, one bench line per workload / path.
TAG=${1:-r03f}; O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log; tail -3 $O/pytest_all.log
python bench.py > $O/bench_line.log 2>&1; tail -1 $O/bench_line.log > $O/bench_line.json
for wl in resnet50 hrnet_w48 hrnet_w32_512; do
  python bench.py --workload $wl --steps 20 > $O/${wl}.log 2>&1; tail -1 $O/${wl}.log > $O/${wl}_bench_line.json
done
python bench.py --path validate > $O/validate.log 2>&1; tail -1 $O/validate.log > $O/validate_bench_line.json
python bench.py --through-loop --no-cpu-baseline --no-roofline > $O/loop.log 2>&1; tail -1 $O/loop.log > $O/through_loop_bench_line.json
ADVMIX_DETERMINISTIC=1 python bench.py --no-cpu-baseline --no-roofline > $O/det.log 2>&1; tail -1 $O/det.log > $O/deterministic_bench_line.json
ADVMIX_FORCE_SYNC=1 python bench.py --no-cpu-baseline --no-roofline > $O/sync.log 2>&1; tail -1 $O/sync.log > $O/force_sync_1rank_bench_line.json
python bench.py --path inputs --no-cpu-baseline > $O/inputs.log 2>&1; tail -1 $O/inputs.log > $O/inputs_bench_line.json
python bench.py --path nms > $O/nms.log 2>&1; tail -1 $O/nms.log > $O/nms_bench_line.json
python tools/phase_times.py > $O/phase_times.log 2>&1
bash tools/profile_step.sh $TAG > $O/profile_step.log 2>&1
for f in $O/*_bench_line.json $O/bench_line.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'))"; done
tail -5 $O/phase_times.log; tail -4 $O/profile_step.log
