#!/bin/bash
# Round-5 measurements committed under profiles/ (GPU box, repo root): bench lines of the headline and the secondary
# configurations, same-box A/B against ADVMIX_WINO=0, PMC traffic of the dominant members, MFMA-busy counter of the step.
#   tools/collect_r05.sh <tag>  ->  gpurun_out/<tag>/
T=${1:-r05}; O=gpurun_out/$T; mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench.err
for i in 1 2; do
  ADVMIX_WINO=0 python bench.py --no-cpu-baseline --no-roofline --no-through-loop --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A direct kernels only (ADVMIX_WINO=0)  %.2f images/s  %.3f ms' % (d['value'], d['ms_per_step']))" >> $O/ab_winograd.log
  python bench.py --no-cpu-baseline --no-roofline --no-through-loop --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B Winograd branches (default)          %.2f images/s  %.3f ms' % (d['value'], d['ms_per_step']))" >> $O/ab_winograd.log
done
for wl in resnet50 hrnet_w48 hrnet_w32_512; do
  python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $O/${wl}_bench_line.json 2>> $O/bench.err
done
python bench.py --path validate --no-cpu-baseline > $O/validate_bench_line.json 2>> $O/bench.err
python bench.py --path inputs --no-cpu-baseline > $O/inputs_bench_line.json 2>> $O/bench.err
python bench.py --path nms --no-cpu-baseline > $O/nms_bench_line.json 2>> $O/bench.err
ADVMIX_DETERMINISTIC=1 python bench.py --no-cpu-baseline --no-roofline --no-through-loop > $O/deterministic_bench_line.json 2>> $O/bench.err
ADVMIX_FORCE_SYNC=1 python bench.py --no-cpu-baseline --no-roofline > $O/force_sync_1rank_bench_line.json 2>> $O/bench.err
ADVMIX_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/dp2_shared_gpu_bench_line.json 2>> $O/bench.err
python tools/phase_times.py > $O/phase_times.log 2>&1
tools/pmc_dominant.sh $T > $O/pmc_dominant.log 2>&1; cp gpurun_out/pmc_$T/pmc_dominant.json gpurun_out/pmc_$T/pmc_wino32.json $O/ 2>/dev/null
MS=$(python -c "import json; print(json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1])['ms_per_step'])")
bash tools/pmc_step.sh $T 3 $MS > $O/pmc_step.log 2>&1          # (utilisation against the UNPROFILED step time of this run)
for f in bench_line resnet50_bench_line hrnet_w48_bench_line hrnet_w32_512_bench_line validate_bench_line inputs_bench_line nms_bench_line deterministic_bench_line force_sync_1rank_bench_line dp2_shared_gpu_bench_line; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'), d.get('step_frac_of_fp32_mfma_peak'), d.get('grad_exchange_verified'), d.get('replicas_identical'))
except Exception as e:
    print('$f FAILED', e)
PY
done
cat $O/ab_winograd.log; tail -12 $O/pmc_dominant.log; tail -5 $O/pmc_step.log
