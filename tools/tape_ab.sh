#!/bin/bash
# A/B of the two step runners (one multi-branch HIP graph per segment vs per-lane tapes), same box.
mkdir -p gpurun_out/r2o; cd $GRAFT_REPO_ROOT
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout -s ABRT 150 python -X faulthandler bench.py --steps 30 --warmup 8 > gpurun_out/r2o/b_$tag.json 2> gpurun_out/r2o/b_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r2o/b_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])
except Exception as e:
    print('$tag failed', e); print(open('gpurun_out/r2o/b_$tag.err').read()[-3500:])
PY
}
run L4_T0 ADVMIX_LANES=4 ADVMIX_TAPE=0
run L4_T1_Q8 ADVMIX_LANES=4 ADVMIX_TAPE=1 GPU_MAX_HW_QUEUES=8
run L4_T0_Q8 ADVMIX_LANES=4 ADVMIX_TAPE=0 GPU_MAX_HW_QUEUES=8
run L2_T0 ADVMIX_LANES=2 ADVMIX_TAPE=0
run L2_T1_Q8 ADVMIX_LANES=2 ADVMIX_TAPE=1 GPU_MAX_HW_QUEUES=8
run L1_T0 ADVMIX_LANES=1 ADVMIX_TAPE=0
run L1_T1_Q8 ADVMIX_LANES=1 ADVMIX_TAPE=1 GPU_MAX_HW_QUEUES=8
