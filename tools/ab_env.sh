#!/bin/bash
# Same-box A/B of environment switches on the step: tools/ab_env.sh <out file> <workload> <steps> "<VAR=val ...>" ["<VAR=val ...>" ...]
# Each configuration runs twice, interleaved (A B A B) to see the box's drift.
OUT=$1; WL=$2; STEPS=$3; shift 3
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg python bench.py --workload $WL --no-cpu-baseline --no-roofline --steps $STEPS --warmup 5 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$WL [$cfg]', d['value'], d['ms_per_step'])" >> $OUT
  done
done
cat $OUT
