#!/bin/bash
# Same-box comparison of the round-2 tree (git 72f4536, exported to tools/_dbg/r02tree and built there:
#   mkdir -p tools/_dbg/r02tree && git archive 72f4536 | tar -x -C tools/_dbg/r02tree && (cd tools/_dbg/r02tree && python -m advmix_amd.build))
# with the current one.   usage: tools/ab_rounds.sh <out> "<workload> <steps>" ...      (each configuration twice, interleaved)
OUT=$1; shift; : > $OUT; R=$PWD; SPECS=("$@")
for rep in 1 2; do
  for spec in "${SPECS[@]}"; do
    WL=${spec% *}; ST=${spec#* }
    for tree in r02 r03; do
      if [ $tree = r02 ]; then cd $R/tools/_dbg/r02tree; else cd $R; fi
      python bench.py --workload $WL --no-cpu-baseline --no-roofline --steps $ST --warmup 5 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tree $WL', d['value'], d['ms_per_step'])" >> $R/$OUT
      cd $R
    done
  done
done
cat $OUT
