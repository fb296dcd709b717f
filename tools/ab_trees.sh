#!/bin/bash
# Same-box comparison of an exported older tree (built under tools/_dbg/<name>tree:
#   mkdir -p tools/_dbg/r03tree && git archive <rev> | tar -x -C tools/_dbg/r03tree && (cd tools/_dbg/r03tree && python -m advmix_amd.build))
# with the current one, interleaved twice.   usage: tools/ab_trees.sh <out> <old name> "<workload> <steps>" ...
OUT=$1; OLD=$2; shift 2; : > $OUT; R=$PWD; SPECS=("$@")
for rep in 1 2; do
  for spec in "${SPECS[@]}"; do
    WL=${spec% *}; ST=${spec#* }
    for tree in $OLD now; do
      if [ $tree = now ]; then cd $R; EXTRA="--no-through-loop"; else cd $R/tools/_dbg/${OLD}tree; EXTRA="--no-through-loop"; fi
      python bench.py --workload $WL --no-cpu-baseline --no-roofline --steps $ST --warmup 5 $EXTRA 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tree $WL', d['value'], d['ms_per_step'])" >> $R/$OUT
      cd $R
    done
  done
done
cat $OUT
