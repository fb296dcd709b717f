#!/bin/bash
# rocprofv3 kernel trace + stats of the headline step (run on the GPU box from the repo root):
#   tools/profile_step.sh <tag> [extra bench.py args]   ->  gpurun_out/prof_<tag>/
#     stats_4lanes.csv, trace_summary_4lanes.txt   the step as benchmarked (4 launch lanes): per-template totals, busy
#                                                   fraction, kernels in flight
#     stats_1lane.csv, per_shape_1lane.csv          ADVMIX_LANES=1 (serialized): per-SHAPE attribution - the library logs
#                                                   every MFMA launch's shape (bench.py --dump-shapes) and
#                                                   tools/kernel_shapes.py joins it with the trace
R=$PWD; TAG=$1; shift; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in 4 1; do
  export ADVMIX_LANES=$L
  rm -rf $OUT/raw
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o p -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-through-loop "$@" > $OUT/bench_${L}lane.log 2>&1
  T=$(ls $OUT/raw/*/*kernel_trace.csv $OUT/raw/*kernel_trace.csv 2>/dev/null | head -1)
  S=$(ls $OUT/raw/*/*kernel_stats.csv $OUT/raw/*kernel_stats.csv 2>/dev/null | head -1)
  cp $S $OUT/stats_${L}lane.csv
  if [ $L = 4 ]; then
    python3 $R/tools/analyze_trace.py $T 0.5 cat_views_kernel 4 > $OUT/trace_summary_4lanes.txt 2>&1   # the last 4 whole replayed steps
  else
    python3 $R/bench.py --dump-shapes $OUT/shapes.csv "$@" > $OUT/dump.log 2>&1
    python3 $R/tools/kernel_shapes.py $T $OUT/shapes.csv $OUT/per_shape_1lane.csv 0.5 > $OUT/per_shape_1lane.txt 2>&1
  fi
done
rm -rf $OUT/raw
unset ADVMIX_LANES
grep -o '"ms_per_step": [0-9.]*' $OUT/bench_4lane.log $OUT/bench_1lane.log; head -1 $OUT/trace_summary_4lanes.txt; tail -2 $OUT/trace_summary_4lanes.txt; head -40 $OUT/per_shape_1lane.txt; tail -1 $OUT/per_shape_1lane.txt
