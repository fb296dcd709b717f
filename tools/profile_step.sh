#!/bin/bash
# rocprofv3 kernel trace + stats of the headline step, split per shape (run on the GPU box from the repo root):
#   tools/profile_step.sh <tag> [extra bench.py args]      ->  gpurun_out/prof_<tag>/{stats.csv, per_shape.csv, trace_summary.txt}
R=$PWD; TAG=$1; shift; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
python3 $R/bench.py --dump-shapes $OUT/shapes.csv "$@" > $OUT/dump.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o p -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $OUT/bench.log 2>&1
T=$(ls $OUT/raw/*/*kernel_trace.csv $OUT/raw/*kernel_trace.csv 2>/dev/null | head -1)
S=$(ls $OUT/raw/*/*kernel_stats.csv $OUT/raw/*kernel_stats.csv 2>/dev/null | head -1)
cp $S $OUT/stats.csv
python3 $R/tools/analyze_trace.py $T 0.5 > $OUT/trace_summary.txt 2>&1
python3 $R/tools/kernel_shapes.py $T $OUT/shapes.csv $OUT/per_shape.csv 0.5 > $OUT/per_shape.txt 2>&1
rm -rf $OUT/raw
tail -1 $OUT/bench.log | cut -c1-200; head -5 $OUT/trace_summary.txt; tail -3 $OUT/trace_summary.txt; head -30 $OUT/per_shape.txt; tail -1 $OUT/per_shape.txt
