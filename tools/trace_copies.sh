#!/bin/bash
# Who launches the step's non-library kernels?  rocprofv3 kernel trace of the step as benchmarked, then
# tools/trace_neighbours.py for __amd_rocclr_copyBuffer / fillBufferAligned / at::native - predecessor and successor on the
# same queue, grid sizes - over the last four replayed steps.   tools/trace_copies.sh <tag>  ->  gpurun_out/copies_<tag>/
R=$PWD; TAG=$1; shift; OUT=$R/gpurun_out/copies_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/raw
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o p -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-through-loop "$@" > $OUT/bench.log 2>&1
T=$(ls $OUT/raw/*/*kernel_trace.csv $OUT/raw/*kernel_trace.csv 2>/dev/null | head -1)
S=$(ls $OUT/raw/*/*kernel_stats.csv $OUT/raw/*kernel_stats.csv 2>/dev/null | head -1)
cp $S $OUT/stats_4lanes.csv
python3 $R/tools/analyze_trace.py $T 0.5 cat_views_kernel 4 > $OUT/trace_summary_4lanes.txt 2>&1
for k in copyBuffer fillBuffer at::native; do
  python3 $R/tools/trace_neighbours.py $T $k > $OUT/neighbours_$k.txt 2>&1
done
rm -rf $OUT/raw
head -3 $OUT/trace_summary_4lanes.txt; grep -A3 "kernels in flight" $OUT/trace_summary_4lanes.txt | head -3; grep "idle gaps" $OUT/trace_summary_4lanes.txt
for k in copyBuffer fillBuffer at::native; do head -30 $OUT/neighbours_$k.txt; done
