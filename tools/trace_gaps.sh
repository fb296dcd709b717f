#!/bin/bash
# 4-lane kernel trace of the headline step -> gpurun_out/<tag>_trace_gaps.txt (tools/analyze_trace.py incl. the idle-gap tables)
R=$PWD; TAG=${1:-gaps}; shift; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/raw_gaps
rocprofv3 --kernel-trace --output-format csv -d /tmp/raw_gaps -o p -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $OUT/${TAG}_bench.log 2>&1
T=$(ls /tmp/raw_gaps/*/*kernel_trace.csv /tmp/raw_gaps/*kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/analyze_trace.py $T 0.5 cat_views_kernel 4 > $OUT/${TAG}_trace_gaps.txt 2>&1
tail -75 $OUT/${TAG}_trace_gaps.txt
