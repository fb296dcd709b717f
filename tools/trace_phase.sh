#!/bin/bash
# tools/trace_phase.sh <tag> <phase> [workload]: kernel trace of one phase's graph replays -> gpurun_out/<tag>_<phase>_trace.txt
R=$PWD; TAG=$1; PH=$2; shift 2; OUT=$R/gpurun_out; mkdir -p $OUT
EXC=${EXC:-}        # optional: first row of the excerpt (tools/trace_excerpt.py's own default when unset; was read without ever being set)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/raw_ph
rocprofv3 --kernel-trace --output-format csv -d /tmp/raw_ph -o p -- python3 $R/tools/trace_phase.py $PH "$@" > $OUT/${TAG}_${PH}.log 2>&1
T=$(ls /tmp/raw_ph/*/*kernel_trace.csv /tmp/raw_ph/*kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/analyze_trace.py $T 0.5 cos_kernel 8 > $OUT/${TAG}_${PH}_trace.txt 2>&1
python3 $R/tools/trace_excerpt.py $T cos_kernel $EXC > $OUT/${TAG}_${PH}_excerpt.txt 2>&1
head -14 $OUT/${TAG}_${PH}_trace.txt; grep -A3 "kernels in flight" $OUT/${TAG}_${PH}_trace.txt
