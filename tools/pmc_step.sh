#!/bin/bash
# MFMA utilisation of the WHOLE step from hardware counters (north_star: "rocprof reports ... MFMA utilisation against gfx950
# peak"): one rocprofv3 --pmc pass (kernel trace only, as the pool requires) over a few eager steps; the counters are per
# kernel dispatch, so serialisation under the profiler does not change them.   usage: tools/pmc_step.sh <tag> [steps=3] [step_ms of the unprofiled replayed step]
R=$PWD; TAG=$1; STEPS=${2:-3}; MS=${3:-}; OUT=$R/gpurun_out/pmc_step_$TAG; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 $R/bench.py --exec eager --steps $STEPS --warmup 1 --no-cpu-baseline --no-roofline --no-through-loop > $OUT.log 2>&1
# memory-side traffic of the step: FETCH_SIZE and WRITE_SIZE need a pass each (TCC counter slots)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${OUT}_fetch -- python3 $R/bench.py --exec eager --steps $STEPS --warmup 1 --no-cpu-baseline --no-roofline --no-through-loop > ${OUT}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${OUT}_write -- python3 $R/bench.py --exec eager --steps $STEPS --warmup 1 --no-cpu-baseline --no-roofline --no-through-loop > ${OUT}_write.log 2>&1
cd $R
python3 tools/summarize_pmc_step.py $OUT $STEPS 1 gpurun_out/pmc_step_$TAG.json $MS
