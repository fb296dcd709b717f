#!/bin/bash
# rocprofv3 counter passes on the dominant conv (run on the GPU box from the repo root).
# Separate --pmc passes, kernel-trace only (no other trace domains), as the pool requires.
R=$PWD; OUT=$R/gpurun_out/pmc_$1; shift
cd /tmp && export TMPDIR=/tmp
ARGS="${CONV_ARGS:-32 32 64 48 32 3 1 1} ${MODE:-fwd} ${ITERS:-20}"   # CONV_ARGS="B Ci H W Co k s p" of tools/microbench_conv.py
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- python3 $R/tools/microbench_conv.py $ARGS > $OUT.sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/microbench_conv.py $ARGS > $OUT.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/microbench_conv.py $ARGS > $OUT.write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc -- python3 $R/tools/microbench_conv.py $ARGS > $OUT.tcc.log 2>&1
tail -1 $OUT.sq.log
