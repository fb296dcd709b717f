#!/bin/bash
# memory-path counters for the dominant conv (separate passes, kernel-trace only).
# No TA_* set: those counters hang rocprofv3 on this pool (tools/README.md).
R=$PWD; OUT=$R/gpurun_out/pmcm_$1; shift
cd /tmp && export TMPDIR=/tmp
ARGS="${ARGS:-32 32 64 48 32 3 1 1 fwd 20}"
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/microbench_conv.py $ARGS > $OUT.p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$OUT/p*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print('%-40s %.4g' % (k, sum(v)/len(v)))
PY
