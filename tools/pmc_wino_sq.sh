#!/bin/bash
# Where a Winograd conv launch spends its SIMD cycles (rocprofv3 PMC, two SQ passes + GRBM, kernel-trace only):
#   tools/pmc_wino_sq.sh <tag> [C:role = 32:fwd+sums]   ->  gpurun_out/pmc_<tag>/wino_sq.json
R=$PWD; OUT=$R/gpurun_out/pmc_$1; CASE=${2:-32:fwd+sums}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/sq_a -- python3 $R/tools/microbench_wino.py 32 20 $CASE > $OUT/sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC \
  --output-format csv -d $OUT/sq_b -- python3 $R/tools/microbench_wino.py 32 20 $CASE > $OUT/sq_b.log 2>&1
python3 - $OUT "$CASE" <<'PY'
import csv, glob, json, sys, collections
out, case = sys.argv[1], sys.argv[2]
vals = {}
dur = []
for sub in ('sq_a', 'sq_b'):
    fs = glob.glob('%s/%s/*/*_counter_collection.csv' % (out, sub)) + glob.glob('%s/%s/*_counter_collection.csv' % (out, sub))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'conv_wino' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        vals[k] = sum(v) / len(v)
    ts = glob.glob('%s/%s/*/*_kernel_trace.csv' % (out, sub)) + glob.glob('%s/%s/*_kernel_trace.csv' % (out, sub))
    if sub == 'sq_a':
        dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(ts[0])) if 'conv_wino' in r['Kernel_Name']]
us = sum(dur) / len(dur) / 1e3
cyc = vals.get('GRBM_GUI_ACTIVE', 0) / 8.0                  # summed over the 8 XCDs
simd_cycles = cyc * 1024
wc = 4.0 * vals.get('SQ_WAVE_CYCLES', 0)                    # counted in quad-cycles
res = {'case': case, 'kernel_us_under_pmc': round(us, 2), 'clock_ghz': round(cyc / us / 1e3, 3), 'raw': vals,
       'waves_per_simd_avg': round(wc / simd_cycles, 3),
       'mfma_busy_of_simd_cycles': round(vals.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd_cycles, 3),
       'of_wave_cycles': {k: round(4.0 * vals.get(n, 0) / wc, 3) for k, n in (
           ('parked (waitcnt / barrier)', 'SQ_WAIT_ANY'), ('issue stall', 'SQ_WAIT_INST_ANY'), ('issuing', 'SQ_ACTIVE_INST_ANY'),
           ('issue stall on LDS', 'SQ_WAIT_INST_LDS'), ('VALU issuing', 'SQ_ACTIVE_INST_VALU'), ('LDS issuing', 'SQ_ACTIVE_INST_LDS'),
           ('VMEM issuing', 'SQ_ACTIVE_INST_VMEM'), ('misc issuing', 'SQ_ACTIVE_INST_MISC'))},
       'lds_idx_active_of_cu_cycles': round(vals.get('SQ_LDS_IDX_ACTIVE', 0) / (cyc * 256), 3),
       'lds_bank_conflict_of_cu_cycles': round(vals.get('SQ_LDS_BANK_CONFLICT', 0) / (cyc * 256), 3)}
json.dump(res, open(out + '/wino_sq.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
