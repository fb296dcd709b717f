#!/bin/bash
# HBM traffic (rocprofv3 PMC: separate FETCH_SIZE / WRITE_SIZE passes, kernel-trace only) of (a) bench.py's roofline.dominant -
# the member with the largest weight x time: 3x3 256->256 @8x6 forward + BatchNorm sums (csrc/conv_smap.hip since r05z) - and (b) the Winograd
# kernel at 32->32 @64x48 forward + sums.  usage (GPU box, repo root): tools/pmc_dominant.sh <tag>  ->  gpurun_out/pmc_<tag>/*.json
R=$PWD; OUT=$R/gpurun_out/pmc_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/dom_$C -- python3 $R/tools/microbench_wino.py 32 20 256:fwd+sums > $OUT/dom_$C.log 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/wino_$C -- python3 $R/tools/microbench_wino.py 32 20 32:fwd+sums > $OUT/wino_$C.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
def counter(sub, pat):
    fs = glob.glob('%s/%s/*/*_counter_collection.csv' % (out, sub)) + glob.glob('%s/%s/*_counter_collection.csv' % (out, sub))
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(fs[0])) if pat in r['Kernel_Name']]
    return sum(v) / len(v), len(v)
res = {}
for tag, pat, name, algo in (('dom', 'conv_smap', '3x3 s1 256->256 @8x6 fwd+BN-sums', 2 * 32 * 8 * 6 * 256 * 4 + 9 * 256 * 256 * 4),
                             ('wino', 'conv_wino', '3x3 s1 32->32 @64x48 fwd+BN-sums', 2 * 32 * 64 * 48 * 32 * 4 + 16 * 32 * 32 * 4)):
    f, n = counter(tag + '_FETCH_SIZE', pat)
    w, _ = counter(tag + '_WRITE_SIZE', pat)
    rb, wb = f * 1024 * 2, w * 1024      # KiB; FETCH_SIZE counts 128-B requests as 64 B on gfx950 (MI355X_MICROARCH.md): doubled
    res[tag] = {'kernel': name, 'launches_averaged': n, 'hbm_read_bytes_corrected': rb, 'hbm_write_bytes': wb,
                'hbm_bytes_per_launch': rb + wb, 'algorithmic_bytes_per_launch': algo, 'traffic_ratio': (rb + wb) / algo}
json.dump(res['dom'], open(out + '/pmc_dominant.json', 'w'), indent=1)
json.dump(res['wino'], open(out + '/pmc_wino32.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
