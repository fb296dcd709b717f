#!/bin/bash
# HBM traffic (rocprofv3 PMC: separate FETCH_SIZE / WRITE_SIZE passes, kernel-trace only) of the candidates for bench.py's
# roofline.dominant - the forward + BatchNorm-sums members of the four branch widths' kernels (the member with the largest
# weight x time is one of them: 256 -> 256 @8x6 on csrc/conv_smap.hip, or a Winograd one) - as ONE file keyed by member name.
# usage (GPU box, repo root): tools/pmc_dominant.sh <tag>  ->  gpurun_out/pmc_<tag>/pmc_dominant.json (+ pmc_wino32.json)
R=$PWD; OUT=$R/gpurun_out/pmc_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  for W in 256 128 64 32; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/c${W}_$C -- python3 $R/tools/microbench_wino.py 32 20 $W:fwd+sums > $OUT/c${W}_$C.log 2>&1
  done
done
python3 - $OUT <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def counter(sub, pat):
    fs = glob.glob('%s/%s/*/*_counter_collection.csv' % (out, sub)) + glob.glob('%s/%s/*_counter_collection.csv' % (out, sub))
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(fs[0])) if pat in r['Kernel_Name']]
    return sum(v) / len(v), len(v)
res = {}
for C, H, W, pat, U in ((256, 8, 6, 'conv_smap', 16), (128, 16, 12, 'conv_wino', 16), (64, 32, 24, 'conv_wino', 16), (32, 64, 48, 'conv_wino', 16)):
    name = '3x3 s1 %d->%d @%dx%d fwd+BN-sums' % (C, C, H, W)
    algo = 2 * 32 * H * W * C * 4 + 9 * C * C * 4           # input + output once, the 3x3 filters once
    f, n = counter('c%d_FETCH_SIZE' % C, pat)
    w, _ = counter('c%d_WRITE_SIZE' % C, pat)
    rb, wb = f * 1024 * 2, w * 1024      # KiB; FETCH_SIZE counts 128-B requests as 64 B on gfx950 (MI355X_MICROARCH.md): doubled
    res[name] = {'kernel': name, 'launches_averaged': n, 'hbm_read_bytes_corrected': rb, 'hbm_write_bytes': wb,
                 'hbm_bytes_per_launch': rb + wb, 'algorithmic_bytes_per_launch': algo, 'traffic_ratio': (rb + wb) / algo}
json.dump({'members': res}, open(out + '/pmc_dominant.json', 'w'), indent=1)
json.dump(res['3x3 s1 32->32 @64x48 fwd+BN-sums'], open(out + '/pmc_wino32.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
