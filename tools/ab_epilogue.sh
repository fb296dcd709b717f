#!/bin/bash
# A/B of two builds of the library on one box: per-kernel microbenchmarks of the four HRNet-W32 branch convs in the
# variants the step launches, then the step itself.   usage: tools/ab_epilogue.sh <out dir> <old .so>
OUT=$1; OLD=$2; mkdir -p $OUT
for lib in new old; do
  if [ $lib = old ]; then export ADVMIX_SO=$OLD; else unset ADVMIX_SO; fi
  for shp in "32 64 48" "64 32 24" "128 16 12" "256 8 6"; do
    set -- $shp
    for mode in fwd_stats fwd dgrad_bt_add dgrad_bnb; do
      python tools/microbench_conv.py 32 $1 $2 $3 $1 3 1 1 $mode 300 | sed "s/^/$lib /" >> $OUT/micro.log
    done
  done
done
sort -k3,3 -k4,4 -k8,8 -s $OUT/micro.log > $OUT/micro_sorted.log
for lib in new old new old; do
  if [ $lib = old ]; then export ADVMIX_SO=$OLD; else unset ADVMIX_SO; fi
  python bench.py --no-cpu-baseline --no-roofline --steps 40 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d.get('INVALID_variant_build_flags'))" >> $OUT/step.log
done
cat $OUT/step.log
