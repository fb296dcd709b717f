#!/bin/bash
# Experimental per-wave tile shapes (tools/_dbg/libexp.so: ADVMIX_CFG=7 64x32 / 8 32x64 / 9 64x64 per wave, 4 waves split K)
# against the shipped choices, on HRNet-W32's three lower-resolution branch convs.  usage: tools/exp_tiles.sh <out>
OUT=$1; : > $OUT
export ADVMIX_SO=tools/_dbg/libexp.so
for shp in "64 32 24" "128 16 12" "256 8 6"; do
  set -- $shp
  for cfg in 0 7 8 9; do
    for mode in fwd_stats dgrad_bnb; do
      if [ $cfg = 0 ]; then unset ADVMIX_CFG; else export ADVMIX_CFG=$cfg; fi
      python tools/exp_tiles.py 32 $1 $2 $3 $mode 300 | sed "s/^/cfg$cfg /" >> $OUT
    done
  done
done
cat $OUT
