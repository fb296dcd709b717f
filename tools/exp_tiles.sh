#!/bin/bash
# A forced conv_direct tile configuration (ADVMIX_CFG=<n>; 7 = eight waves: two row tiles x four-way K split, weights staged
# once) against the shipped choice on HRNet-W32's lower-resolution branch convs: time + checksums.
#   usage: tools/exp_tiles.sh <out> [cfgs="0 7"] [library]      (configurations 8+ need a scratch build)
OUT=$1; CFGS=${2:-"0 7"}; : > $OUT
if [ -n "$3" ]; then export ADVMIX_SO=$3; fi
for shp in "64 32 24" "128 16 12" "256 8 6"; do
  set -- $shp
  for cfg in $CFGS; do
    for mode in fwd_stats dgrad_bnb; do
      if [ $cfg = 0 ]; then unset ADVMIX_CFG; else export ADVMIX_CFG=$cfg; fi
      python tools/exp_tiles.py 32 $1 $2 $3 $mode 300 | sed "s/^/cfg$cfg /" >> $OUT
    done
  done
done
cat $OUT
