#!/bin/bash
# What bounds conv_direct's THROUGHPUT?  Debug variants of the library with parts of the kernel compiled out (CD_DBG bits:
# 1 = no activation loads, 2 = no weight staging / LDS reads / barriers, 4 = no output stores; the switches live in
# tools/variants/conv_direct_dbg.patch) are built by tools/build_variant.sh into tools/_dbg/ (run here, CPU), then
# `tools/conv_limiter.sh run [B Ci H W Co k s p]` on the GPU box times one conv configuration with each of them
# (default: the 3x3 s1 32 -> 32 @64x48 branch conv at B = 32, forward + BatchNorm sums, on the direct kernel).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/tools/_dbg
if [ "$1" != "run" ]; then
  for m in 1 2 3 4 7; do $R/tools/build_variant.sh dbg$m conv_direct "-DCD_DBG=$m"; done
else
  shift
  CFG=${@:-32 32 64 48 32 3 1 1}
  echo "== full kernel"; ADVMIX_WINO=0 python $R/tools/microbench_conv.py $CFG fwd_stats
  for m in 1 2 3 4 7; do echo "== CD_DBG=$m"; ADVMIX_WINO=0 ADVMIX_SO=$D/libdbg$m.so python $R/tools/microbench_conv.py $CFG fwd_stats; done
fi
