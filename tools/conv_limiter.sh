#!/bin/bash
# What bounds conv_direct's THROUGHPUT?  Builds debug variants of the library with parts of the kernel
# compiled out (CD_DBG bits: 1 = no activation loads, 2 = no weight staging / LDS reads / barriers,
# 4 = no output stores) into tools/_dbg/ (run here, CPU), then `tools/conv_limiter.sh run` on the GPU box
# measures the 4-stream aggregate of the dominant conv for each.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/tools/_dbg
if [ "$1" != "run" ]; then
  mkdir -p $D
  for m in 1 2 3 4 7; do
    objs=""
    for f in conv_mfma conv_direct conv_wino wgrad_direct wgrad_lds wgrad_wino norm pointwise advmix_ops postproc inputpipe nms; do
      if [ $f = conv_direct ]; then
        /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -std=c++17 -DCD_DBG=$m -c $R/advmix_amd/csrc/$f.hip -o $D/$f.$m.o
        objs="$objs $D/$f.$m.o"
      else
        objs="$objs $R/advmix_amd/csrc/$f.o"
      fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libdbg$m.so $objs
    echo built $D/libdbg$m.so
  done
else
  echo "== full kernel"; python $R/tools/microbench_concurrent.py 32 conv
  for m in 1 2 3 4 7; do echo "== CD_DBG=$m"; ADVMIX_SO=$D/libdbg$m.so python $R/tools/microbench_concurrent.py 32 conv; done
fi
