#!/bin/bash
# Build a variant of the library with extra defines on ONE source file (A/B experiments; run here, CPU):
#   tools/build_variant.sh <name> <source without .hip> "<-Dflags>"   ->  tools/_dbg/lib<name>.so   (use with ADVMIX_SO=...)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=$R/tools/_dbg; mkdir -p $D
NAME=$1; SRC=$2; DEFS=$3
objs=""
for f in conv_mfma conv_direct wgrad_direct wgrad_lds norm pointwise advmix_ops postproc inputpipe nms; do
  if [ $f = $SRC ]; then
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -std=c++17 -Wno-unused-result $DEFS -c $R/advmix_amd/csrc/$f.hip -o $D/$f.$NAME.o
    objs="$objs $D/$f.$NAME.o"
  else
    objs="$objs $R/advmix_amd/csrc/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib$NAME.so $objs
echo built $D/lib$NAME.so
