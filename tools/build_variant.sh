#!/bin/bash
# Build a variant of the library with extra defines on ONE source file (A/B experiments; run here, CPU):
#   tools/build_variant.sh <name> <source without .hip> "<-Dflags>"   ->  tools/_dbg/lib<name>.so   (use with ADVMIX_SO=...)
# The measurement variants of conv_direct (-DCD_DBG=<bits> / -DCD_PRELOAD / -DCD_CLK / -DCD_NO_PRE) are NOT in the shipped
# source (round 3): tools/variants/conv_direct_dbg.patch adds them to a scratch copy under tools/_dbg/ before compiling;
# likewise -DWL_DBG=<bits> of wgrad_lds (round 4: tools/variants/wgrad_lds_dbg.patch).  advmix_build_flags() != 0 in both.
# (If the patch no longer applies after a kernel change: re-create it from the last revision it applied to.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=$R/tools/_dbg; mkdir -p $D
NAME=$1; SRC=$2; DEFS=$3
objs=""
for f in conv_mfma conv_direct conv_wino conv_wino4 conv_smap conv_pw wgrad_direct wgrad_lds wgrad_wino norm pointwise advmix_ops postproc inputpipe nms; do
  if [ $f = $SRC ]; then
    in=$R/advmix_amd/csrc/$f.hip
    if [ $f = conv_direct ] && echo "$DEFS" | grep -q "CD_DBG\|CD_PRELOAD\|CD_CLK\|CD_NO_PRE"; then
      cp $in $D/conv_direct.hip && cp $R/advmix_amd/csrc/common.h $D/ && patch -s $D/conv_direct.hip $R/tools/variants/conv_direct_dbg.patch
      in=$D/conv_direct.hip; DEFS="$DEFS -I$R/advmix_amd/csrc -I$R/include"
    fi
    if [ $f = wgrad_lds ] && echo "$DEFS" | grep -q "WL_DBG"; then      # (round 4: these switches left the shipped source too)
      cp $in $D/wgrad_lds.hip && cp $R/advmix_amd/csrc/common.h $D/ && patch -s $D/wgrad_lds.hip $R/tools/variants/wgrad_lds_dbg.patch
      in=$D/wgrad_lds.hip; DEFS="$DEFS -I$R/advmix_amd/csrc -I$R/include"
    fi
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -std=c++17 -Wno-unused-result $DEFS -c $in -o $D/$f.$NAME.o
    objs="$objs $D/$f.$NAME.o"
  else
    objs="$objs $R/advmix_amd/csrc/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib$NAME.so $objs
echo built $D/lib$NAME.so
