#!/bin/bash
# Round-3 counter passes (run on the GPU box from the repo root): the dominant forward (conv + BN sums) with and without
# the XCD-aware row-tile order, and the BatchNorm-backward input gradient.  Separate --pmc passes, kernel trace only.
R=$PWD
for tag in fwd_remap1 fwd_remap0 bnb_remap1; do
  case $tag in
    fwd_remap1) export ADVMIX_XCD_REMAP=1; M=fwd_stats; ALGO=25202688;;
    fwd_remap0) export ADVMIX_XCD_REMAP=0; M=fwd_stats; ALGO=25202688;;
    bnb_remap1) export ADVMIX_XCD_REMAP=1; M=dgrad_bnb; ALGO=$((5 * 12582912 + 36864));;
  esac
  MODE=$M bash tools/pmc_conv.sh r03_$tag
  python3 tools/summarize_pmc.py gpurun_out/pmc_r03_$tag gpurun_out/r03_pmc_$tag.json $ALGO > /dev/null
  python3 -c "import json; d=json.load(open('gpurun_out/r03_pmc_$tag.json')); print('$tag', d['kernel'][:60], 'read MB %.1f write MB %.1f ratio %.3f L2 hit %.3f mfma busy %.3f' % (d['hbm_read_bytes_corrected']/1e6, d['hbm_write_bytes']/1e6, d['traffic_ratio'], d['l2_hit_rate'], d['mfma_busy_frac_of_wave_cycles']))"
done
