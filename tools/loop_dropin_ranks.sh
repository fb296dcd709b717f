#!/bin/bash
# the N-rank drop-in (reference tools/train.py main() on the mirrors) in a loop -
# 2 ranks x $1 (default 50), 8 ranks x $2 (default 20), each on a fresh OUTPUT_DIR.  Build container only (needs /root/reference).
#   tools/loop_dropin_ranks.sh [n2] [n8]  ->  prints one line per run and the tally
N2=${1:-50}; N8=${2:-20}; ok2=0; ok8=0
for i in $(seq 1 $N2); do
  if python -m pytest tests/test_dropin_cpu.py -x -q -k two_ranks 2>&1 | tail -1 | grep -q "1 passed"; then ok2=$((ok2+1)); else echo "2-rank run $i FAILED"; fi
done
echo "2 ranks: $ok2 / $N2 passed"
for i in $(seq 1 $N8); do
  if python -m pytest tests/test_dropin_cpu.py -x -q -k eight_ranks 2>&1 | tail -1 | grep -q "1 passed"; then ok8=$((ok8+1)); else echo "8-rank run $i FAILED"; fi
done
echo "8 ranks: $ok8 / $N8 passed"
