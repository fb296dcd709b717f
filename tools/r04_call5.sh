#!/bin/bash
# round 4, GPU call 5: what else on the NULL stream breaks the replays; is the chains-vs-levels difference noise or the new epilogue
O=gpurun_out/r04e; mkdir -p $O
timeout 300 python tools/diag_chains.py > $O/diag_chains.log 2>&1; cat $O/diag_chains.log | tail -8
timeout 600 python tools/dp_graph_repro.py --out $O/dp_repro_item.jsonl --runs 2 --replays 6 --only r2_l1_product_item r2_l1_product_fold > $O/dp_repro_item.log 2>&1
grep -h SUMMARY $O/dp_repro_item.log
for i in 1 2 3; do timeout 400 python -m pytest tests/test_models_gpu.py -x -q -k "two_ranks_on_one_gpu or launch_chains" > $O/tests_$i.log 2>&1; tail -1 $O/tests_$i.log; done
grep -h "Error\|assert " $O/tests_1.log | head -8
