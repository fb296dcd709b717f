#!/bin/bash
# round 4, GPU call 8: fill-based slices of the grouped weight gradients; sibling-stream experiment
O=gpurun_out/r04g; mkdir -p $O
timeout 120 python tools/microbench_wgrad_group.py 8 2>&1 | grep -v amdgpu.ids > $O/microbench_wgrad_group_fill.log; cat $O/microbench_wgrad_group_fill.log
timeout 120 python tools/microbench_wgrad_group.py 4 2>&1 | grep -v amdgpu.ids >> $O/microbench_wgrad_group_fill.log; tail -5 $O/microbench_wgrad_group_fill.log
timeout 400 python -m pytest tests/test_ops_gpu.py -x -q -k "grouped_weight or groups_its_weight or chain_bn or deterministic_mode" > $O/tests.log 2>&1; tail -2 $O/tests.log
ADVMIX_WGRAD_SIBLING=1 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -x -q -k "groups_its_weight or chain_bn or forward_backward_vs_oracle or graph_runner_matches or one_rank_real_rccl" > $O/tests_sibling.log 2>&1; tail -2 $O/tests_sibling.log
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-through-loop"
for i in 1 2; do
  for g in 0 1; do
    ADVMIX_WGRAD_SIBLING=$g timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sibling=$g', d['value'], d['ms_per_step'])"
  done
done > $O/ab_wgrad_sibling.log 2>&1; cat $O/ab_wgrad_sibling.log
ADVMIX_WGRAD_GROUP=0 timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group=0', d['value'], d['ms_per_step'])" | tee -a $O/ab_wgrad_sibling.log
for wl in resnet50 hrnet_w48; do for g in 0 1; do ADVMIX_WGRAD_GROUP=$g timeout 300 python bench.py --workload $wl --steps 15 --warmup 5 --no-cpu-baseline --no-roofline --no-through-loop 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl group=$g', d['value'], d['ms_per_step'])"; done; done | tee -a $O/ab_wgrad_sibling.log
