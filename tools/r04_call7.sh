#!/bin/bash
# round 4, GPU call 7: grouped weight gradients - microbenchmark, tests, in-step A/B
O=gpurun_out/r04f; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "grouped_weight or groups_its_weight" > $O/tests.log 2>&1; tail -3 $O/tests.log
for nb in 512 768 1024 1536; do echo "blocks $nb"; ADVMIX_WGRAD_GROUP_BLOCKS=$nb timeout 120 python tools/microbench_wgrad_group.py 8 2>&1 | grep -v amdgpu.ids; done > $O/microbench_wgrad_group.log 2>&1; cat $O/microbench_wgrad_group.log
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-through-loop"
for i in 1 2; do
  for g in 0 1; do
    ADVMIX_WGRAD_GROUP=$g timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group=$g', d['value'], d['ms_per_step'])"
  done
done > $O/ab_wgrad_group.log 2>&1; cat $O/ab_wgrad_group.log
for nb in 512 1024; do ADVMIX_WGRAD_GROUP_BLOCKS=$nb timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group blocks=$nb', d['value'], d['ms_per_step'])"; done >> $O/ab_wgrad_group.log 2>&1; tail -2 $O/ab_wgrad_group.log
ADVMIX_WGRAD_GROUP=1 timeout 300 python bench.py --workload hrnet_w48 --steps 15 --warmup 5 --no-cpu-baseline --no-roofline --no-through-loop 2>/dev/null | tail -1 | cut -c1-140
ADVMIX_WGRAD_GROUP=0 timeout 300 python bench.py --workload hrnet_w48 --steps 15 --warmup 5 --no-cpu-baseline --no-roofline --no-through-loop 2>/dev/null | tail -1 | cut -c1-140
