#!/bin/bash
# Round-4 evidence in one GPU call (run from the repo root on the GPU box): GPU suite, bench lines, N-rank self-verification,
# knock-outs, phase times, rocprofv3 kernel summaries, PMC passes, the round-3 tree beside this one.
TAG=${1:-r04}; O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
for i in 1 2 3 4; do timeout 400 python -m pytest tests/test_models_gpu.py -q -k "two_ranks_on_one_gpu" > $O/two_rank_repeat_$i.log 2>&1; tail -1 $O/two_rank_repeat_$i.log; done
python bench.py > $O/bench_line.log 2>&1; tail -1 $O/bench_line.log > $O/bench_line.json
for wl in resnet50 hrnet_w48; do
  python bench.py --workload $wl --steps 20 --no-cpu-baseline > $O/${wl}.log 2>&1; tail -1 $O/${wl}.log > $O/${wl}_bench_line.json
done
ADVMIX_DETERMINISTIC=1 python bench.py --no-cpu-baseline --no-roofline --no-through-loop > $O/det.log 2>&1; tail -1 $O/det.log > $O/deterministic_bench_line.json
ADVMIX_FORCE_SYNC=1 python bench.py --no-cpu-baseline --no-roofline > $O/sync.log 2>&1; tail -1 $O/sync.log > $O/force_sync_1rank_bench_line.json
ADVMIX_FORCE_SYNC=1 python bench.py --exec eager --no-cpu-baseline --no-roofline --steps 20 > $O/sync_eager.log 2>&1; tail -1 $O/sync_eager.log > $O/force_sync_1rank_eager_bench_line.json
ADVMIX_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/share2.log 2>&1; tail -1 $O/share2.log > $O/dp2_shared_gpu_bench_line.json
python bench.py --path nms --no-cpu-baseline > $O/nms.log 2>&1; tail -1 $O/nms.log > $O/nms_bench_line.json
python tools/knockout.py 30 > $O/knockout.log 2>&1; cat $O/knockout.log | tail -8
python tools/phase_times.py > $O/phase_times.log 2>&1; tail -8 $O/phase_times.log
bash tools/ab_trees.sh $O/ab_round3_tree_vs_round4.log r03 "hrnet_w32 30" "resnet50 30" "hrnet_w48 15" > /dev/null 2>&1; cat $O/ab_round3_tree_vs_round4.log
bash tools/profile_step.sh $TAG > $O/profile_step.log 2>&1; tail -4 $O/profile_step.log
MODE=fwd_stats bash tools/pmc_conv.sh ${TAG}_conv32_epi > $O/pmc_epi.log 2>&1
MODE=dgrad_bnb bash tools/pmc_conv.sh ${TAG}_conv32_dgrad_bnb > $O/pmc_bnb.log 2>&1
python tools/summarize_pmc.py gpurun_out/pmc_${TAG}_conv32_epi $O/pmc_conv32_epi.json > /dev/null 2>&1
python tools/summarize_pmc.py gpurun_out/pmc_${TAG}_conv32_dgrad_bnb $O/pmc_conv32_dgrad_bnb.json $((32*64*48*32*4*4 + 32*64*48*32/4 + 32*9*32*4)) > /dev/null 2>&1
bash tools/pmc_step.sh $TAG 3 $(python -c "import json; print(json.load(open('$O/bench_line.json'))['ms_per_step'])") > $O/pmc_step.log 2>&1;   # (utilisation against THIS call's step time)
 cp gpurun_out/pmc_step_$TAG.json $O/ 2>/dev/null
rm -rf gpurun_out/pmc_${TAG}_conv32_epi gpurun_out/pmc_${TAG}_conv32_dgrad_bnb gpurun_out/pmc_step_${TAG} gpurun_out/pmc_step_${TAG}_fetch gpurun_out/pmc_step_${TAG}_write
for f in $O/*_bench_line.json $O/bench_line.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'), d.get('grad_exchange_verified'), d.get('replicas_identical'))"; done
python -c "import json; d=json.load(open('$O/pmc_conv32_dgrad_bnb.json')); print('bnb traffic', d['hbm_bytes_per_launch'], d['traffic_ratio'])"
