#!/bin/bash
# round 4, GPU call 4: the whole GPU suite on the final tree; the N-rank verification through RCCL (one rank) and gloo (two ranks on
# one GPU); the two-rank test repeated; a full default bench line
O=gpurun_out/r04d; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
B="python -X faulthandler bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
ADVMIX_FORCE_SYNC=1 timeout 400 $B > $O/bench_force_sync.json 2> $O/bench_force_sync.err; echo "force_sync rc=$?"; tail -4 $O/bench_force_sync.err | cut -c1-300; tail -1 $O/bench_force_sync.json | cut -c1-200
ADVMIX_BENCH_SHARE_GPU=1 timeout 600 $B --gpus 2 --steps 10 --warmup 3 > $O/bench_share2.json 2> $O/bench_share2.err; echo "share2 rc=$?"; tail -4 $O/bench_share2.err | cut -c1-300
for i in 1 2 3 4 5; do
  timeout 400 python -m pytest tests/test_models_gpu.py -x -q -k "two_ranks_on_one_gpu" > $O/two_rank_$i.log 2>&1; tail -1 $O/two_rank_$i.log
done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; tail -1 $O/bench_default.json | cut -c1-300
