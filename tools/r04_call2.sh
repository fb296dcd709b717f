#!/bin/bash
# round 4, GPU call 2: what triggers the NULL-stream replay failure; the shipped (own-stream) runner; A/B of the replay stream
O=gpurun_out/r04b; mkdir -p $O
timeout 1100 python tools/dp_graph_repro.py --out $O/dp_repro_triggers.jsonl --runs 2 --replays 6 \
   --only p1_l1_hostrt_hiprio p1_l1_gloo1 p2_l1_devmul p1_l1_noise r2_l4_pkt0 > $O/dp_repro_triggers.log 2>&1
timeout 900 python tools/dp_graph_repro.py --out $O/dp_repro_product.jsonl --runs 4 --replays 10 \
   --only r2_l1_product r2_l4_product > $O/dp_repro_product.log 2>&1
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-through-loop"
for i in 1 2; do
  ADVMIX_REPLAY_STREAM=own  timeout 300 $B > $O/bench_own_$i.json 2> $O/bench_own_$i.err
  ADVMIX_REPLAY_STREAM=null timeout 300 $B > $O/bench_null_$i.json 2> $O/bench_null_$i.err
done
ADVMIX_REPLAY_STREAM=null DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 $B > $O/bench_null_pkt0.json 2> $O/bench_null_pkt0.err
ADVMIX_FORCE_SYNC=1 timeout 400 $B > $O/bench_force_sync.json 2> $O/bench_force_sync.err
ADVMIX_BENCH_SHARE_GPU=1 timeout 600 $B --gpus 2 --steps 10 --warmup 3 > $O/bench_share2.json 2> $O/bench_share2.err
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
tail -3 $O/gpu_tests.log; grep -h SUMMARY $O/*.log; for f in $O/bench_*.json; do echo $f; tail -1 $f | cut -c1-400; done
