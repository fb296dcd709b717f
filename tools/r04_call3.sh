#!/bin/bash
# round 4, GPU call 3: which use of the NULL stream breaks the replays; where bench.py's verification crashed; the mask epilogue
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python tools/dp_graph_repro.py --out $O/dp_repro_nullstream.jsonl --runs 2 --replays 6 \
   --only r2_l1_product_chk r2_l1_null_chk r2_l1_loopstream r2_l4_loopstream p1_l1_gloo1_nullchk > $O/dp_repro_nullstream.log 2>&1
grep -h SUMMARY $O/dp_repro_nullstream.log
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "bn_backward_epilogue or grouped_conv or chain_bn or deterministic_mode_keeps or batch_norm or fuse_layer" > $O/gpu_tests_bnb.log 2>&1; tail -3 $O/gpu_tests_bnb.log
B="python -X faulthandler bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
ADVMIX_FORCE_SYNC=1 timeout 400 $B --no-through-loop > $O/bench_force_sync.json 2> $O/bench_force_sync.err; echo "force_sync rc=$?"; tail -25 $O/bench_force_sync.err | cut -c1-300
timeout 400 $B > $O/bench_through.json 2> $O/bench_through.err; echo "through rc=$?"; tail -12 $O/bench_through.err | cut -c1-300; tail -1 $O/bench_through.json | cut -c1-300
