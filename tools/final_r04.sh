#!/bin/bash
# the round's last GPU call on the final tree: GPU suite, the driver's bench command, the default bench line
O=gpurun_out/r04_final; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.log 2>&1; tail -1 $O/bench_driver_form.log > $O/bench_driver_form.json; tail -1 $O/bench_driver_form.json | cut -c1-200
python bench.py > $O/bench_line.log 2>&1; tail -1 $O/bench_line.log > $O/bench_line.json; tail -1 $O/bench_line.json | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
