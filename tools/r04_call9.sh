#!/bin/bash
# round 4, GPU call 9: per-shape profile on the final library (grouped weight gradients now carry their shapes)
bash tools/profile_step.sh r04final > gpurun_out/profile_step_r04final.log 2>&1; tail -3 gpurun_out/profile_step_r04final.log | cut -c1-200
timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -k "grouped_weight or groups_its_weight" 2>&1 | tail -1
ls /sys/class/kfd/kfd/topology/nodes 2>&1 | head -3; python -c "
import sys; sys.path.insert(0,'.')
from advmix_amd.launch import visible_gpus, _kfd_gpu_nodes
print('kfd gpu nodes', _kfd_gpu_nodes(), 'visible', visible_gpus())"
