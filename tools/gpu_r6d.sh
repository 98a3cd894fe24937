#!/bin/bash
T=${1:-r6d}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wide.py tests/test_gpu_config3.py -x -q -m gpu -s > gpurun_out/${T}_tests.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests.log
timeout 300 python tools/host_profile_train.py > gpurun_out/${T}_host_static.log 2>&1
timeout 300 python tools/host_profile_train.py --updated > gpurun_out/${T}_host_updated.log 2>&1
tail -4 gpurun_out/${T}_tests.log; grep "un-profiled" gpurun_out/${T}_host_*.log
