#!/bin/bash
# round 6, first GPU call: the tests this round's ADVICE fixes touch + the default bench line (headline size, 10M leg, wall clock)
T=${1:-r6a}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_reorder.py -x -q -m gpu > gpurun_out/${T}_tests1.log 2>&1; echo "tests1 rc=$?" >> gpurun_out/${T}_tests1.log
timeout 1200 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "updated_training_step" -s > gpurun_out/${T}_tests2.log 2>&1; echo "tests2 rc=$?" >> gpurun_out/${T}_tests2.log
S=$(date +%s)
DGNN_BENCH_TAG=${T} timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s" >> gpurun_out/${T}_bench.err
tail -3 gpurun_out/${T}_tests1.log gpurun_out/${T}_tests2.log; tail -2 gpurun_out/${T}_bench.err; wc -c gpurun_out/${T}_bench.json
