#!/bin/bash
T=${1:-r6e}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_trainer_cpu.py -x -q -m gpu > gpurun_out/${T}_tests.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train or bn or batch or trainer or golden or f3 or f2" > gpurun_out/${T}_tests2.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests2.log
for args in "--points 3000 --batch 8" "--points 3000 --batch 8 --updated --dtype bf16" "" "--updated --dtype bf16" "--widths 128,256,512,1024 --batch 1024" "--updated --dtype bf16 --widths 128,256,512,1024 --batch 1024"; do
  timeout 300 python tools/bench_train.py $args --steps 300 --warmup 300 --no-roofline >> gpurun_out/${T}_train.log 2>&1
done
tail -3 gpurun_out/${T}_tests.log gpurun_out/${T}_tests2.log; grep -o '"model": "[^"]*", "dtype": "[^"]*"\|"ms_per_step": [0-9.]*\|"host_issue_ms_per_step": [0-9.]*\|"avg_block_tets": [0-9.]*' gpurun_out/${T}_train.log
