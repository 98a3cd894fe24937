#!/bin/bash
T=${1:-r6j}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_train.py -x -q -m gpu > gpurun_out/${T}_tests.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q -m gpu -k "train or wgrad or bwd or backward or updated or golden or linear or gemm" > gpurun_out/${T}_tests2.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests2.log
python tools/bench_gemm_train_shapes.py > gpurun_out/${T}_gemm_default.txt 2>&1
for args in "" "--updated --dtype bf16" "--widths 128,256,512,1024 --batch 1024" "--updated --dtype bf16 --widths 128,256,512,1024 --batch 1024" "--widths 64,128,256,512 --batch 1024" "--updated --dtype bf16 --widths 64,128,256,512 --batch 1024"; do
  timeout 300 python tools/bench_train.py $args --steps 300 --warmup 300 --no-roofline >> gpurun_out/${T}_train.log 2>&1
done
for f in gpurun_out/${T}_tests.log gpurun_out/${T}_tests2.log; do tail -n 4 $f; done
cat gpurun_out/${T}_gemm_default.txt | cut -c1-100
grep -o '"model": "[^"]*", "dtype": "[^"]*"\|"ms_per_step": [0-9.]*\|"avg_block_tets": [0-9.]*' gpurun_out/${T}_train.log
