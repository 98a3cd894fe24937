#!/bin/bash
T=${1:-r6i}
mkdir -p gpurun_out
python tools/bench_gemm_train_shapes.py > gpurun_out/${T}_gemm_default.txt 2>&1
DGNN_X3_SMALL=0 DGNN_BF16_SMALL=0 python tools/bench_gemm_train_shapes.py > gpurun_out/${T}_gemm_tiled.txt 2>&1
DGNN_SMALL_SPLITK=0 python tools/bench_gemm_train_shapes.py > gpurun_out/${T}_gemm_nosplit.txt 2>&1
paste -d'|' gpurun_out/${T}_gemm_default.txt gpurun_out/${T}_gemm_tiled.txt gpurun_out/${T}_gemm_nosplit.txt | cut -c1-260
