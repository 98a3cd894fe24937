#!/bin/bash
T=${1:-r6c}
mkdir -p gpurun_out
timeout 300 python tools/dbg_wide_batch2.py > gpurun_out/${T}_dbg_wide2.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_dbg_wide2.log
timeout 600 python -m pytest tests/test_gpu_config3.py -x -q -m gpu -s > gpurun_out/${T}_config3.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_config3.log
timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "updated_training_step" > gpurun_out/${T}_scale.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_scale.log
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for V in tiny full; do
  A=""; [ $V = tiny ] && A="--points 3000 --batch 8"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_trace_$V -- python3 tools/bench_train.py $A --steps 60 --warmup 100 --no-roofline > gpurun_out/${T}_trace_$V.log 2>&1
  F=$(ls gpurun_out/${T}_trace_$V/*/*kernel_trace.csv | head -1)
  python tools/trace_gaps.py $F 105 40 --seq > gpurun_out/${T}_seq_$V.txt 2>&1
  rm -rf gpurun_out/${T}_trace_$V
done
tail -30 gpurun_out/${T}_dbg_wide2.log; tail -15 gpurun_out/${T}_config3.log; tail -3 gpurun_out/${T}_scale.log; sed -n 1,3p gpurun_out/${T}_seq_tiny.txt
