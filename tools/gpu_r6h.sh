#!/bin/bash
T=${1:-r6h}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for A in "" "--updated --dtype bf16" "--widths 128,256,512,1024 --batch 1024" "--updated --dtype bf16 --widths 128,256,512,1024 --batch 1024"; do
  i=$((i+1))
  timeout 300 python tools/bench_train.py $A --steps 300 --warmup 300 --no-roofline >> gpurun_out/${T}_train.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_trace_$i -- python3 tools/bench_train.py $A --steps 60 --warmup 100 --no-roofline > gpurun_out/${T}_trace_$i.log 2>&1
  F=$(ls gpurun_out/${T}_trace_$i/*/*kernel_trace.csv | head -1)
  python tools/trace_gaps.py $F 105 40 --seq > gpurun_out/${T}_seq_$i.txt 2>&1
  rm -rf gpurun_out/${T}_trace_$i
done
grep -o '"model": "[^"]*", "dtype": "[^"]*"\|"ms_per_step": [0-9.]*\|"avg_block_tets": [0-9.]*' gpurun_out/${T}_train.log
