#!/bin/bash
# round 6: (1) where inference_layer_batch at wide widths goes wrong; (2) the training step's floor: a tiny scene (every kernel at its fixed cost) vs the bench scene
T=${1:-r6b}
mkdir -p gpurun_out
timeout 300 python tools/dbg_wide_batch.py > gpurun_out/${T}_dbg_wide.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_dbg_wide.log
for args in "--points 3000 --batch 8" "--points 3000 --batch 8 --updated --dtype bf16" "" "--updated --dtype bf16"; do
  timeout 300 python tools/bench_train.py $args --steps 300 --warmup 300 --no-roofline >> gpurun_out/${T}_train_floor.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${T}_trace_tiny -o tiny -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --points 3000 --batch 8 --steps 60 --warmup 100 --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/${T}_trace_tiny.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $(ls gpurun_out/${T}_trace_tiny/*kernel_trace.csv gpurun_out/${T}_trace_tiny/*/*kernel_trace.csv 2>/dev/null | head -1) 105 40 --seq > gpurun_out/${T}_trace_tiny_gaps.txt 2>&1
rm -rf gpurun_out/${T}_trace_tiny/*/*.db gpurun_out/${T}_trace_tiny/*.db
tail -12 gpurun_out/${T}_dbg_wide.log; grep -o '"model": "[^"]*", "dtype": "[^"]*"\|"ms_per_step": [0-9.]*\|"host_issue_ms_per_step": [0-9.]*\|"avg_block_tets": [0-9.]*' gpurun_out/${T}_train_floor.log
