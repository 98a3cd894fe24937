#!/bin/bash
# Round-6 end-state set, bench lines first (a fresh box): bash tools/gpu_final6.sh <tag>
#   bench lines (default with its nested legs, bf16, both wide widths, 10M tets, exact fp32 alone) -> gpurun_out/<tag>_bench*.json
#   training-step set of tools/gpu_round6.sh (six bench lines + four kernel traces)            -> gpurun_out/<tag>t/
#   pytest -m gpu (whole suite)                                                                 -> gpurun_out/<tag>_tests.log
T=${1:-r6v}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
DGNN_BENCH_TAG=$T timeout 400 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 300 python bench.py --dtype bf16 > gpurun_out/${T}_bench_bf16.json 2> gpurun_out/${T}_bench_bf16.err
timeout 200 python bench.py --widths 64,128,256,512 --no-train --no-extras > gpurun_out/${T}_bench_w512.json 2> gpurun_out/${T}_bench_w512.err
timeout 200 python bench.py --widths 128,256,512,1024 --no-train --no-extras > gpurun_out/${T}_bench_w1024.json 2> gpurun_out/${T}_bench_w1024.err
timeout 300 python bench.py --points 1485000 --steps 20 --warmup 10 --no-train --no-extras > gpurun_out/${T}_bench_10m.json 2> gpurun_out/${T}_bench_10m.err
timeout 200 python bench.py --gemm-mode f32 --no-train --no-extras > gpurun_out/${T}_bench_exact.json 2> gpurun_out/${T}_bench_exact.err
for f in gpurun_out/${T}_bench*.json; do python -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1])
r=j.get('roofline') or {}
print('$f', j.get('value'), j['ms_per_step'], j.get('ms_per_step_median'), str(r.get('kernel'))[:40], r.get('frac'), r.get('traffic'), (j.get('check') or {}).get('ok'))"; done
bash tools/gpu_round6.sh train ${T}t > gpurun_out/${T}_train.txt 2>&1; tail -n 20 gpurun_out/${T}_train.txt
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/${T}_tests.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests.log; tail -n 5 gpurun_out/${T}_tests.log
