#!/bin/bash
# A/B of whole-library builds on ONE box: bash tools/gpu_ab_libs.sh <tag> <lib> [<lib> ...]   (libs: paths relative to the repo; "default" = the built one)
# every lib: the default bench line (no extras), 3 times interleaved; prints value, ms/step and the per-launch breakdown
T=${1:-ab}; shift
mkdir -p gpurun_out/$T
for rep in 1 2 3; do
  for L in "$@"; do
    if [ "$L" = default ]; then unset DGNN_LIB_PATH; else export DGNN_LIB_PATH=$PWD/$L; fi
    timeout 300 python bench.py --no-train --no-extras --no-cpu-baseline --steps 20 ${AB_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open('gpurun_out/bench_full.json')); print('$L', round(d['value']/1e6,2), d['ms_per_step'], (d.get('check') or {}).get('ok'), {k:round(v,3) for k,v in f['config'].get('replay_breakdown_ms',{}).items()})"
  done
done
