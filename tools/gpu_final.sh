# end-of-round measurement set: rocprofv3 passes (fp32, bf16), bench lines (default, bf16, wide, 10M), wide-layer trace
#   bash tools/gpu_final.sh <tag>     (through gpurun; every step under its own timeout)
cd $GRAFT_REPO_ROOT
T=${1:-r3f}
timeout 600 bash tools/prof_round2.sh ${T}_f32 --no-extras
timeout 600 bash tools/prof_round2.sh ${T}_bf16 --dtype bf16 --no-extras
timeout 400 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 300 python bench.py --dtype bf16 > gpurun_out/${T}_bench_bf16.json 2> gpurun_out/${T}_bench_bf16.err
timeout 200 python bench.py --widths 64,128,256,512 --no-train --no-extras > gpurun_out/${T}_bench_w512.json 2> gpurun_out/${T}_bench_w512.err
timeout 200 python bench.py --widths 128,256,512,1024 --no-train --no-extras > gpurun_out/${T}_bench_w1024.json 2> gpurun_out/${T}_bench_w1024.err
timeout 300 python bench.py --points 1485000 --steps 20 --warmup 10 --no-train --no-extras > gpurun_out/${T}_bench_10m.json 2> gpurun_out/${T}_bench_10m.err
for f in gpurun_out/${T}_bench*.json; do python -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1])
r=j.get('roofline') or {}
print('$f', j.get('value'), j['ms_per_step'], str(r.get('kernel'))[:40], r.get('frac'), (j.get('check') or {}).get('ok'))"; done
