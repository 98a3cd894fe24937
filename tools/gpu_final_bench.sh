# bench-line part of tools/gpu_final.sh (re-taken when a run of the full set caught the box throttling its MFMA-heavy kernels after the PMC passes)
cd $GRAFT_REPO_ROOT
T=${1:-r9v}
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --dtype bf16 > gpurun_out/${T}_bench_bf16.json 2> gpurun_out/${T}_bench_bf16.err
DGNN_BF16_MODE=single python bench.py --dtype bf16 --no-train > gpurun_out/${T}_bench_bf16_single.json 2> /dev/null
python bench.py --widths 64,128,256,512 --no-train > gpurun_out/${T}_bench_w512.json 2> gpurun_out/${T}_bench_w512.err
python bench.py --widths 128,256,512,1024 --no-train > gpurun_out/${T}_bench_w1024.json 2> gpurun_out/${T}_bench_w1024.err
python bench.py --points 1485000 --steps 20 --warmup 10 --no-train > gpurun_out/${T}_bench_10m.json 2> gpurun_out/${T}_bench_10m.err
for f in gpurun_out/${T}_bench.json gpurun_out/${T}_bench_bf16.json gpurun_out/${T}_bench_bf16_single.json gpurun_out/${T}_bench_w512.json gpurun_out/${T}_bench_w1024.json gpurun_out/${T}_bench_10m.json; do python -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1])
r=j.get('roofline') or {}
print('$f', j.get('value'), j['ms_per_step'], str(r.get('kernel'))[:40], r.get('frac'), r.get('traffic'), (j.get('check') or {}).get('ok'), (j.get('training_step') or {}).get('ms_per_step'))"; done
