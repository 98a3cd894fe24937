cd $GRAFT_REPO_ROOT
bash tools/prof_round2.sh r2j_f32
bash tools/prof_round2.sh r2j_bf16 --dtype bf16
python bench.py > gpurun_out/r2j_bench.json 2> gpurun_out/r2j_bench.err
python bench.py --dtype bf16 > gpurun_out/r2j_bench_bf16.json 2> gpurun_out/r2j_bench_bf16.err
DGNN_BF16_MODE=single python bench.py --dtype bf16 > gpurun_out/r2j_bench_bf16_single.json 2> /dev/null
python bench.py --widths 64,128,256,512 > gpurun_out/r2j_bench_w512.json 2> gpurun_out/r2j_bench_w512.err
python bench.py --widths 128,256,512,1024 > gpurun_out/r2j_bench_w1024.json 2> gpurun_out/r2j_bench_w1024.err
python bench.py --points 1485000 --steps 5 --warmup 2 > gpurun_out/r2j_bench_10m.json 2> gpurun_out/r2j_bench_10m.err
for f in gpurun_out/r2j_bench*.json; do python -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', j['value'], j['ms_per_step'], j['roofline']['kernel'][:40], j['roofline']['frac'], (j['check'] or {}).get('ok'))"; done
