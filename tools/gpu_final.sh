# end-of-round measurement set: rocprofv3 passes (fp32, bf16), bench lines (default, bf16, bf16 single, wide, 10M), training traces + bench lines
cd $GRAFT_REPO_ROOT
T=${1:-r9w}
bash tools/prof_round2.sh ${T}_f32
bash tools/prof_round2.sh ${T}_bf16 --dtype bf16
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --dtype bf16 > gpurun_out/${T}_bench_bf16.json 2> gpurun_out/${T}_bench_bf16.err
DGNN_BF16_MODE=single python bench.py --dtype bf16 --no-train > gpurun_out/${T}_bench_bf16_single.json 2> /dev/null
python bench.py --widths 64,128,256,512 --no-train > gpurun_out/${T}_bench_w512.json 2> gpurun_out/${T}_bench_w512.err
python bench.py --widths 128,256,512,1024 --no-train > gpurun_out/${T}_bench_w1024.json 2> gpurun_out/${T}_bench_w1024.err
python bench.py --points 1485000 --steps 20 --warmup 10 --no-train > gpurun_out/${T}_bench_10m.json 2> gpurun_out/${T}_bench_10m.err
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in "" "--updated" "--updated --dtype bf16"; do
tag=${T}_train$(echo $a | tr -d ' -')
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline $a > gpurun_out/$tag.log 2>&1
python tools/trace_gaps.py gpurun_out/$tag/*/*kernel_trace.csv 105 40 > gpurun_out/$tag.gaps.txt
done
for a in "" "--updated" "--updated --dtype bf16" "--dtype bf16"; do
  python tools/bench_train.py $a 2>/dev/null > gpurun_out/${T}_bench_train$(echo $a | tr -d ' -').json
done
python tools/ab_train.py whole=1 whole=0 composite=0 > gpurun_out/${T}_ab_train.txt 2>&1
for f in gpurun_out/${T}_bench*.json; do python -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1])
r=j.get('roofline') or {}
print('$f', j.get('value', j.get('targets_per_s')), j['ms_per_step'], str(r.get('kernel'))[:40], r.get('frac'), (j.get('check') or {}).get('ok'))"; done
tail -3 gpurun_out/${T}_ab_train.txt
