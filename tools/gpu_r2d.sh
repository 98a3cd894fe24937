cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2d_tests.log
python bench.py --widths 64,128,256,512 > gpurun_out/r2d_bench_w512.json 2> gpurun_out/r2d_bench_w512.err
python bench.py --widths 128,256,512,1024 > gpurun_out/r2d_bench_w1024.json 2> gpurun_out/r2d_bench_w1024.err
for a in "" "--updated" "--dtype bf16" "--updated --dtype bf16"; do python tools/bench_train.py $a >> gpurun_out/r2d_train.json 2>> gpurun_out/r2d_train.err; done
DGNN_GEMM_MODE=f32 python tools/bench_train.py >> gpurun_out/r2d_train.json 2>> gpurun_out/r2d_train.err
cat gpurun_out/r2d_tests.log; for f in gpurun_out/r2d_bench_w512.json gpurun_out/r2d_bench_w1024.json; do python -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['breakdown_ms'], j['roofline']['kernel'], j['roofline']['frac'], j['check'])"; done; cat gpurun_out/r2d_train.json; tail -3 gpurun_out/r2d_train.err
