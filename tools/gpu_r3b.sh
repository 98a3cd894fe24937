cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_scale.py -m gpu -q -x -k "x3_gemm" 2>&1 | tail -4
python tools/bench_gemm.py 2>&1 | tail -20
python bench.py --no-cpu-baseline --widths 64,128,256,512 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['breakdown_ms'], d['roofline']['achieved'], d['check'])"
python bench.py --no-cpu-baseline --widths 128,256,512,1024 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['breakdown_ms'], d['roofline']['achieved'], d['check'])"
