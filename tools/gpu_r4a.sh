cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_scale.py tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -4
python tools/dbg_gemm.py 2>/dev/null | tail -2
DGNN_X3_N64=0 python tools/dbg_gemm.py 2>/dev/null | tail -1
python bench.py --no-cpu-baseline --no-train --widths 64,128,256,512 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['breakdown_ms'])"
python tools/ab_train.py whole=1 2>&1 | tail -1
DGNN_X3_N64=0 python tools/ab_train.py whole=1 2>&1 | tail -1
