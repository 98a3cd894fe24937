cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/dbg_gemm3.py 2>/dev/null | tail -5
DGNN_X3_BIG=0 python tools/dbg_gemm3.py 2>/dev/null | tail -5
python bench.py --no-cpu-baseline --no-train --widths 64,128,256,512 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['breakdown_ms'])"
