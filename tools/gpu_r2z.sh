cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
PREFETCH=1 python tools/diag_train.py 60 2>/dev/null | tail -5
for rep in 1 2 3; do
for pf in none stream; do
  python tools/bench_train.py --steps 80 --prefetch $pf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['model'],d['dtype'],d['block_builder'],d['ms_per_step'],d['final_loss'])"
done
done
for a in "--updated" "--updated --dtype bf16" "--dtype bf16"; do
  python tools/bench_train.py --steps 80 $a --prefetch stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['model'],d['dtype'],d['block_builder'],d['ms_per_step'],d['final_loss'])"
done
