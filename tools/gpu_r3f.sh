cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -q -x 2>&1 | tail -6
for a in "--updated" "--updated --dtype bf16" "" "--updated" "--updated --dtype bf16"; do
  python tools/bench_train.py --steps 80 $a --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['model'],d['dtype'],d['block_builder'],d['ms_per_step'],d['final_loss'])"
done
