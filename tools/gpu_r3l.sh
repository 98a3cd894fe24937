cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -q -x 2>&1 | tail -4
for a in "" "--updated" "--updated --dtype bf16" "--dtype bf16"; do
  python tools/bench_train.py $a --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['model'],d['dtype'],d['ms_per_step'],d['targets_per_s'],d['final_loss'])"
done
for a in "" "--updated" "--updated --dtype bf16"; do
  DGNN_TRAIN_AUX_STREAM=0 python tools/bench_train.py $a --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('aux=0', d['model'],d['dtype'],d['ms_per_step'],d['targets_per_s'],d['final_loss'])"
done
