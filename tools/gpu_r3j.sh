cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -6
for rep in 1 2 3; do
for w in 1 0; do
  DGNN_TRAIN_WHOLE_MODEL=$w python tools/bench_train.py --steps 100 --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('whole=$w', d['model'],d['dtype'],d['ms_per_step'],d['final_loss'])"
done
done
