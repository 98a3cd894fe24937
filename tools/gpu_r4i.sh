cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
for v in 1 0; do
 DGNN_X3_SMALL=$v python tools/bench_train.py --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('small=$v', d['model'],d['dtype'],d['ms_per_step'],d['final_loss'])"
 DGNN_X3_SMALL=$v python tools/bench_train.py --no-roofline --updated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('small=$v', d['model'],d['dtype'],d['ms_per_step'],d['final_loss'])"
done
done
