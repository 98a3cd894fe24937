cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
DGNN_TRAIN_AUX_STREAM=0 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "composite" 2>&1 | tail -2
python tools/ab_train.py whole=1 whole=0 2>&1 | tail -3
DGNN_TRAIN_AUX_STREAM=0 python tools/ab_train.py whole=1 whole=0 2>&1 | tail -3
