cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r2p_tests.log
cat gpurun_out/r2p_tests.log
PREFETCH=0 python tools/diag_train.py 40 > gpurun_out/r2p_diag_nopf.log 2>&1; tail -12 gpurun_out/r2p_diag_nopf.log
PREFETCH=1 python tools/diag_train.py 40 > gpurun_out/r2p_diag_pf.log 2>&1; tail -12 gpurun_out/r2p_diag_pf.log
python tools/bench_train.py --steps 60 > gpurun_out/r2p_train.json 2> gpurun_out/r2p_train.err
cut -c1-400 gpurun_out/r2p_train.json; tail -2 gpurun_out/r2p_train.err
