cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r2m_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2m_smoke.log 2>&1
python bench.py > gpurun_out/r2m_bench.json 2> gpurun_out/r2m_bench.err
cat gpurun_out/r2m_tests.log; tail -2 gpurun_out/r2m_smoke.log; cut -c1-260 gpurun_out/r2m_bench.json; tail -2 gpurun_out/r2m_bench.err
