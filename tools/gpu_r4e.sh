cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4e_train -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline > gpurun_out/r4e_train.log 2>&1
python tools/trace_gaps.py gpurun_out/r4e_train/*/*kernel_trace.csv 105 40 | sed -n 1,3p
python tools/trace_gaps.py gpurun_out/r4e_train/*/*kernel_trace.csv 105 40 | sed -n '/kernel time/,$p' | head -34
