cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 1 0; do
export DGNN_WGRAD_FINE=$v
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4l_f$v -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline > gpurun_out/r4l_f$v.log 2>&1
echo "DGNN_WGRAD_FINE=$v"; tail -1 gpurun_out/r4l_f$v.log | cut -c1-80
python tools/trace_gaps.py gpurun_out/r4l_f$v/*/*kernel_trace.csv 105 40 | sed -n 2,2p
python tools/trace_gaps.py gpurun_out/r4l_f$v/*/*kernel_trace.csv 105 40 | sed -n '/kernel time/,$p' | grep "wgrad"
done
