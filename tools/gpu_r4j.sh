cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 1 0; do
export DGNN_X3_SMALL=$v
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4j_small$v -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline > gpurun_out/r4j_small$v.log 2>&1
echo "DGNN_X3_SMALL=$v"
python tools/trace_gaps.py gpurun_out/r4j_small$v/*/*kernel_trace.csv 105 40 | sed -n 2,2p
python tools/trace_gaps.py gpurun_out/r4j_small$v/*/*kernel_trace.csv 105 40 | grep "k_linear_fwd"
done
