cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in 16384 32768 131072; do
export DGNN_X3_SMALL_M=$m
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4k_m$m -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline > gpurun_out/r4k_m$m.log 2>&1
echo "DGNN_X3_SMALL_M=$m"
python tools/trace_gaps.py gpurun_out/r4k_m$m/*/*kernel_trace.csv 105 40 | sed -n 2,2p
python tools/trace_gaps.py gpurun_out/r4k_m$m/*/*kernel_trace.csv 105 40 | sed -n '/kernel time/,$p' | grep "k_linear_fwd"
done
