# kernel trace of the training step (Static fp32 by default) + the launch sequence of one step
#   bash tools/gpu_train_trace.sh <tag> [bench_train.py args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-tt}; shift
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_ttrace -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline "$@" > gpurun_out/${T}_ttrace.log 2>&1
F=$(ls gpurun_out/${T}_ttrace/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $F 105 40 --seq > gpurun_out/${T}_train_seq.txt 2>&1
rm -rf gpurun_out/${T}_ttrace
timeout 300 python tools/bench_train.py "$@" > gpurun_out/${T}_train.json 2> gpurun_out/${T}_train.err
