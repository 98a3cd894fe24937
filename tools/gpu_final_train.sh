# end-of-round measurement set of the training step: bench lines, the interleaved A/B of the fused launch chains, kernel trace + launch sequence
#   bash tools/gpu_final_train.sh <tag>     (through gpurun; every step under its own timeout)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r3t}
timeout 300 python tools/bench_train.py > gpurun_out/${T}_bench_train.json 2> gpurun_out/${T}_bench_train.err
timeout 300 python tools/bench_train.py --updated --dtype bf16 --no-roofline > gpurun_out/${T}_bench_train_updated_bf16.json 2>> gpurun_out/${T}_bench_train.err
timeout 300 python tools/ab_train.py fused=3 fused=1 fused=0 > gpurun_out/${T}_ab_train.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_ttrace -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline > gpurun_out/${T}_ttrace.log 2>&1
F=$(ls gpurun_out/${T}_ttrace/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $F 105 40 --seq > gpurun_out/${T}_train_seq.txt 2>&1
rm -rf gpurun_out/${T}_ttrace
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_utrace -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline --updated --dtype bf16 > gpurun_out/${T}_utrace.log 2>&1
F=$(ls gpurun_out/${T}_utrace/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $F 105 40 --seq > gpurun_out/${T}_train_seq_updated_bf16.txt 2>&1
rm -rf gpurun_out/${T}_utrace
timeout 300 python tools/bench_train.py --updated --no-roofline > gpurun_out/${T}_bench_train_updated.json 2>> gpurun_out/${T}_bench_train.err
tail -3 gpurun_out/${T}_ab_train.txt; sed -n 2,2p gpurun_out/${T}_train_seq.txt
