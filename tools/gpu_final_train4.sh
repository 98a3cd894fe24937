# round-4 measurement set of the training step: bench lines (shipped widths, the reference's training widths, Updated bf16), kernel trace + launch sequence
#   bash tools/gpu_final_train4.sh <tag>     (through gpurun; every step under its own timeout)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r4h}
timeout 300 python tools/bench_train.py > gpurun_out/${T}_bench_train.json 2> gpurun_out/${T}_bench_train.err
timeout 300 python tools/bench_train.py --widths 64,128,256,512 --batch 1024 > gpurun_out/${T}_bench_train_w512.json 2>> gpurun_out/${T}_bench_train.err
timeout 300 python tools/bench_train.py --widths 128,256,512,1024 --batch 1024 > gpurun_out/${T}_bench_train_w1024.json 2>> gpurun_out/${T}_bench_train.err
timeout 300 python tools/bench_train.py --updated --dtype bf16 > gpurun_out/${T}_bench_train_updated_bf16.json 2>> gpurun_out/${T}_bench_train.err
timeout 300 python tools/bench_train.py --updated --dtype bf16 --widths 64,128,256,512 --batch 1024 --no-roofline > gpurun_out/${T}_bench_train_updated_bf16_w512.json 2>> gpurun_out/${T}_bench_train.err
timeout 300 python tools/bench_train.py --updated --no-roofline > gpurun_out/${T}_bench_train_updated.json 2>> gpurun_out/${T}_bench_train.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_ttrace -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline > gpurun_out/${T}_ttrace.log 2>&1
F=$(ls gpurun_out/${T}_ttrace/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $F 105 40 --seq > gpurun_out/${T}_train_seq.txt 2>&1
cp $(ls gpurun_out/${T}_ttrace/*/*kernel_stats.csv | head -1) gpurun_out/${T}_train_kernel_stats.csv
rm -rf gpurun_out/${T}_ttrace
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_wtrace -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline --widths 128,256,512,1024 --batch 1024 > gpurun_out/${T}_wtrace.log 2>&1
F=$(ls gpurun_out/${T}_wtrace/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $F 105 40 --seq > gpurun_out/${T}_train_seq_w1024.txt 2>&1
cp $(ls gpurun_out/${T}_wtrace/*/*kernel_stats.csv | head -1) gpurun_out/${T}_train_kernel_stats_w1024.csv
rm -rf gpurun_out/${T}_wtrace
sed -n 2,2p gpurun_out/${T}_train_seq.txt; sed -n 2,2p gpurun_out/${T}_train_seq_w1024.txt
