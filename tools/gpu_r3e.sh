cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in "--updated --dtype bf16" "--updated"; do
tag=$(echo $a | tr -d ' -')
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3e_$tag -- python3 tools/bench_train.py --steps 40 --warmup 5 $a > gpurun_out/r3e_$tag.log 2>&1
tail -1 gpurun_out/r3e_$tag.log | cut -c1-300
python tools/trace_gaps.py gpurun_out/r3e_$tag/*/*kernel_trace.csv | sed -n 1,3p
python tools/trace_gaps.py gpurun_out/r3e_$tag/*/*kernel_trace.csv | sed -n '/kernel time/,$p' | head -28
done
