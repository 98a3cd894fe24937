cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in "" "--updated" "--updated --dtype bf16"; do
tag=r3i_train$(echo $a | tr -d ' -')
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 tools/bench_train.py --steps 60 --warmup 8 --no-roofline $a > gpurun_out/$tag.log 2>&1
python tools/trace_gaps.py gpurun_out/$tag/*/*kernel_trace.csv 15 40 > gpurun_out/$tag.gaps.txt
head -3 gpurun_out/$tag.gaps.txt
done
for a in "" "--updated" "--updated --dtype bf16" "--dtype bf16"; do
  python tools/bench_train.py --steps 100 $a 2>/dev/null > gpurun_out/r3i_bench_train$(echo $a | tr -d ' -').json
  python -c "import sys,json; d=json.loads(open('gpurun_out/r3i_bench_train$(echo $a | tr -d ' -').json').read()); print(d['model'],d['dtype'],d['ms_per_step'],d['targets_per_s'],d['final_loss'])"
done
