cd $GRAFT_REPO_ROOT
bash tools/prof_round2.sh r3h_f32
bash tools/prof_round2.sh r3h_bf16 --dtype bf16
python bench.py > gpurun_out/r3h_bench.json 2> gpurun_out/r3h_bench.err
python bench.py --dtype bf16 > gpurun_out/r3h_bench_bf16.json 2> gpurun_out/r3h_bench_bf16.err
ls gpurun_out | grep r3h | head -30
