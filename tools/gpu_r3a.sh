cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; cut -c1-900 gpurun_out/r3a_bench.json
python bench.py --no-cpu-baseline --widths 64,128,256,512 > gpurun_out/r3a_bench_w512.json 2> gpurun_out/r3a_bench_w512.err; cut -c1-1500 gpurun_out/r3a_bench_w512.json
python bench.py --no-cpu-baseline --widths 128,256,512,1024 > gpurun_out/r3a_bench_w1024.json 2> gpurun_out/r3a_bench_w1024.err; cut -c1-1500 gpurun_out/r3a_bench_w1024.json
