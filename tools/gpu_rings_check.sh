# ring parts: tests + a 2-rank and a 4-rank validation run of bench.py on the one GPU (gloo: ranks share the device; numbers of such a run are not a benchmark)
python -m pytest tests/test_gpu_infer.py tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -k "partition or ring or multi" > gpurun_out/r4r_test.log 2>&1
for n in 2 4; do
DGNN_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r4r_bench_n$n.json 2> gpurun_out/r4r_bench_n$n.err
echo "rc=$?" >> gpurun_out/r4r_bench_n$n.err
done
DGNN_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --halo exchange > gpurun_out/r4r_bench_n2x.json 2> gpurun_out/r4r_bench_n2x.err
echo "rc=$?" >> gpurun_out/r4r_bench_n2x.err
