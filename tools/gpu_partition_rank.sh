# per-rank cost of the partitioned forward on one GPU (tools/bench_partition_rank.py)
python -m pytest tests/test_gpu_infer.py -x -q -k "partitioned or ring" > gpurun_out/r4p_test.log 2>&1
for w in 2 4 8; do
  python tools/bench_partition_rank.py --world $w --ranks 0,$((w-1)) --halo recompute > gpurun_out/r4q_world${w}_rings.json 2> gpurun_out/r4q_world${w}_rings.err
done
python tools/bench_partition_rank.py --world 8 --ranks 0,1,2,3,4,5,6,7 --halo recompute --steps 100 > gpurun_out/r4q_world8_rings_all.json 2>&1
python tools/bench_partition_rank.py --world 8 --ranks 0 --halo recompute --one-call 0 > gpurun_out/r4q_world8_rings_chain.json 2>&1
python tools/bench_partition_rank.py --world 8 --ranks 0 --halo recompute --dtype bf16 > gpurun_out/r4q_world8_rings_bf16.json 2>&1
