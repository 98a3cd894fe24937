python -m pytest tests -m gpu -q 2>&1 | tail -2
bash tools/prof_round2.sh r9t_f32
bash tools/prof_round2.sh r9t_bf16 --dtype bf16
bash tools/gpu_final_bench.sh r9s 2>&1 | tail -8
