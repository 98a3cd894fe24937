python -m pytest tests -m gpu -q 2>&1 | tail -2
bash tools/prof_round2.sh r9w_f32
bash tools/prof_round2.sh r9w_bf16 --dtype bf16
bash tools/gpu_final_bench.sh r9v 2>&1 | tail -8
