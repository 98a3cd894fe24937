python -m pytest tests/test_gpu_parity.py -q -x -k "gemm_modes or partition or golden" 2>&1 | tail -5
for m in ${MODES:-bf16x3f f16x2}; do echo == $m; DGNN_GEMM_MODE=$m python tools/trace_fused.py 2>&1 | tail -5; python bench.py --gemm-mode $m --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/f16_$m.json; python - <<PY
import json
d=json.load(open("gpurun_out/f16_$m.json"))
print("$m", d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), (d.get("check") or {}).get("max_abs_err"), (d.get("check") or {}).get("rms_err"), d["config"].get("breakdown_ms"))
PY
done
