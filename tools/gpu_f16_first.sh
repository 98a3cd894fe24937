python -m pytest tests/test_gpu_parity.py -q -x -k "gemm_modes" -s 2>&1 | tail -15
for m in bf16x3f f16x2d f16x2; do python bench.py --gemm-mode $m --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/f16_$m.json; python - <<PY
import json
d=json.load(open("gpurun_out/f16_$m.json"))
print("$m", d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("check",{}).get("max_abs_err"), d.get("check",{}).get("rms_err"), d.get("breakdown_ms") or d.get("layers_ms") or "")
PY
done
