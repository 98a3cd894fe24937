# quick A/B of the fused layer: parity subset, phase trace, bench line with per-layer times
python -m pytest tests/test_gpu_parity.py -q -x -k "gemm_modes or partition or golden" 2>&1 | tail -2
for m in ${MODES:-f16x2}; do echo == $m; [ -n "$TRACE" ] && DGNN_GEMM_MODE=$m python tools/trace_fused.py 2>&1 | tail -5; for rep in 1 2; do python bench.py --gemm-mode $m --steps 30 --warmup 5 --no-cpu-baseline --no-train 2>&1 | tail -1 > gpurun_out/f16_$m.json; python - <<PY
import json
d=json.load(open("gpurun_out/f16_$m.json"))
print("$m", d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d["config"].get("breakdown_ms"))
PY
done; done
