# samples the GPU's clocks / power while the fused layers run back to back (is the part power-limited in these kernels?)
python tools/variants.py --rounds 40 --reps 20 dgnn_amd/libdgnn_hip.so > gpurun_out/clock_probe_variants.txt 2>&1 &
BP=$!
sleep 25
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use|fclk" | tr '\n' ' '; echo
  sleep 1
done
wait $BP
tail -3 gpurun_out/clock_probe_variants.txt
