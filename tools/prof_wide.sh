# kernel-trace of the wide-layer bench lines (aggregate + GEMM pairs; gemm mode f16x2 -> k_linear_fwd_x2h_big for layers wider than 256)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for wd in 64,128,256,512 128,256,512,1024; do
  tag=r9r_wide_$(echo $wd | tr ',' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 bench.py --widths $wd --steps 3 --warmup 1 --no-cpu-baseline --no-train > gpurun_out/$tag.log 2>&1
  python - <<PY
import glob, pandas as pd
f = glob.glob("gpurun_out/$tag/*/*kernel_stats.csv")[0]
d = pd.read_csv(f)
print("== --widths $wd")
print(d[["Name", "Calls", "AverageNs", "Percentage"]].head(12).to_string(index=False))
PY
done
