cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mode in AD SMALL; do
if [ $mode = AD ]; then export DGNN_X3_AD=1; else export DGNN_X3_AD=0; export DGNN_X3_BIG=0; fi
timeout 150 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r4b_${mode}_p1 -- python3 tools/dbg_gemm2.py > gpurun_out/r4b_${mode}_p1.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r4b_${mode}_p2 -- python3 tools/dbg_gemm2.py > gpurun_out/r4b_${mode}_p2.log 2>&1
python3 - <<PY
import pandas as pd, glob
for p in ("p1","p2"):
    fs=glob.glob('gpurun_out/r4b_${mode}_%s/*/*counter_collection.csv'%p)
    if not fs: print("$mode",p,"no output"); continue
    d=pd.read_csv(fs[0])
    d=d[d.Kernel_Name.str.contains('k_linear_fwd_x3')]
    print("$mode", p, d.groupby('Counter_Name').Counter_Value.mean().to_dict())
PY
done
