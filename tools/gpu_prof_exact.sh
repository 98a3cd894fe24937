#!/bin/bash
# exact-fp32 mode (--gemm-mode f32): bench line + per-launch breakdown, kernel stats, one SQ counter pass.  bash tools/gpu_prof_exact.sh <tag>
T=${1:-r6x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$T
B="python3 bench.py --gemm-mode f32 --no-train --no-extras --no-cpu-baseline --steps 3 --warmup 1"
timeout 200 python bench.py --gemm-mode f32 --no-train --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open('gpurun_out/bench_full.json')); print(round(d['value']/1e6,2), d['ms_per_step'], d.get('ms_per_step_median'), {k:round(v,3) for k,v in f['config'].get('replay_breakdown_ms',{}).items()})"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/kt -- $B > gpurun_out/$T/kt.log 2>&1
python3 - <<PY
import glob, pandas as pd
f = glob.glob("gpurun_out/$T/kt/*/*kernel_stats.csv")[0]
d = pd.read_csv(f); d["Name"] = d["Name"].str.replace("(anonymous namespace)::", "").str.replace("void ", "").str[:60]
print(d[["Name", "Calls", "AverageNs", "Percentage"]].head(8).to_string())
PY
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$T/p1 -- $B > gpurun_out/$T/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d gpurun_out/$T/p2 -- $B > gpurun_out/$T/p2.log 2>&1
python3 - <<PY
import glob, pandas as pd
for p in ("p1", "p2"):
    cs = glob.glob("gpurun_out/$T/%s/*/*counter_collection.csv" % p)
    if not cs: print(p, "no csv"); continue
    d = pd.read_csv(cs[0])
    d["K"] = d["Kernel_Name"].str.replace("(anonymous namespace)::", "").str.replace("void ", "").str.split("(").str[0].str[:50]
    d["dur_us"] = (d["End_Timestamp"] - d["Start_Timestamp"]) / 1e3
    k = d[d["K"].str.contains("k_sage|k_dec|k_lin|k_agg")]
    t = k.pivot_table(index="K", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    t["dur_us"] = k.groupby("K")["dur_us"].mean()
    pd.set_option("display.width", 300); print(t.round(0).to_string())
PY
rm -rf gpurun_out/$T/kt gpurun_out/$T/p1 gpurun_out/$T/p2
