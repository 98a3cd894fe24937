#!/bin/bash
# round 6: the wide aggregate's ticket walk (DGNN_AGG_SR_TICKETS=1, default) against the static walk (=0): tests, time, fabric traffic
T=${1:-r6k}
mkdir -p gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_wide.py -x -q -m gpu > gpurun_out/$T/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/$T/tests.log; tail -n 3 gpurun_out/$T/tests.log
for TK in 1 0 1 0; do
  for W in 64,128,256,512 128,256,512,1024; do
    DGNN_AGG_SR_TICKETS=$TK python bench.py --widths $W --no-train --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open('gpurun_out/bench_full.json')); print('tickets=$TK', '$W', round(d['value']/1e6,2), d['ms_per_step'], {k:round(v,3) for k,v in f['config']['replay_breakdown_ms'].items()})"
  done
done
for TK in 1 0; do
  B="python3 bench.py --widths 64,128,256,512 --no-train --no-extras --no-cpu-baseline --no-breakdown --steps 3 --warmup 1"
  DGNN_AGG_SR_TICKETS=$TK rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$T/pmc3_tk$TK -- $B > gpurun_out/$T/pmc3_tk$TK.log 2>&1
  DGNN_AGG_SR_TICKETS=$TK rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/$T/pmc4_tk$TK -- $B > gpurun_out/$T/pmc4_tk$TK.log 2>&1
  python3 - <<PY
import glob, pandas as pd
for i in (3, 4):
    cs = glob.glob("gpurun_out/$T/pmc%d_tk$TK/*/*counter_collection.csv" % i)
    if not cs: print("no csv", i); continue
    d = pd.read_csv(cs[0])
    d["K"] = d["Kernel_Name"].str.replace("(anonymous namespace)::", "").str.replace("void ", "").str.split("(").str[0].str[:40]
    d["dur_us"] = (d["End_Timestamp"] - d["Start_Timestamp"]) / 1e3
    k = d[d["K"].str.contains("k_agg_sr|k_gemm_sr")]
    t = k.pivot_table(index="K", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    t["dur_us"] = k.groupby("K")["dur_us"].mean()
    print("tickets=$TK pmc%d" % i); print(t.round(0).to_string())
PY
done
rm -rf gpurun_out/$T/pmc*_tk*/
