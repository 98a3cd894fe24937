#!/bin/bash
# Round-6 measurement sets (run on the GPU box through gpurun; every step under its own timeout):
#   bash tools/gpu_round6.sh train  <tag>   training step: bench lines of six configurations + kernel traces of four -> profiles/r06h_training.md
#   bash tools/gpu_round6.sh floor  <tag>   the training step's floor: a tiny scene (every kernel at its fixed cost) + host profile (cProfile)
#   bash tools/gpu_round6.sh gemm   <tag>   the training step's GEMM shapes under the three kernel selections (tools/bench_gemm_train_shapes.py)
#   bash tools/gpu_round6.sh wide   <tag>   the wide aggregate: static walk against the ticket walk(s): time, FETCH_SIZE, L2 hit rate -> profiles/r06_wide.md
#   bash tools/gpu_round6.sh tests  <tag>   pytest -m gpu (whole suite)
#   bash tools/gpu_round6.sh diag   <tag>   vector-memory path counters (L1 latency, L1->L2 latency, TLB, address / data stalls, wave states) of the
#                                           wide aggregate and of the default line's launches, one counter family per pass (a TA_* pass was tried:
#                                           rocprofv3 never returned from it, twice -- every pass stays under its own timeout)
M=${1:-train}; T=${2:-r6}
mkdir -p gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CFGS=("" "--updated --dtype bf16" "--widths 128,256,512,1024 --batch 1024" "--updated --dtype bf16 --widths 128,256,512,1024 --batch 1024" "--widths 64,128,256,512 --batch 1024" "--updated --dtype bf16 --widths 64,128,256,512 --batch 1024")
case $M in
train)
  for A in "${CFGS[@]}"; do timeout 300 python tools/bench_train.py $A --steps 300 --warmup 300 --no-roofline >> gpurun_out/$T/train.log 2>&1; done
  i=0
  for A in "${CFGS[@]:0:4}"; do
    i=$((i+1))
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/trace_$i -- python3 tools/bench_train.py $A --steps 60 --warmup 100 --no-roofline > gpurun_out/$T/trace_$i.log 2>&1
    F=$(ls gpurun_out/$T/trace_$i/*/*kernel_trace.csv | head -1)
    timeout 120 python tools/trace_gaps.py $F 105 40 --seq > gpurun_out/$T/seq_$i.txt 2>&1
    cp $(ls gpurun_out/$T/trace_$i/*/*kernel_stats.csv | head -1) gpurun_out/$T/kernel_stats_$i.csv
    rm -rf gpurun_out/$T/trace_$i
  done
  grep -o '"model": "[^"]*", "dtype": "[^"]*"\|"ms_per_step": [0-9.]*\|"avg_block_tets": [0-9.]*' gpurun_out/$T/train.log;;
floor)
  for A in "--points 3000 --batch 8" "--points 3000 --batch 8 --updated --dtype bf16"; do timeout 300 python tools/bench_train.py $A --steps 300 --warmup 300 --no-roofline >> gpurun_out/$T/floor.log 2>&1; done
  timeout 300 python tools/host_profile_train.py > gpurun_out/$T/host_static.log 2>&1
  timeout 300 python tools/host_profile_train.py --updated > gpurun_out/$T/host_updated.log 2>&1
  grep "un-profiled" gpurun_out/$T/host_*.log; grep -o '"ms_per_step": [0-9.]*' gpurun_out/$T/floor.log;;
gemm)
  timeout 200 python tools/bench_gemm_train_shapes.py > gpurun_out/$T/gemm_default.txt 2>&1
  DGNN_X3_SMALL=0 DGNN_BF16_SMALL=0 DGNN_GEMM_MID=0 timeout 200 python tools/bench_gemm_train_shapes.py > gpurun_out/$T/gemm_tiled.txt 2>&1
  DGNN_SMALL_BY_TILES=0 DGNN_GEMM_MID=0 DGNN_SMALL_SPLITK=0 timeout 200 python tools/bench_gemm_train_shapes.py > gpurun_out/$T/gemm_small.txt 2>&1
  paste -d'|' gpurun_out/$T/gemm_default.txt gpurun_out/$T/gemm_tiled.txt gpurun_out/$T/gemm_small.txt | cut -c1-260;;
wide)
  for CFG in "TK=0 G=1" "TK=1 G=1" "TK=1 G=2" "TK=1 G=4" "TK=0 G=1 WGS=1"; do
    WGS=2; eval $CFG
    export DGNN_AGG_SR_TICKETS=$TK DGNN_AGG_SR_TK_G=$G DGNN_AGG_SR_WGS=$WGS
    timeout 200 python bench.py --widths 64,128,256,512 --no-train --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open('gpurun_out/bench_full.json')); print('$CFG', round(d['value']/1e6,2), d['ms_per_step'], {k:round(v,3) for k,v in f['config']['replay_breakdown_ms'].items()})"
    B="python3 bench.py --widths 64,128,256,512 --no-train --no-extras --no-cpu-baseline --no-breakdown --steps 3 --warmup 1"
    timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$T/pmc3 -- $B > gpurun_out/$T/pmc3.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/$T/pmc4 -- $B > gpurun_out/$T/pmc4.log 2>&1
    python3 - <<PY
import glob, pandas as pd
for i in (3, 4):
    cs = glob.glob("gpurun_out/$T/pmc%d/*/*counter_collection.csv" % i)
    if not cs: print("$CFG: no csv", i); continue
    d = pd.read_csv(cs[0])
    d["K"] = d["Kernel_Name"].str.replace("(anonymous namespace)::", "").str.replace("void ", "").str.split("(").str[0].str[:40]
    d["dur_us"] = (d["End_Timestamp"] - d["Start_Timestamp"]) / 1e3
    k = d[d["K"].str.contains("k_agg_sr")]
    t = k.pivot_table(index="K", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    t["dur_us"] = k.groupby("K")["dur_us"].mean()
    print("$CFG pmc%d" % i); print(t.round(0).to_string())
PY
    rm -rf gpurun_out/$T/pmc3 gpurun_out/$T/pmc4
  done;;
diag)
  for W in wide default; do
    if [ $W = wide ]; then B="python3 bench.py --widths 64,128,256,512 --no-train --no-extras --no-cpu-baseline --no-breakdown --steps 3 --warmup 1"; PAT="k_agg_sr|k_gemm_sr";
    else B="python3 bench.py --no-train --no-extras --no-cpu-baseline --no-breakdown --steps 3 --warmup 1"; PAT="k_sage_fused"; fi
    i=0
    for C in "TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
             "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      timeout 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/$T/d_${W}_$i -- $B > gpurun_out/$T/d_${W}_$i.log 2>&1; echo "$W pass $i rc=$?"
      python3 - <<PY
import glob, pandas as pd
cs = glob.glob("gpurun_out/$T/d_${W}_$i/*/*counter_collection.csv")
if not cs: print("$W pass $i: no csv")
else:
    d = pd.read_csv(cs[0])
    d["K"] = d["Kernel_Name"].str.replace("(anonymous namespace)::", "").str.replace("void ", "").str.split("(").str[0].str[:44]
    d["dur_us"] = (d["End_Timestamp"] - d["Start_Timestamp"]) / 1e3
    k = d[d["K"].str.contains("$PAT")]
    t = k.pivot_table(index="K", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    t["dur_us"] = k.groupby("K")["dur_us"].mean(); t["n"] = k.groupby("K")["dur_us"].count() / max(1, k["Counter_Name"].nunique())
    pd.set_option("display.width", 250); print(t.round(0).to_string())
PY
      rm -rf gpurun_out/$T/d_${W}_$i
    done
  done;;
tests)
  timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/$T/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$T/tests.log; tail -n 5 gpurun_out/$T/tests.log;;
esac
