"""profiles/r06h_training.md out of the files of `bash tools/gpu_round6.sh train <tag>`, `... floor <tag2>`, `... gemm <tag3>` under gpurun_out/:
    python tools/prof_train_summarize6.py <train tag> [<floor tag> [<gemm tag>]]"""
import json, os, sys
import pandas as pd
T = sys.argv[1]
TF = sys.argv[2] if len(sys.argv) > 2 else None
TG = sys.argv[3] if len(sys.argv) > 3 else None
G = "gpurun_out"
rd = lambda *p: open(os.path.join(G, *p)).read().strip()
names = ["Static fp32, shipped widths [64,128,128,128], 2048 targets", "Updated bf16 storage, shipped widths, 2048 targets",
         "Static fp32, [128,256,512,1024], 1024 targets (configs/modelnet.yaml:44,56)", "Updated bf16 storage, [128,256,512,1024], 1024 targets (BASELINE config 3's workload)"]
out = ["# Round 6: training step (block builder + forward + backward + Adam) -- MI355X, rocprofv3 kernel trace", "",
       "Commands (`bash tools/gpu_round6.sh train %s` on the GPU box): `python tools/bench_train.py <config> --steps 300 --warmup 300 --no-roofline` (un-profiled lines)," % T,
       "`rocprofv3 --kernel-trace --stats -- python3 tools/bench_train.py <config> --steps 60 --warmup 100 --no-roofline`, `python tools/trace_gaps.py <kernel_trace.csv> 105 40 --seq`",
       "(steps delimited by the library's Adam kernel; queue of the Adam kernel = the training step's stream, the other queue = the block builder's).  Under the profiler the host",
       "issues more slowly: `span` is longer than the un-profiled `ms_per_step`; `main-queue busy` is the GPU time of the step itself.", "",
       "What changed this round and why: DESIGN.md section 8.  Round 5's record: `profiles/r05h_training.md` (69 launches on the Static step's stream, 0.83 ms of GPU time,",
       "0.84-0.92 ms per step; 1.16-1.18 ms at [128,256,512,1024]; Updated bf16 0.84-0.93 / 1.02-1.05 ms).", "",
       "## bench lines (un-profiled)", "", "```json"]
for l in rd(T, "train.log").split("\n"):
    if l.startswith("{"):
        d = json.loads(l)
        d.pop("roofline", None)
        out.append(json.dumps(d))
out += ["```", ""]
if TF:
    out += ["## the step's floor: a tiny scene (3000 points, batch 8: 566-cell blocks, every kernel at its fixed cost) and the host's share", "",
            "`bash tools/gpu_round6.sh floor %s`: bench lines of the tiny scene, then `tools/host_profile_train.py` (the loop timed un-profiled, then under cProfile)." % TF, "", "```"]
    for l in rd(TF, "floor.log").split("\n"):
        if l.startswith("{"):
            d = json.loads(l)
            out.append("%s %s: %.3f ms per step (host issue %.3f), %d-cell blocks" % (d["model"], d["dtype"], d["ms_per_step"], d["host_issue_ms_per_step"], d["avg_block_tets"]))
    for f in ("host_static.log", "host_updated.log"):
        txt = rd(TF, f).split("\n")
        out += [l for l in txt if l.startswith("un-profiled")]
        k = next((i for i, l in enumerate(txt) if "Ordered by: internal time" in l), None)
        if k is not None:
            out += [f + " (top of cProfile by internal time):"] + [l[:150] for l in txt[k + 3:k + 20]]
    out += ["```", ""]
for i, nm in enumerate(names, 1):
    ks = os.path.join(G, T, "kernel_stats_%d.csv" % i)
    if os.path.exists(ks):
        d = pd.read_csv(ks)
        d["Name"] = d["Name"].str.replace("(anonymous namespace)::", "", regex=False).str.replace("void ", "", regex=False).str.slice(0, 64)
        d["us/step"] = (d["TotalDurationNs"] / 160 / 1e3).round(1)
        out += ["## kernel stats: %s (kernel-trace --stats; 160 steps incl. warm-up; us per step = TotalDurationNs / 160, both queues)" % nm, "", "```",
                d[["Name", "Calls", "AverageNs", "us/step", "Percentage"]].head(22).to_string(index=False), "```", ""]
    sq = os.path.join(G, T, "seq_%d.txt" % i)
    if os.path.exists(sq):
        seq = open(sq).read().strip().split("\n")
        cut = next((j for j, l in enumerate(seq) if l.startswith("sequence of step")), len(seq))
        out += ["## GPU timeline and launch sequence: %s" % nm, "", "```"] + [l[:170] for l in seq[:cut][:42]] + [l[:170] for l in seq[cut:]] + ["```", ""]
if TG:
    out += ["## the step's GEMM shapes under three kernel selections (`bash tools/gpu_round6.sh gemm %s`; `tools/bench_gemm_train_shapes.py`)" % TG, "",
            "Left: the library's own choice (round 6).  Middle: the tiled kernels everywhere (`DGNN_X3_SMALL=0 DGNN_BF16_SMALL=0 DGNN_GEMM_MID=0`).  Right: round 5's choice", 
            "(`DGNN_SMALL_BY_TILES=0 DGNN_GEMM_MID=0 DGNN_SMALL_SPLITK=0`: the small-problem kernel for every M <= 16384).", "", "```"]
    a, b, c = (rd(TG, f).split("\n") for f in ("gemm_default.txt", "gemm_tiled.txt", "gemm_small.txt"))
    for x, y, z in zip(a, b, c):
        if x.startswith("M="):
            out.append("%s | %s | %s" % (x, y.split(":", 1)[1].strip(), z.split(":", 1)[1].strip()))
    out += ["```", ""]
open(os.path.join("profiles", "r06h_training.md"), "w").write("\n".join(out))
print("wrote profiles/r06h_training.md")
