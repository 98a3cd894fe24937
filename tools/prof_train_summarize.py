"""profiles/<name>.md for the training step out of tools/gpu_final.sh's files under gpurun_out/:
    python tools/prof_train_summarize.py <tag> <name>"""
import os, sys
T, name = sys.argv[1], sys.argv[2]
G = "gpurun_out"
out = ["# Round 2: training step (block builder + forward + backward + Adam) -- MI355X, rocprofv3 kernel trace", "",
       "Commands (`tools/gpu_final.sh %s` on the GPU box): `rocprofv3 --kernel-trace --stats -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline" % T,
       "[--updated] [--dtype bf16]`, then `python tools/trace_gaps.py <kernel_trace.csv> 105 40` (steps delimited by the fused Adam kernel; queue 1 = the training",
       "step's stream, queue 2 = the block builder's side stream driven by the library's own host thread).  The profiler roughly doubles the host's launch cost, so",
       "`span` here is longer than the un-profiled `ms_per_step` of the bench lines below; `main-queue busy` is the GPU time of the step itself.", "",
       "## bench lines (un-profiled, same box, 300 warm-up + 200 timed steps)", "", "```json"]
for t in ["", "updated", "updateddtypebf16", "dtypebf16"]:
    out.append(open("%s/%s_bench_train%s.json" % (G, T, t)).read().strip())
out += ["```", "", "Interleaved A/B inside one process (`tools/ab_train.py whole=1 whole=0 composite=0`: all layers in one call / one call per layer / separate",
        "Functions; 7 rounds of 40 steps each):", "", "```"] + open("%s/%s_ab_train.txt" % (G, T)).read().strip().split("\n")[-4:] + ["```", ""]
for t, title in [("", "StaticEdgeFilters fp32 (all layers in one library call each way, fused loss)"),
                 ("updated", "UpdatedEdgeFilters sage+ fp32 (one call per conv each way, sparse edge chaining)"),
                 ("updateddtypebf16", "UpdatedEdgeFilters sage+ bf16 storage (BASELINE config 3's shape of work)")]:
    out += ["## " + title, "", "```"] + open("%s/%s_train%s.gaps.txt" % (G, T, t)).read().rstrip().split("\n") + ["```", ""]
out += ["## reading", "",
        "* ~110 kernels of 5-100 us on the step's stream, 1.3-1.6 ms of GPU time; the un-profiled step takes 1.4-2.3 ms depending on the box: it is",
        "  host-bound, and the A/B above shows what each layer of host-side work costs (all layers per call < one call per layer < separate Functions).",
        "* Round 2 removed the launches that were pure overhead (the ~40-launch loss, the per-layer `edge_attr[e_id]` copies, the whole-scene `[E_all, C]` edge",
        "  tensors of the Updated variant, the transposed-plan builds on the main stream, three host round trips in the metrics, four in the block builder,",
        "  the BatchNorm fold, one weight transpose and the zero fills per layer) and moved the block builder to the library's own thread and stream.",
        "* Largest GPU items of the Static step: the fp32-class GEMMs (15 launches, ~0.35 ms: M <= 70k rows, latency-bound), `k_agg_bwd` (4 launches, 0.18 ms,",
        "  0.56-1.0 TB/s on its compulsory bytes after the 4-edge batching / 16 waves per CU of this round, was 0.33-0.68), the BatchNorm column reductions",
        "  (`k_colreduce` 0.18 ms), `k_linear_wgrad_x3` + reduce (0.19 ms).",
        "* What would move it further: BatchNorm statistics out of the GEMM epilogue (removes `k_colreduce<0>` and one pass over z), one kernel for a layer's",
        "  three dz consumers (dWj, dWi, dbj) and one reduction behind it, the same model-level entry point for the Updated variant.", ""]
open(os.path.join("profiles", name + ".md"), "w").write("\n".join(out))
print("wrote profiles/%s.md" % name)
