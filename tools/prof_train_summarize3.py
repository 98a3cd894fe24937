"""profiles/<name>.md for the training step out of tools/gpu_final_train.sh's files under gpurun_out/:
    python tools/prof_train_summarize3.py <tag> <name>"""
import os, sys
T, name = sys.argv[1], sys.argv[2]
G = "gpurun_out"
rd = lambda f: open(os.path.join(G, f)).read().strip()
seq = rd("%s_train_seq.txt" % T).split("\n")
cut = next(i for i, l in enumerate(seq) if l.startswith("sequence of step"))
out = ["# Round 3: training step (block builder + forward + backward + Adam), Static fp32 -- MI355X, rocprofv3 kernel trace", "",
       "Commands (`bash tools/gpu_final_train.sh %s` on the GPU box): `python tools/bench_train.py` (un-profiled lines), `python tools/ab_train.py fused=3 fused=1 fused=0`" % T,
       "(interleaved A/B of the fused launch chains inside one process: `dgnn_train_set_fused`, bit 0 = backward chain, bit 1 = batch statistics from the forward",
       "GEMM's epilogue), `rocprofv3 --kernel-trace --stats -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline`, then",
       "`python tools/trace_gaps.py <kernel_trace.csv> 105 40 --seq` (steps delimited by the fused Adam kernel; queue 1 = the training step's stream, queue 2 = the block",
       "builder's side stream driven by the library's own host thread).  The profiler roughly doubles the host's launch cost: `span` is longer than the un-profiled",
       "`ms_per_step`; `main-queue busy` is the GPU time of the step itself.", "",
       "## bench lines (un-profiled, same box, 300 warm-up + 200 timed steps)", "", "```json",
       rd("%s_bench_train.json" % T).split("\n")[-1], rd("%s_bench_train_updated_bf16.json" % T).split("\n")[-1],
       rd("%s_bench_train_updated.json" % T).split("\n")[-1], "```", "",
       "`host_issue_ms_per_step` = the main thread's time to issue the timed steps (before the final synchronize): equal to `ms_per_step`, the step is host-bound.", "",
       "## interleaved A/B of the fused launch chains (7 rounds of 40 steps per setting)", "", "```"] + rd("%s_ab_train.txt" % T).split("\n")[-4:] + ["```", "",
       "## GPU timeline of the Static step", "", "```"] + seq[:cut] + ["```", "", "## launch sequence of one step (main queue)", "", "```"] + seq[cut:] + ["```", "",
       "## Updated variant, bf16 storage (BASELINE configs 3 and 5): GPU timeline and launch sequence", "", "```"] + rd("%s_train_seq_updated_bf16.txt" % T).split("\n") + ["```", "",
       "## reading", "",
       "* Start of the round (`profiles/r02h_training.md`): 108 launches on the step's stream, 1.11 ms of GPU time, 1.17-1.24 ms per step.  Now 71 launches, 0.86-0.88 ms of",
       "  GPU time, 0.93-0.95 ms per step.  What went: per conv layer one weight-gradient launch pair instead of three (`k_linear_wgrad_x3_cat`), one input-gradient",
       "  GEMM instead of two with the sum folded into the aggregate backward's store (`k_agg_bwd_c<..., true>`), one reduction launch instead of two (`k_reduce_layer`),",
       "  no transposes (one `k_transpose_many` per pass), no column reduction for the forward statistics (GEMM epilogue), the decoder's output Linear inside the",
       "  whole-model calls.",
       "* The step is host-bound now (`host_issue_ms_per_step` = `ms_per_step`): about 0.28 ms of it are the 71 launches themselves, the rest Python -- torch's optimizer",
       "  and autograd engine, three indexing ops, this package's argument marshalling (trimmed this round: gradient views by one `as_strided` each, cached pointer",
       "  tables, plans that cut views of the block builder's buffers only on demand: 1.0 -> 0.93-0.95 ms).",
       "* Updated variant (bf16 storage): 129 launches / 1.29 ms of GPU time / 1.28-1.36 ms per step at the start of this work -> 90 launches / 0.85 ms / 0.92-1.05 ms:",
       "  all conv layers and the edge chaining per library call (one autograd node instead of three per layer: the step was bound by ~16 Python-level nodes each",
       "  way), merged bf16 weight gradients with 16-byte staging (`k_linear_wgrad_b_cat`), one input-gradient GEMM with the sum folded into the aggregate backward,",
       "  one transpose launch per layer, lane-group aggregate kernels (`k_agg_bwd_g` / `k_agg_fwd_g`: 4 channels per lane, 2-8 rows per wavefront instruction), the",
       "  output network inside the same autograd node, a one-wavefront-per-block bf16 GEMM for the small blocks (`k_linear_fwd_b_small`).  fp32 storage: 1.87 -> 1.03 ms.",
       "* Largest GPU items: `k_agg_bwd_c` 0.15 ms and `k_agg_fwd` 0.09 ms (one edge per wavefront instruction: at 28-64 channels the filter's 20 fma per edge are",
       "  the kernel's time -- a lane-group mapping, 4 channels per lane and 2-8 edges per instruction, is the next step), the small fp32-class GEMMs 0.19 ms, the",
       "  BatchNorm backward's reduce / finalise / apply chain 0.13 ms, `k_linear_wgrad_x3_cat` + reduction 0.13 ms.", ""]
open(os.path.join("profiles", name + ".md"), "w").write("\n".join(out))
print("wrote profiles/%s.md" % name)
