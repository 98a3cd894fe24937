"""profiles/r05_wide.md from the outputs of tools/gpu_wide.sh <tag> (bench lines + kernel traces) and tools/gpu_wide_pmc.sh <tag>_pmc<W> <widths> (PMC passes):
    python tools/wide_summarize.py <tag>"""
import glob, json, os, sys
import pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
N = 1010078
out = ["# Round 5: wide layers on split rows (`csrc/wide.hip`) -- MI355X, rocprofv3\n",
       "Commands: `bash tools/gpu_wide.sh %s` (tests/test_gpu_wide.py, `python bench.py --widths W --no-train --no-extras --steps 10`, `rocprofv3 --kernel-trace --stats -- "
       "python3 bench.py --widths W ... --steps 5 --warmup 2`) and `bash tools/gpu_wide_pmc.sh %s_pmc_<W> W` (four `--pmc` passes, counters only).\n" % (T, T)]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]


def agg_bytes(c):       # one launch of k_agg_sr over C channels: own row + 4 neighbour rows of split rows (4 B per element) + attributes + output
    return {"reuse": (c * 4 + 336 + c * 4), "none": (5 * c * 4 + 336 + c * 4)}


for W in ("64_128_256_512", "128_256_512_1024"):
    widths = [int(v) for v in W.split("_")]
    out.append("## widths %s\n" % widths)
    bj = os.path.join(G, T, "bench_%s.json" % W)
    if os.path.exists(bj):
        out += ["```json", open(bj).read().strip().splitlines()[-1], "```\n"]
    ks = glob.glob(os.path.join(G, T, "trace_%s" % W, "*", "*kernel_stats.csv"))
    if ks:
        d = pd.read_csv(ks[0])
        d["Name"] = d["Name"].map(short)
        out += ["kernel trace (5 timed + 2 warm-up + set-up steps):\n", "```", d[["Name", "Calls", "AverageNs", "Percentage"]].head(10).to_string(index=False), "```\n"]
    means = {}
    for i in (1, 2, 3, 4):
        cs = glob.glob(os.path.join(G, "%s_pmc_%s" % (T, W), "pmc%d" % i, "*", "*counter_collection.csv"))
        if not cs:
            continue
        d = pd.read_csv(cs[0])
        d["K"] = d["Kernel_Name"].map(short)
        d["dur_us"] = (d["End_Timestamp"] - d["Start_Timestamp"]) / 1e3
        d["grid"] = d["Grid_Size"] if "Grid_Size" in d.columns else 0
        keep = d[d["K"].str.contains("k_agg_sr|k_gemm_sr|k_sage_fused")]
        t = keep.pivot_table(index="K", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
        t["dur_us"] = keep.groupby("K")["dur_us"].mean().round(0)
        t["launches"] = keep.groupby("K")["Dispatch_Id"].nunique() if "Dispatch_Id" in keep.columns else 0
        means[i] = t
        out += ["pmc%d (mean per dispatch):\n" % i, "```", t.round(0).to_string(), "```\n"]
    if 3 in means and 4 in means and 1 in means:
        out.append("reading (FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; means over a kernel's launches of DIFFERENT widths -- the per-width split is in the bench line's `replay_breakdown_ms`):\n")
        for k in means[3].index:
            f, du = means[3].loc[k, "FETCH_SIZE"], means[3].loc[k, "dur_us"]
            wr = means[4].loc[k, "WRITE_SIZE"] if k in means[4].index else float("nan")
            hit, miss = (means[4].loc[k, "TCC_HIT_sum"], means[4].loc[k, "TCC_MISS_sum"]) if k in means[4].index else (float("nan"), float("nan"))
            gui = means[3].loc[k, "GRBM_GUI_ACTIVE"]
            clk = gui / 8 / du / 1e3
            busy = means[1].loc[k, "SQ_VALU_MFMA_BUSY_CYCLES"] if k in means[1].index else float("nan")
            wany, wc = (means[1].loc[k, "SQ_WAIT_ANY"], means[1].loc[k, "SQ_WAVE_CYCLES"]) if k in means[1].index else (float("nan"), 1.0)
            out.append("* `%s`: %.0f us, fabric traffic (2 x %.0f + %.0f) KiB = %.2f GB -> %.2f TB/s; L2 hit rate %.0f %%; matrix pipe %.0f %% busy; %.0f %% of wave cycles parked; %.2f GHz" % (
                k, du, f, wr, (2 * f + wr) * 1024 / 1e9, (2 * f + wr) * 1024 / du / 1e6, 100 * hit / (hit + miss), 100 * busy / 1024 / (clk * 1e3 * du), 100 * wany / wc, clk))
        out.append("")
open(os.path.join(ROOT, "profiles", "r05_wide.md"), "w").write("\n".join(out))
print("wrote profiles/r05_wide.md")
