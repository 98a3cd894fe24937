"""Turns the rocprofv3 outputs of tools/prof_round1.sh <tag> (under gpurun_out/) into profiles/<name>.md and
profiles/<name>_traffic.json.  Usage: python tools/prof_summarize.py r1d r01d_final "<title>" """
import glob, json, os, sys
import pandas as pd

tag, name, title = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
DOM = "k_sage_fused_mfma<128, 128"


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


out = ["# %s -- MI355X, rocprofv3\n" % title,
       "Commands: `tools/prof_round1.sh %s` = `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`, "
       "four `--pmc` passes of the same command (counters only, no tracing domains), then `python3 bench.py --steps 20 --warmup 5` "
       "(default mode), `--gemm-mode f32`, `--cached-plan`, `tools/bench_train.py`.\n" % tag]
for key in ("bench", "bench_f32", "bench_cachedplan", "train"):
    p = os.path.join(G, "%s_%s.json" % (tag, key))
    if os.path.exists(p) and os.path.getsize(p):
        out += ["## %s\n" % key, "```json", open(p).read().strip().splitlines()[-1], "```\n"]
ks = glob.glob(os.path.join(G, tag + "_trace", "*", "*kernel_stats.csv"))
if ks:
    d = pd.read_csv(ks[0])
    d["Name"] = d["Name"].map(short)
    out += ["## kernel stats (kernel-trace --stats)\n", "```", d[["Name", "Calls", "AverageNs", "Percentage"]].head(20).to_string(index=False), "```\n"]
means = {}
for i in (1, 2, 3, 4):
    cs = glob.glob(os.path.join(G, "%s_pmc%d" % (tag, i), "*", "*counter_collection.csv"))
    if not cs:
        continue
    d = pd.read_csv(cs[0])
    d["K"] = d["Kernel_Name"].map(short)
    d["dur_us"] = (d["End_Timestamp"] - d["Start_Timestamp"]) / 1e3
    keep = d[d["K"].str.contains("k_sage|k_decoder|k_plan_regular|k_plan_sorted")]
    t = keep.pivot_table(index="K", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    t["dur_us"] = keep.groupby("K")["dur_us"].mean().round(0)
    means[i] = t
    out += ["## pmc%d (mean per dispatch)\n" % i, "```", t.round(0).to_string(), "```\n"]


def get(i, col):
    t = means.get(i)
    if t is None:
        return None
    rows = [r for r in t.index if DOM in r]
    return float(t.loc[rows[0], col]) if rows and col in t.columns else None


fetch, write, dur = get(3, "FETCH_SIZE"), get(4, "WRITE_SIZE"), get(3, "dur_us")
if fetch and write:
    n = 1010078
    traffic = (2 * fetch + write) * 1024
    algo = 1360 * n
    gui, busy, valu = get(3, "GRBM_GUI_ACTIVE"), get(1, "SQ_VALU_MFMA_BUSY_CYCLES"), get(1, "SQ_INSTS_VALU")
    hit, miss = get(4, "TCC_HIT_sum"), get(4, "TCC_MISS_sum")
    clk = gui / 8 / dur / 1e3
    cyc = clk * 1e3 * dur
    out += ["## reading (%s, %.0f us per launch under the profiler, N = 1 010 078)\n" % (DOM, dur),
            "* effective clock: GRBM_GUI_ACTIVE %.3g / 8 XCDs / %.0f us = %.2f GHz" % (gui, dur, clk),
            "* matrix pipe: SQ_VALU_MFMA_BUSY_CYCLES %.4g / 1024 SIMDs = %.3g busy cycles per SIMD of %.3g elapsed -> %.0f %% busy" % (
                busy, busy / 1024, cyc, 100 * busy / 1024 / cyc),
            "* VALU: SQ_INSTS_VALU %.4g wave instructions x 4 cycles / 1024 SIMDs -> %.0f %% of SIMD cycles" % (valu, 100 * valu * 4 / 1024 / cyc),
            "* L2: TCC hit rate %.0f %%" % (100 * hit / (hit + miss)),
            "* HBM traffic per launch: (2 x FETCH_SIZE %.0f KiB [gfx950 correction, MI355X_MICROARCH.md HBM section] + WRITE_SIZE %.0f KiB) x 1024"
            % (fetch, write),
            "  = %.3f GB against %.3f GB algorithmic (1360 B/tet) -> %.2fx\n" % (traffic / 1e9, algo / 1e9, traffic / algo)]
    json.dump({"kernel": DOM, "traffic_bytes_per_launch": traffic, "fetch_kib": fetch, "write_kib": write,
               "mfma_busy_frac": round(busy / 1024 / cyc, 4), "valu_busy_frac": round(valu * 4 / 1024 / cyc, 4),
               "tcc_hit_rate": round(hit / (hit + miss), 4), "clock_ghz": round(clk, 3),
               "source": "profiles/%s.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH doubled per the guide)" % name},
              open(os.path.join(ROOT, "profiles", name + "_traffic.json"), "w"))
open(os.path.join(ROOT, "profiles", name + ".md"), "w").write("\n".join(out))
print("wrote profiles/%s.md" % name)
