"""Turns the rocprofv3 outputs of tools/prof_round2.sh <tag> (under gpurun_out/) into profiles/<name>.md and
profiles/<name>_traffic.json (read by bench.py for roofline.traffic / roofline.pmc, valid only for the kernel sources whose
hash it records).

    python tools/prof_summarize.py <tag> <name> "<title>" [--dtype f32|bf16] [--kernel <substring of the dominant kernel>]
                                   [--shape 128,128] [--bench gpurun_out/<file>.json ...]
"""
import argparse
import glob
import json
import os
import subprocess
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # csrc_sha, layer_bytes

ap = argparse.ArgumentParser()
ap.add_argument("tag"), ap.add_argument("name"), ap.add_argument("title")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--kernel", default=None)
ap.add_argument("--shape", default="128,128")
ap.add_argument("--args", default="", help="bench.py arguments the profile was taken with (documentation)")
ap.add_argument("--bench", nargs="*", default=[])
a = ap.parse_args()
G = os.path.join(ROOT, "gpurun_out")
shape = tuple(int(v) for v in a.shape.split(","))
# (fp32: the plain layer kernel -- template argument DEC = false; the last layer's launch that also carries the decoder is the `true` instantiation)
DOM = a.kernel or ("k_sage_fused_bf16<%d, %d" % shape if a.dtype == "bf16" else "k_sage_fused_mfma<%d, %d, 8, 2, 2, 2, f" % shape)
elem = 2 if a.dtype == "bf16" else 4
N = 1010078


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


out = ["# %s -- MI355X, rocprofv3\n" % a.title,
       "Commands: `bash tools/prof_round2.sh %s %s` = `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline %s` "
       "and four `--pmc` passes of the same command (counters only, no tracing domains).\n" % (a.tag, a.args, a.args)]
for p in a.bench:
    if os.path.exists(p) and os.path.getsize(p):
        out += ["## %s\n" % os.path.basename(p), "```json", open(p).read().strip().splitlines()[-1], "```\n"]
ks = glob.glob(os.path.join(G, a.tag + "_trace", "*", "*kernel_stats.csv"))
trace_ns = None
if ks:
    d = pd.read_csv(ks[0])
    d["Name"] = d["Name"].map(short)
    d_stats = d
    out += ["## kernel stats (kernel-trace --stats)\n", "```", d[["Name", "Calls", "AverageNs", "Percentage"]].head(16).to_string(index=False), "```\n"]
    rows = d[d["Name"].str.contains(DOM, regex=False)]
    if len(rows):
        trace_ns = float(rows.iloc[0]["AverageNs"])
means = {}
for i in (1, 2, 3, 4):
    cs = glob.glob(os.path.join(G, "%s_pmc%d" % (a.tag, i), "*", "*counter_collection.csv"))
    if not cs:
        continue
    d = pd.read_csv(cs[0])
    d["K"] = d["Kernel_Name"].map(short)
    d["dur_us"] = (d["End_Timestamp"] - d["Start_Timestamp"]) / 1e3
    keep = d[d["K"].str.contains("k_sage|k_decoder|k_plan_regular|k_agg|k_linear")]
    t = keep.pivot_table(index="K", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    t["dur_us"] = keep.groupby("K")["dur_us"].mean().round(0)
    means[i] = t
    out += ["## pmc%d (mean per dispatch)\n" % i, "```", t.round(0).to_string(), "```\n"]


def get(i, col, dom=None):
    t = means.get(i)
    if t is None:
        return None
    rows = [r for r in t.index if (dom or DOM) in r]
    return float(t.loc[rows[0], col]) if rows and col in t.columns else None


def dec_stats():
    """the same counters for the instantiation of the 128 -> 128 layer that also carries the decoder (template argument DEC = true), fp32 only"""
    dom = "k_sage_fused_bf16<128, 128" if a.dtype == "bf16" else (
        "k_sage_fused_ws<128, 2, true, true, false>" if "k_sage_fused_ws" in DOM else "k_sage_fused_mfma<128, 128, 8, 2, 2, 2, t")
    f, wr, du = get(3, "FETCH_SIZE", dom), get(4, "WRITE_SIZE", dom), get(3, "dur_us", dom)
    if not (f and wr is not None and du):
        return None
    gui, busy, valu = get(3, "GRBM_GUI_ACTIVE", dom), get(1, "SQ_VALU_MFMA_BUSY_CYCLES", dom), get(1, "SQ_INSTS_VALU", dom)
    hit, miss = get(4, "TCC_HIT_sum", dom), get(4, "TCC_MISS_sum", dom)
    wc, wany = get(1, "SQ_WAVE_CYCLES", dom), get(1, "SQ_WAIT_ANY", dom)
    clk = gui / 8 / du / 1e3
    cyc = clk * 1e3 * du
    tr = None
    if ks:
        rr = d_stats[d_stats["Name"].str.contains(dom, regex=False)]
        tr = float(rr.iloc[0]["AverageNs"]) / 1e3 if len(rr) else None
    return {"kernel": dom if dom.endswith(">") else (dom + ", ..., DEC = true>" if a.dtype == "bf16" else dom + "rue>"), "traffic_bytes_per_launch": (2 * f + wr) * 1024, "fetch_kib": f, "write_kib": wr, "avg_launch_us_profiled": du,
            "avg_launch_us_kernel_trace": tr, "mfma_busy_frac": round(busy / 1024 / cyc, 4), "valu_busy_frac": round(valu * 2 / 1024 / cyc, 4),
            "tcc_hit_rate": round(hit / (hit + miss), 4), "clock_ghz": round(clk, 3), "wait_any_frac": round(wany / wc, 4)}


fetch, write, dur = get(3, "FETCH_SIZE"), get(4, "WRITE_SIZE"), get(3, "dur_us")
if fetch and write:
    traffic = (2 * fetch + write) * 1024
    per_tet = bench.layer_bytes(shape[0], shape[1], elem)
    algo = per_tet * N
    gui, busy, valu = get(3, "GRBM_GUI_ACTIVE"), get(1, "SQ_VALU_MFMA_BUSY_CYCLES"), get(1, "SQ_INSTS_VALU")
    hit, miss = get(4, "TCC_HIT_sum"), get(4, "TCC_MISS_sum")
    wc, wany, winst, act = get(1, "SQ_WAVE_CYCLES"), get(1, "SQ_WAIT_ANY"), get(1, "SQ_WAIT_INST_ANY"), get(1, "SQ_ACTIVE_INST_ANY")
    clk = gui / 8 / dur / 1e3
    cyc = clk * 1e3 * dur
    out += ["## reading (%s, %.0f us per launch under the profiler%s, N = 1 010 078)\n" % (
                DOM, dur, (", %.1f us in the kernel trace" % (trace_ns / 1e3)) if trace_ns else ""),
            "* effective clock: GRBM_GUI_ACTIVE %.3g / 8 XCDs / %.0f us = %.2f GHz" % (gui, dur, clk),
            "* matrix pipe: SQ_VALU_MFMA_BUSY_CYCLES %.4g / 1024 SIMDs = %.3g busy cycles per SIMD of %.3g elapsed -> %.0f %% busy" % (
                busy, busy / 1024, cyc, 100 * busy / 1024 / cyc),
            "* VALU: SQ_INSTS_VALU %.4g wave instructions x 2 cycles (SIMD-32, MI355X_MICROARCH.md; rounds 1-2 priced them at 4) / 1024 SIMDs -> %.0f %% of SIMD cycles" % (
                valu, 100 * valu * 2 / 1024 / cyc),
            "* waves: %.0f %% of wave cycles parked at s_waitcnt / s_barrier (SQ_WAIT_ANY), %.0f %% stalled at issue (SQ_WAIT_INST_ANY), %.0f %% issuing" % (
                100 * wany / wc, 100 * winst / wc, 100 * act / wc),
            "* L2: TCC hit rate %.0f %%" % (100 * hit / (hit + miss)),
            "* fabric traffic per launch: (2 x FETCH_SIZE %.0f KiB [gfx950 correction, MI355X_MICROARCH.md HBM section] + WRITE_SIZE %.0f KiB) x 1024"
            % (fetch, write),
            "  = %.3f GB against %.3f GB algorithmic (%d B/tet) -> %.2fx; at %.0f us that is %.2f TB/s through the L2 <-> fabric interface "
            "(Infinity-Cache hits are counted there, MI355X_MICROARCH.md)\n" % (traffic / 1e9, algo / 1e9, per_tet, traffic / algo, dur, traffic / dur / 1e6)]
    ds = dec_stats() if (a.dtype == "f32" or "k_sage_fused_ws" in DOM) else None
    if ds and a.dtype == "bf16":
        cb = (bench.layer_bytes(128, 128, 2) + 2 * 128 + 8) * N
        out += ["## reading (%s: the last layer's launch with the decoder inside, %.0f us per launch under the profiler)\n" % (ds["kernel"], ds["avg_launch_us_profiled"]),
                "* fabric traffic per launch (2 x FETCH_SIZE %.0f KiB + WRITE_SIZE %.0f KiB) x 1024 = %.3f GB against the contract's %.3f GB (layer row %d + decoder row %d B/tet) -> %.2fx"
                % (ds["fetch_kib"], ds["write_kib"], ds["traffic_bytes_per_launch"] / 1e9, cb / 1e9, bench.layer_bytes(128, 128, 2), 2 * 128 + 8, ds["traffic_bytes_per_launch"] / cb),
                "* matrix pipe %.0f %% busy, VALU %.0f %%, %.0f %% of wave cycles parked, L2 hit rate %.0f %%, clock %.2f GHz\n" % (
                    100 * ds["mfma_busy_frac"], 100 * ds["valu_busy_frac"], 100 * ds["wait_any_frac"], 100 * ds["tcc_hit_rate"], ds["clock_ghz"])]
    elif ds:
        cb = (bench.layer_bytes(128, 128) + 520) * N
        out += ["## reading (%s: the last layer's launch with the decoder inside, %.0f us per launch under the profiler)\n" % (ds["kernel"], ds["avg_launch_us_profiled"]),
                "* fabric traffic per launch (2 x FETCH_SIZE %.0f KiB + WRITE_SIZE %.0f KiB) x 1024 = %.3f GB: the layer's 0.52 GB of output and the decoder's 0.52 GB read are gone "
                "(plain layer: %.3f GB); against the contract's algorithmic figure for what this launch executes (SURVEY 8d: layer row 1360 + decoder row 520 = 1880 B/tet = "
                "%.3f GB) that is %.2fx" % (ds["fetch_kib"], ds["write_kib"], ds["traffic_bytes_per_launch"] / 1e9, traffic / 1e9, cb / 1e9, ds["traffic_bytes_per_launch"] / cb),
                "* matrix pipe %.0f %% busy, VALU %.0f %%, %.0f %% of wave cycles parked, L2 hit rate %.0f %%, clock %.2f GHz\n" % (
                    100 * ds["mfma_busy_frac"], 100 * ds["valu_busy_frac"], 100 * ds["wait_any_frac"], 100 * ds["tcc_hit_rate"], ds["clock_ghz"])]
    try:
        commit = open(os.path.join(G, a.tag + "_commit.txt")).read().strip()
    except OSError:
        commit = None
    if not commit:
        commit = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None
    json.dump({"kernel": DOM, "dtype": a.dtype, "shape": list(shape), "n_tets": N, "traffic_bytes_per_launch": traffic, "fetch_kib": fetch,
               "write_kib": write, "mfma_busy_frac": round(busy / 1024 / cyc, 4), "valu_busy_frac": round(valu * 2 / 1024 / cyc, 4),
               "tcc_hit_rate": round(hit / (hit + miss), 4), "clock_ghz": round(clk, 3), "avg_launch_us_profiled": dur,
               "avg_launch_us_kernel_trace": trace_ns / 1e3 if trace_ns else None, "wait_any_frac": round(wany / wc, 4),
               "with_decoder": dec_stats() if (a.dtype == "f32" or "k_sage_fused_ws" in DOM) else None, "commit": commit, "csrc_sha": bench.csrc_sha(),
               "source": "profiles/%s.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH doubled per the guide)" % a.name},
              open(os.path.join(ROOT, "profiles", a.name + "_traffic.json"), "w"))
open(os.path.join(ROOT, "profiles", a.name + ".md"), "w").write("\n".join(out))
print("wrote profiles/%s.md" % a.name)
