"""Timing probe for a wave-specialised fused layer (DESIGN 10 "next" 1): how fast is the PRODUCER side alone -- gathers + filter product + mean + split,
k_agg_sr<8, false> on 128-channel fp32 rows of the 1M-tet metric scene -- with its HBM stores switched off (DGNN_SR_NT=3: what a producer that parks its
rows in LDS would do), at 16 and at 8 wavefronts per CU.  One child process per setting (the knobs are read once per process).
    python tools/probe_producer.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, numpy as np, torch
sys.path.insert(0, %r)
import bench
from dgnn_amd import ops
from dgnn_amd.graph import GraphPlan
from dgnn_amd.processing.reorder import reorder_edges, scene_order
dev = "cuda:0"
adj, cent, x, ea = bench.make_scene(150000, 0)
n = adj.shape[0] // 4
ei = torch.from_numpy(adj.T.astype(np.int64)).to(dev)
co = scene_order(ei, n, centroids=torch.from_numpy(cent).to(dev), kind="morton")
ei_o, rows_o = reorder_edges(ei, co.order, co.rank)
ea = ops.gather_rows(ea.to(dev), rows_o)
h = torch.relu(torch.randn(n, 128, device=dev))
plan = GraphPlan(ei_o, n, n, hint=ops.PLAN_HINT_REFERENCE)
We, be = torch.randn(128, 20, device=dev) / 4, torch.randn(128, device=dev)
prep = ops.sr_prepare_filter(We, be)
fn = lambda: ops.aggregate_sr(plan.rowptr, plan.src, plan.eid, n, h, ea, We, be, prep, own_rows=True)
for _ in range(5): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"ms": e0.elapsed_time(e1) / 20, "n": n}))
''' % ROOT

for nt, wgs, ilv in ((1, 2, 2), (3, 2, 2), (3, 1, 2), (3, 2, 1), (3, 1, 1), (3, 2, 4)):
    env = dict(os.environ, DGNN_SR_NT=str(nt), DGNN_AGG_SR_WGS=str(wgs), DGNN_AGG_SR_ILV=str(ilv))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print("stores %-3s  %2d wavefronts / CU  chains %d :" % ("off" if nt & 2 else "on", 8 * wgs, ilv), line[-1] if line else r.stderr[-400:])
