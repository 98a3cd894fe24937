"""A/B harness for kernel variants: times the fused conv layers of the metric graph (N = 1 010 078) on the REAL activations of
the kf96 network (post-BN/ReLU values run ~20 % faster than N(0,1) noise: operand toggling / DVFS) for several builds of the
library, interleaved in rounds inside one GPU session, and checks every variant's output against the first one.

    python tools/variants.py [--dtype bf16] [--rounds 5] [--points 150000] lib_a.so lib_b.so ...

Each library runs in its own worker process (the ctypes handle is process-global); the workers take turns round by round so
that clock / thermal drift hits all variants alike (guide rule 24).  Build variants with tools/build_variant.sh."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(lib, dtype, points):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dgnn_amd._lib as L
    L.LIB_PATH = os.path.abspath(lib)
    import numpy as np
    import torch
    from dgnn_amd import ops
    from dgnn_amd.config import Config, reconbench_pretrained
    from dgnn_amd.graph import GraphPlan
    from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
    import bench
    dev = "cuda:0"
    adj, cent, x, ea = bench.make_scene(points, 0)
    n = adj.shape[0] // 4
    if os.environ.get("DGNN_VARIANTS_ORDER", "loader") == "loader":      # the scene as the package's loader leaves it (ingest-time Morton order), as bench.py
        from dgnn_amd.synthetic import loader_cell_order
        adj, cent, order = loader_cell_order(adj, cent)
        x = x[torch.from_numpy(order)]
        ea = ea.view(n, 4, -1)[torch.from_numpy(order)].reshape(4 * n, -1)
    net = SurfaceNet(reconbench_pretrained(device=dev))
    net.load_state_dict(bench.load_weights())
    net = net.to(dev).eval()
    if dtype == "bf16":
        net.set_storage_dtype(torch.bfloat16)
    data = Config(x=x.to(dev), edge_attr=ea.to(dev), edge_index=torch.from_numpy(adj.T.astype(np.int64)).to(dev))
    plan = GraphPlan(data.edge_index, n, n, hint=ops.PLAN_HINT_REFERENCE)
    h = net._input_rows(data.x)
    if dtype == "bf16":
        h = net._storage_input(h)            # (the first fused layer reads the fp32 rows in place)
    f32 = lambda t: t if t.dtype == torch.float32 else ops.cast_to_f32(t)     # bf16 / unsigned 16-bit rows -> values
    fns, outs = [], []
    for i in range(4):
        fn = lambda h=h, i=i: net._eval_layers(h, n, data.edge_attr, [plan] * 4, True, only=i)
        fns.append(fn)
        h = fn()
        outs.append(h)
    dec = lambda: net._eval_decoder(outs[3])
    fns.append(dec)
    logits = dec()
    full = lambda: net.inference_layer(data, plan=GraphPlan(data.edge_index, n, n, hint=ops.PLAN_HINT_REFERENCE))
    fns.append(full)
    torch.cuda.synchronize()
    chk = [float(f32(o).double().abs().sum()) for o in outs] + [float(logits.double().abs().sum())]
    sample = [f32(o)[::997].float().cpu().numpy().tolist() for o in (outs[3], logits)]
    print(json.dumps({"ready": True, "checksums": chk}), flush=True)
    for line in sys.stdin:
        cmd = line.strip()
        if cmd == "quit":
            break
        if cmd == "sample":
            print(json.dumps(sample), flush=True)
            continue
        reps = int(cmd)
        res = []
        for fn in fns:
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / reps)
        print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    if args.worker:
        return worker(args.libs[0], args.dtype, args.points)
    import numpy as np
    procs = []
    for spec in args.libs:
        lib, _, envs = spec.partition("@")          # lib.so@KEY=VAL,KEY2=VAL2 sets environment variables for that worker
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k_, _, v_ = kv.partition("=")
            env[k_] = v_
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", "--dtype", args.dtype, "--points", str(args.points), lib],
                             stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env)
        procs.append(p)

    def read_json(p):
        while True:
            line = p.stdout.readline()
            if not line:
                raise RuntimeError("worker died")
            line = line.strip()
            if line.startswith("{") or line.startswith("["):
                return json.loads(line)
    ready = [read_json(p) for p in procs]
    samples = []
    for p in procs:
        p.stdin.write("sample\n"); p.stdin.flush()
        samples.append(read_json(p))
    times = [[] for _ in procs]
    for r in range(args.rounds):
        for k, p in enumerate(procs):
            p.stdin.write("%d\n" % args.reps); p.stdin.flush()
            times[k].append(read_json(p))
    for p in procs:
        p.stdin.write("quit\n"); p.stdin.flush()
    names = ["L0", "L1", "L2", "L3", "dec", "step"]
    print("%-34s " % "variant (median ms over %d rounds)" % args.rounds + " ".join("%8s" % n_ for n_ in names) + "   max|d relu3| max|d logit| vs first")
    base = samples[0]
    for k, lib in enumerate(args.libs):
        t = np.median(np.asarray(times[k]), axis=0)
        tmin = np.min(np.asarray(times[k]), axis=0)
        d3 = float(np.abs(np.asarray(samples[k][0]) - np.asarray(base[0])).max())
        dl = float(np.abs(np.asarray(samples[k][1]) - np.asarray(base[1])).max())
        print("%-34s " % os.path.basename(lib)[-34:] + " ".join("%8.4f" % v for v in t) + "   %.2e %.2e" % (d3, dl))
        print("%-34s " % "   (min)" + " ".join("%8.4f" % v for v in tmin))


if __name__ == "__main__":
    main()
