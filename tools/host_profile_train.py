"""Host side of the training step: cProfile of N steps on a tiny scene (every kernel at its fixed cost, so the host is the bound) -- where the
Python / ctypes time of a step goes.   python tools/host_profile_train.py [--updated] [--steps 1500]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
upd = "--updated" in sys.argv
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1500
import runpy
# reuse bench_train's set-up by running it with a tiny workload first (warm-up), then profile the loop here
sys.argv = ["bench_train.py", "--points", "3000", "--batch", "8", "--steps", "50", "--warmup", "300", "--no-roofline"] + (["--updated", "--dtype", "bf16"] if upd else [])
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train.py"), run_name="__main__")
import torch
tr, it, all_, opt, clf, Config = ns["tr"], ns["it"], ns["all_"], ns["opt"], ns["clf"], ns["Config"]
# the loader of bench_train is sized for its own steps; build a longer one
from dgnn_amd.sampler import NeighborSampler
n, batch, ei, dev = ns["n"], ns["batch"], ns["ei"], ns["dev"]
g = torch.Generator().manual_seed(0)
per = (n // batch) * batch
need = batch * (steps + 200)
idx = torch.cat([torch.randperm(n, generator=g)[:per] for _ in range(need // per + 1)])[:need]
loader = NeighborSampler(ei, sizes=[-1] * 4, node_idx=idx.to(dev), num_nodes=n, batch_size=batch, prefetch=True, reuse_buffers=True)
tr.attach_block_rows(loader, all_, ns["net"])
it = iter(loader)
for _ in range(100):
    bs, n_id, adjs = next(it)
    tr.train(Config(all=all_, batch_n_id=n_id, batch_adjs=adjs), opt, clf)
torch.cuda.synchronize()
t_next = t_train = 0.0
t0 = time.perf_counter()
for _ in range(steps // 3):
    a = time.perf_counter()
    bs, n_id, adjs = next(it)
    b = time.perf_counter()
    tr.train(Config(all=all_, batch_n_id=n_id, batch_adjs=adjs), opt, clf)
    c = time.perf_counter()
    t_next += b - a
    t_train += c - b
torch.cuda.synchronize()
k = steps // 3
print("un-profiled: %.1f us/step host (next(it) %.1f us, train() %.1f us)" % ((time.perf_counter() - t0) / k * 1e6, t_next / k * 1e6, t_train / k * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps // 3):
    bs, n_id, adjs = next(it)
    tr.train(Config(all=all_, batch_n_id=n_id, batch_adjs=adjs), opt, clf)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
