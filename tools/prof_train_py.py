"""cProfile of the Python side of the training step (the step is launch-bound: ~250 launches issued from Python)."""
import cProfile, pstats, sys, os, io
sys.argv = [sys.argv[0]] + ["--steps", "30", "--warmup", "5"] + sys.argv[1:]
pr = cProfile.Profile()
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train.py")).read()
pr.enable()
exec(compile(src, "bench_train.py", "exec"), {"__name__": "__main__", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train.py")})
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats("dgnn_amd|ctypes|torch.empty|torch.zeros|method", 45)
print(s.getvalue()[:12000])
