"""cProfile of the Python side of the training step, timed loop only (the step is host-bound: the main thread's enqueue time
sets the wall time).  python tools/prof_train_py.py [bench_train args]"""
import cProfile, pstats, sys, os, io
sys.argv = [sys.argv[0]] + ["--steps", "200", "--warmup", "10", "--no-roofline"] + sys.argv[1:]
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "bench_train.py")).read()
# profile from the first timed step on: split the script at the timed loop
head, tail = src.split("sync()\nt0 = time.perf_counter()", 1)
g = {"__name__": "__main__", "__file__": os.path.join(here, "bench_train.py")}
exec(compile(head, "bench_train.py", "exec"), g)
pr = cProfile.Profile()
pr.enable()
exec(compile("sync()\nt0 = time.perf_counter()" + tail, "bench_train.py", "exec"), g)
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(40)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats("dgnn_amd|bench_train|optim", 30)
print(s.getvalue()[:7000])
