import cProfile, pstats, sys, os, io
sys.argv = ["bench_eraft_train.py"]
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
pr = cProfile.Profile()
src = open("tools/bench_eraft_train.py").read().replace("n = 3", "n = 2")
pr.enable()
exec(compile(src, "bench_eraft_train.py", "exec"))
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:60]))
