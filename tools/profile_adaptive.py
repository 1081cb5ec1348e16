"""Developer tool: cProfile of the host side of an adaptive run (bench workload c4_adaptive: K changes every iteration)."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gmmvi_amd.device import get_context
ctx = get_context()
wl = sys.argv[1] if len(sys.argv) > 1 else "c4_adaptive"
w = bench.build(wl, 1, 0)
algo = bench.make_gmmvi(w, 1, 0)
for _ in range(20):
    algo.train_iter()
ctx.sync()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(n):
    algo.train_iter()
ctx.sync()
pr.disable()
print(f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms per iteration, K = {algo.model.num_components}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
