"""Developer tool: mean iteration time over consecutive windows of a run (stream drained at window boundaries only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gmmvi_amd.device import get_context
ctx = get_context()
wl = sys.argv[1] if len(sys.argv) > 1 else "ns"
n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 24
win = int(sys.argv[3]) if len(sys.argv) > 3 else 25
w = bench.build(wl, 1, 0)
algo = bench.make_gmmvi(w, 1, 0)
out = []
for j in range(n_win):
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(win):
        algo.train_iter()
    ctx.sync()
    out.append((time.perf_counter() - t0) / win * 1e6)
print(wl, "us per iteration by window of", win, ":", " ".join(f"{v:.0f}" for v in out))
