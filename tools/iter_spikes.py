"""Developer tool: wall time of every iteration of a bench workload (host clock, stream drained every iteration): where are the spikes?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gmmvi_amd.device import get_context
ctx = get_context()
wl = sys.argv[1] if len(sys.argv) > 1 else "ns"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
w = bench.build(wl, 1, 0)
algo = bench.make_gmmvi(w, 1, 0)
ts = []
for i in range(n):
    t0 = time.perf_counter()
    algo.train_iter()
    ctx.sync()
    ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts)
print(f"{wl}: median {np.median(ts):.1f} us (synchronised every iteration), mean {ts.mean():.1f}")
for i in np.argsort(ts)[-12:][::-1]:
    print(f"  iteration {i + 1}: {ts[i]:.0f} us")
for lo in range(0, n, 100):
    print(f"  iterations {lo + 1}..{lo + 100}: mean {ts[lo:lo + 100].mean():.1f} us")
