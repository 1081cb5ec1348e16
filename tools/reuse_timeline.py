"""Developer tool: host timeline of the single-call iteration with sample reuse (bench workload ns_reuse2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gmmvi_amd.device import get_context
ctx = get_context()
w = bench.build(sys.argv[1] if len(sys.argv) > 1 else "ns_reuse2", 1, 0)
algo = bench.make_gmmvi(w, 1, 0)
fp = algo._fast_path
T = {"counts (wait or issue + read-back)": 0.0, "python before the first call": 0.0, "C call 1": 0.0, "bookkeeping + issue of the next counts": 0.0,
     "C call 2": 0.0}
mark = {}
orig_counts = fp._new_sample_counts
def counts():
    t0 = time.perf_counter()
    out = orig_counts()
    mark["t"] = time.perf_counter()
    T["counts (wait or issue + read-back)"] += mark["t"] - t0
    mark["calls"] = 0
    return out
fp._new_sample_counts = counts
orig_fn = fp._fn
def fn(h, p):
    t0 = time.perf_counter()
    key = "python before the first call" if mark["calls"] == 0 else "bookkeeping + issue of the next counts"
    T[key] += t0 - mark["t"]
    rc = orig_fn(h, p)
    mark["t"] = time.perf_counter()
    T["C call 1" if mark["calls"] == 0 else "C call 2"] += mark["t"] - t0
    mark["calls"] += 1
    return rc
fp._fn = fn
for _ in range(40):
    algo.train_iter()
ctx.sync()
for k in T: T[k] = 0.0
n = 200
t0 = time.perf_counter()
for _ in range(n):
    algo.train_iter()
ctx.sync()
tot = time.perf_counter() - t0
print(f"{tot / n * 1e6:.0f} us per iteration: " + ", ".join(f"{k} {v / n * 1e6:.0f}" for k, v in T.items()) + f", rest {(tot - sum(T.values())) / n * 1e6:.0f}")
