"""Developer tool: host timeline of the single-call iteration with sample reuse (bench workload ns_reuse2): issue of the
effective-sample-size step, wait for its read-back, Python part before the C call, the C call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gmmvi_amd.device import get_context, DeviceArray
from gmmvi_amd.optimization import fused
ctx = get_context()
w = bench.build("ns_reuse2", 1, 0)
algo = bench.make_gmmvi(w, 1, 0)
fp = algo._fast_path
T = {"A_issue": 0.0, "A_wait": 0.0, "B_python": 0.0, "B_call": 0.0, "rest": 0.0}
marks = {}
orig_numpy = DeviceArray.numpy
def numpy_timed(self):
    if marks.get("in_counts") and "t_issue_end" not in marks:
        marks["t_issue_end"] = time.perf_counter()
    return orig_numpy(self)
DeviceArray.numpy = numpy_timed
orig_counts = fp._new_sample_counts
def counts():
    marks.clear(); marks["in_counts"] = True
    t0 = time.perf_counter()
    out = orig_counts()
    t1 = time.perf_counter()
    marks["in_counts"] = False
    ti = marks.get("t_issue_end", t1)
    T["A_issue"] += ti - t0; T["A_wait"] += t1 - ti
    marks["t_counts_end"] = t1
    return out
fp._new_sample_counts = counts
orig_fn = fp._fn
def fn(h, p):
    t0 = time.perf_counter()
    T["B_python"] += t0 - marks["t_counts_end"]
    rc = orig_fn(h, p)
    T["B_call"] += time.perf_counter() - t0
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
T["rest"] = tot - sum(T.values())
print(f"{tot / n * 1e6:.0f} us per iteration: " + ", ".join(f"{k} {v / n * 1e6:.0f}" for k, v in T.items()))
