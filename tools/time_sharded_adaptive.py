"""Developer tool: one rank of the adaptive sharded path on bench.py's c4_adaptive workload (ms per iteration, K)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gmmvi_amd.device import get_context
from gmmvi_amd.sharded_adaptive import ShardedAdaptiveGMMVI
ctx = get_context()
w = bench.build("c4_adaptive", 1, 0)
sh = ShardedAdaptiveGMMVI.build(w, 1, 0)
for _ in range(20):
    sh.train_iter()
ctx.sync()
t0 = time.perf_counter()
for _ in range(100):
    sh.train_iter()
ctx.sync()
print(f"{(time.perf_counter() - t0) * 10:.3f} ms per iteration over iterations 21..120, K = {sh.num_components}")
