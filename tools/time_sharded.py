"""Developer tool: iteration rate of the sharded code path with ONE rank (LocalExchange) at the bench workload -- the host
cost of the multi-GPU orchestration without the collectives."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
from gmmvi_amd.sharded import ShardedGMMVI
from gmmvi_amd.device import get_context
w = bench.build("ns", 1, 0)
sh = ShardedGMMVI.build(w, 1, 0)
ctx = get_context()
for _ in range(20):
    sh.train_iter()
ctx.sync()
t0 = time.perf_counter()
n = 200
for _ in range(n):
    sh.train_iter()
ctx.sync()
dt = time.perf_counter() - t0
print(f"sharded path, 1 rank: {n / dt:.0f} train_iter/s ({1e6 * dt / n:.0f} us/iter)")
t0 = time.perf_counter()
for _ in range(n):
    sh.train_iter()
dt_host = time.perf_counter() - t0
ctx.sync()
print(f"host issue time: {1e6 * dt_host / n:.0f} us/iter")
