"""Developer tool: why is a 20-step run of a fresh process slower than the steady state?  Two fresh algorithm objects in one
process: (a) unsynchronised windows of 5 iterations (wall time and host issue time), (b) per-kernel HIP-event averages over
iterations 6..25 against the same 20 iterations 200 later."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gmmvi_amd.device import get_context
ctx = get_context()
wl = sys.argv[1] if len(sys.argv) > 1 else "ns"
w = bench.build(wl, 1, 0)
algo = bench.make_gmmvi(w, 1, 0)
for j in range(12):
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        algo.train_iter()
    t1 = time.perf_counter()
    ctx.sync()
    print(f"iterations {5 * j + 1:3d}..{5 * j + 5:3d}: {(time.perf_counter() - t0) / 5 * 1e6:7.1f} us per iteration "
          f"(host issue {(t1 - t0) / 5 * 1e6:7.1f})", flush=True)
for _ in range(200):
    algo.train_iter()
ctx.sync()
t0 = time.perf_counter()
for _ in range(20):
    algo.train_iter()
ctx.sync()
print(f"iterations 261..280: {(time.perf_counter() - t0) / 20 * 1e6:7.1f} us per iteration")

algo = bench.make_gmmvi(w, 1, 0)
for _ in range(5):
    algo.train_iter()


def window(n):
    ctx.check(ctx.lib.gmmvi_profile_enable(ctx.handle, 1))
    for _ in range(n):
        algo.train_iter()
    p = bench.parse_profile(ctx)
    ctx.check(ctx.lib.gmmvi_profile_enable(ctx.handle, 0))
    return {k: 1e3 * ms / c for k, (c, ms, _) in p.items()}


first = window(20)
for _ in range(200):
    algo.train_iter()
late = window(20)
for k in first:
    print(f"{k:24s} iterations 6..25: {first[k]:7.2f} us   226..245: {late.get(k, float('nan')):7.2f} us")
print(f"{'sum':24s} {sum(first.values()):7.2f} {sum(late.values()):7.2f}")
