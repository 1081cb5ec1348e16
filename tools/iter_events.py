"""Developer tool: GPU time of every iteration of an UNsynchronised run (one HIP event per iteration on the compute stream,
read after the run) next to the host's issue time for it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gmmvi_amd.device import get_context
ctx = get_context()
wl = sys.argv[1] if len(sys.argv) > 1 else "ns"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = bench.build(wl, 1, 0)
algo = bench.make_gmmvi(w, 1, 0)
for _ in range(warm):
    algo.train_iter()
ctx.sync()
evs = [ctx.event() for _ in range(n + 1)]
host = []
ctx.record(evs[0])
for i in range(n):
    t0 = time.perf_counter()
    algo.train_iter()
    host.append((time.perf_counter() - t0) * 1e6)
    ctx.record(evs[i + 1])
ctx.sync()
gpu = [ctx.elapsed_ms(evs[i], evs[i + 1]) * 1e3 for i in range(n)]
for i in range(n):
    print(f"iteration {warm + i + 1:3d}: gpu {gpu[i]:7.1f} us   host issue {host[i]:7.1f} us")
print(f"mean gpu {sum(gpu) / n:.1f}, first 20: {sum(gpu[:20]) / 20:.1f}, last 20: {sum(gpu[-20:]) / 20:.1f}")
if len(sys.argv) > 4:                       # after a pause with the GPU idle: does the drift of the first iterations start again?
    for _ in range(300):
        algo.train_iter()
    ctx.sync()
    time.sleep(float(sys.argv[4]))
    evs = [ctx.event() for _ in range(41)]
    ctx.record(evs[0])
    for i in range(40):
        algo.train_iter()
        ctx.record(evs[i + 1])
    ctx.sync()
    gpu = [ctx.elapsed_ms(evs[i], evs[i + 1]) * 1e3 for i in range(40)]
    print(f"after 300 more iterations and {sys.argv[4]} s idle:", " ".join(f"{v:.1f}" for v in gpu))
