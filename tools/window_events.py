"""Developer tool: GPU time per window of an UNsynchronised long run (one HIP event per window, read at the end): does a workload
slow down under sustained load (clock / power management) or with the age of the run (database growth)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gmmvi_amd.device import get_context
ctx = get_context()
wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 24
win = int(sys.argv[3]) if len(sys.argv) > 3 else 25
w = bench.build(wl, 1, 0)
algo = bench.make_gmmvi(w, 1, 0)
for _ in range(3):
    algo.train_iter()
ctx.sync()
evs = [ctx.event() for _ in range(n_win + 1)]
host = []
ctx.record(evs[0])
for j in range(n_win):
    t0 = time.perf_counter()
    for _ in range(win):
        algo.train_iter()
    host.append((time.perf_counter() - t0) / win * 1e3)
    ctx.record(evs[j + 1])
ctx.sync()
for j in range(n_win):
    print(f"iterations {3 + j * win + 1:4d}..{3 + (j + 1) * win:4d}: gpu {ctx.elapsed_ms(evs[j], evs[j + 1]) / win:8.3f} ms per iteration, "
          f"host issue {host[j]:7.3f}", flush=True)
