"""Developer tool: iteration time by window over a long run, then a cProfile of the host side and the per-kernel HIP-event
figures at that age (where does a workload's steady-state figure differ from its first iterations?)."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gmmvi_amd.device import get_context
ctx = get_context()
wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 16
win = int(sys.argv[3]) if len(sys.argv) > 3 else 25
w = bench.build(wl, 1, 0)
algo = bench.make_gmmvi(w, 1, 0)
out = []
for j in range(n_win):
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(win):
        algo.train_iter()
    t1 = time.perf_counter()
    ctx.sync()
    out.append(((time.perf_counter() - t0) / win * 1e3, (t1 - t0) / win * 1e3))
    print(f"window {j}: {out[-1][0]:.3f} ms per iteration (host issue {out[-1][1]:.3f})", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(win):
    algo.train_iter()
ctx.sync()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
ctx.check(ctx.lib.gmmvi_profile_enable(ctx.handle, 1))
for _ in range(win):
    algo.train_iter()
prof = bench.parse_profile(ctx)
for name, (c, ms, pr_) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:28s} {c / win:5.1f} launches/iter  {1e3 * ms / c:9.1f} us each  {ms / win:8.3f} ms/iter")
