"""Developer tool: which C entry points the effective-sample-size step of the reuse path calls, per iteration."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gmmvi_amd.device import get_context
ctx = get_context()
wl = sys.argv[1] if len(sys.argv) > 1 else "ns_reuse2"
w = bench.build(wl, 1, 0)
algo = bench.make_gmmvi(w, 1, 0)
fp = algo._fast_path
real = ctx.lib
log = []
class Proxy:
    def __getattr__(self, name):
        f = getattr(real, name)
        if not name.startswith("gmmvi_"):
            return f
        def g(*a):
            log.append(name)
            return f(*a)
        return g
orig = fp._new_sample_counts
def counts():
    ctx.lib = Proxy()
    try:
        return orig()
    finally:
        ctx.lib = real
fp._new_sample_counts = counts
seen = collections.Counter()
for it in range(60):
    log.clear()
    algo.train_iter()
    seen[tuple(log)] += 1
    if it in (0, 1, 2, 3, 10, 59):
        print(it, len(log), log)
print(len(seen), "distinct sequences")
