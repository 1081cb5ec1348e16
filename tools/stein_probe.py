"""Developer tool: the Stein estimate alone at a bench workload's shape (default: north star, K = 100, D = 20, N = 10 000) on
random inputs, launched repeatedly -- for rocprofv3 --kernel-trace / --pmc runs and A/B timing of kernel variants.
usage: python tools/stein_probe.py [workload] [launches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from gmmvi_amd import hip_ops
from gmmvi_amd.device import get_context

wl = sys.argv[1] if len(sys.argv) > 1 else "ns"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
w = bench.spec(wl, 1)
ctx = get_context()
k, d, n = w["k_total"], w["d"], w["n_total"]
rng = np.random.default_rng(0)
means = ctx.asarray(w["means"])
chols, _ = hip_ops.cholesky(ctx, ctx.asarray(w["covs"]))
packed, _ = hip_ops.pack_components(ctx, means, chols)
logw = ctx.asarray(np.full(k, -np.log(k), np.float32))
offs = ctx.asarray((np.arange(k + 1) * (n // k)).astype(np.int32), np.int32)
x, _ = hip_ops.sample_components(ctx, means, chols, offs, n, seed=1)
ld, lq, qg, bg = hip_ops.mixture_eval_dual(ctx, packed, logw, logw, x, d)
tg = ctx.asarray(rng.normal(size=(n, d)).astype(np.float32))
for _ in range(5):
    h, g = hip_ops.stein(ctx, packed, x, ld, qg, bg, tg, d)
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    h, g = hip_ops.stein(ctx, packed, x, ld, qg, bg, tg, d)
ctx.sync()
dt = (time.perf_counter() - t0) / reps
print(f"{wl}: K={k} D={d} N={n}: stein {1e6 * dt:.1f} us per call (wall, {reps} calls); |H| = {np.abs(h.numpy()).mean():.4g}")
