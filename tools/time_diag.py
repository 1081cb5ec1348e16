"""Developer tool: the north-star shape with a DIAGONAL mixture (module-by-module path: the single-call iteration takes
full-covariance models only) -- ms per iteration and per-kernel HIP-event times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bench
from helpers import samtron_config
from gmmvi_amd.device import get_context
from gmmvi_amd.models.diagonal_gmm import DiagonalGMM
ctx = get_context()
w = bench.build("ns", 1, 0)
cfg = samtron_config(w["s"], initial_stepsize=0.1, diag=True)
cfg["model_initialization"].update(prior_mean=0.0, initial_cov=300.0)
k, d = w["k_total"], w["d"]
model = DiagonalGMM(np.ones(k) / k, w["means"], np.full((k, d), 300.0, np.float32))
model.seed = w["seed"]
wrapper = w["GmmWrapper"](model, 0.1, 1e-12, 10000)
algo = w["GMMVI"].build_from_config(cfg, w["target"], wrapper)
for _ in range(20):
    algo.train_iter()
ctx.sync()
t0 = time.perf_counter()
for _ in range(200):
    algo.train_iter()
ctx.sync()
print(f"diagonal mixture, K = {k}, D = {d}, N = {k * w['s']}: {(time.perf_counter() - t0) * 5:.3f} ms per iteration "
      f"(single-call path eligible: {algo._fast_path.eligible()})")
ctx.check(ctx.lib.gmmvi_profile_enable(ctx.handle, 1))
for _ in range(50):
    algo.train_iter()
prof = bench.parse_profile(ctx)
for name, (c, ms, _) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:28s} {c / 50:5.1f} launches/iter  {1e3 * ms / c:8.1f} us each")
