"""Developer tool: the target evaluation of the mixture targets in isolation (K_t components, log value + gradient on N samples),
HIP-event time per launch for the environment settings given on the command line (GMMVI_ME_KY / GMMVI_ME_NW ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gmmvi_amd.device import get_context
from gmmvi_amd import hip_ops
ctx = get_context()
rng = np.random.default_rng(0)
kt, n = int(os.environ.get("KT", 10)), int(os.environ.get("NS", 10000))
for d in (20, 32, 40, 50):
    means = rng.normal(size=(kt, d)).astype(np.float32) * 3
    a = rng.normal(size=(kt, d, d)).astype(np.float32) * 0.1
    covs = a @ np.transpose(a, (0, 2, 1)) + np.eye(d, dtype=np.float32)
    chols, _ = hip_ops.cholesky(ctx, ctx.asarray(covs))
    packed = hip_ops.pack_components(ctx, ctx.asarray(means), chols, want_inverse=False)[0]
    logw = ctx.asarray(np.full(kt, -np.log(kt), np.float32))
    x = ctx.asarray(rng.normal(size=(n, d)).astype(np.float32) * 3)
    for _ in range(5):
        hip_ops.mixture_eval(ctx, packed, logw, x, d, want_lp=True, want_grad=True)
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(50):
        hip_ops.mixture_eval(ctx, packed, logw, x, d, want_lp=True, want_grad=True)
    ctx.record(e1)
    ctx.sync()
    print(f"D = {d}: {ctx.elapsed_ms(e0, e1) * 20:.1f} us per launch (K_t = {kt}, N = {n})")
