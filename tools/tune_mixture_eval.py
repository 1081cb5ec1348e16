"""Developer tool: times gmmvi_mixture_eval at the north-star shape for one (feed, waves/block, K-chunks) geometry taken
from the environment (GMMVI_ME_NW / GMMVI_ME_KY).  Driven by tools/tune.sh on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmmvi_amd.device import get_context
from gmmvi_amd import hip_ops

K, D, N = int(os.environ.get("TK", 100)), int(os.environ.get("TD", 20)), int(os.environ.get("TN", 10000))
rng = np.random.default_rng(0)
ctx = get_context()
means = ctx.asarray(rng.normal(size=(K, D)) * 3)
covs = np.stack([(lambda a: a @ a.T / D + 0.3 * np.eye(D))(rng.normal(size=(D, D))) for _ in range(K)])
chols, _ = hip_ops.cholesky(ctx, ctx.asarray(covs))
packed, _ = hip_ops.pack_components(ctx, means, chols)
logw = ctx.asarray(np.full(K, -np.log(K)))
x = ctx.asarray(rng.normal(size=(N, D)) * 3)

def bench(**kw):
    for _ in range(5):
        hip_ops.mixture_eval(ctx, packed, logw, x, D, **kw)
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(50):
        hip_ops.mixture_eval(ctx, packed, logw, x, D, **kw)
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1) / 50 * 1e3

print(os.environ.get("GMMVI_ME_NW", "-"), os.environ.get("GMMVI_ME_KY", "-"),
      "nograd %.1f us" % bench(want_ld=True, want_lp=True), "grad %.1f us" % bench(want_ld=True, want_lp=True, want_grad=True))
