import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from gmmvi_amd.device import get_context
from gmmvi_amd import hip_ops
ctx = get_context()
rng = np.random.default_rng(0)
for D in (20, 32, 50):
    K, S = 100, 100
    means = ctx.asarray(rng.normal(size=(K, D)))
    covs = np.stack([np.eye(D) * 2.0 for _ in range(K)])
    chols, _ = hip_ops.cholesky(ctx, ctx.asarray(covs))
    offs = ctx.asarray(np.arange(K + 1, dtype=np.int32) * S, np.int32)
    def run():
        hip_ops.sample_components(ctx, means, chols, offs, K * S, seed=1, first_index=0)
    for _ in range(5): run()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(200): run()
    ctx.record(e1)
    print(D, "sample_components alone: %.1f us" % (ctx.elapsed_ms(e0, e1) / 200 * 1e3))
