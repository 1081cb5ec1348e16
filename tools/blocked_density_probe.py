"""Drive the blocked density path (D = 300 shard of config C5) a few times: target of rocprofv3 --pmc runs."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmmvi_amd.device import get_context
from gmmvi_amd import hip_ops

ctx = get_context()
rng = np.random.default_rng(0)
k, d, n = 64, 300, 19968
means = ctx.asarray(rng.normal(size=(k, d)) * 3)
chols = np.tril(rng.normal(size=(k, d, d)) * 0.05) + np.eye(d)[None] * 1.5
chols = ctx.asarray(chols)
packed, _ = hip_ops.pack_components(ctx, means, chols)
x = ctx.asarray(rng.normal(size=(n, d)))
logw = ctx.asarray(np.full(k, -np.log(k)))
for _ in range(3):
    hip_ops.mixture_eval(ctx, packed, logw, x, d, want_ld=True, want_grad=True)
ctx.sync()
