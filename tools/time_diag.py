"""Developer tool: the dedicated diagonal sweeps (csrc/diag_sweep.hip) against the dense kernels on the embedded factors
diag(sigma) -- what DiagonalGMM ran on until round 3 -- at a few shapes."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmmvi_amd.device import get_context
from gmmvi_amd import hip_ops
ctx = get_context()
rng = np.random.default_rng(0)


def timed(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1) / reps * 1e3


for K, D, N, reps in ((100, 20, 10000, 100), (100, 50, 10000, 50), (64, 300, 20000, 5)):
    means = ctx.asarray(rng.normal(size=(K, D)) * 3)
    sigma = ctx.asarray(rng.uniform(0.5, 2.0, size=(K, D)))
    logw = ctx.asarray(np.full(K, -np.log(K)))
    x = ctx.asarray(rng.normal(size=(N, D)) * 3)
    tg = ctx.asarray(rng.normal(size=(N, D)))
    pd = hip_ops.diag_pack(ctx, means, sigma)
    dense = hip_ops.diag_embed(ctx, sigma)
    pk, _ = hip_ops.pack_components(ctx, means, dense)
    t_diag = timed(lambda: hip_ops.diag_mixture_eval(ctx, pd, logw, x, D, want_ld=True, want_lp=True, want_grad=True), reps)
    t_dense = timed(lambda: hip_ops.mixture_eval(ctx, pk, logw, x, D, want_ld=True, want_lp=True, want_grad=True), reps)
    ld, lp, grad = hip_ops.diag_mixture_eval(ctx, pd, logw, x, D, want_ld=True, want_lp=True, want_grad=True)
    ts_diag = timed(lambda: hip_ops.diag_stein(ctx, pd, x, ld, grad, lp, tg, D), reps)
    ts_dense = timed(lambda: hip_ops.stein(ctx, pk, x, ld, grad, lp, tg, D), reps)
    print(f"K={K} D={D} N={N}: density+gradient sweep {t_diag:.0f} us (dense on diag(sigma): {t_dense:.0f}); "
          f"Stein {ts_diag:.0f} us (dense: {ts_dense:.0f})", flush=True)
