"""Developer tool: the blocked-path launches of one C5-shard iteration (K = 64, D = 300, N = 19 968) timed one entry point at
a time with HIP events: density values (whitening, no Z store), density + gradient (whitening + Z store + gradient),
Stein estimate.  GMMVI_BLOCKED_F32=1 selects the f32 matrix-core route, GMMVI_BG_DEBUG the experiment switches of
bgemm_ws_kernel (csrc/blocked.hip)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmmvi_amd.device import get_context
from gmmvi_amd import hip_ops
ctx = get_context()
rng = np.random.default_rng(0)
K, D, N = int(os.environ.get("TB_K", 64)), int(os.environ.get("TB_D", 300)), int(os.environ.get("TB_N", 19968))


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1) / reps * 1e3


means = rng.normal(size=(K, D)) * 0.5
A = rng.normal(size=(K, D, D)) / np.sqrt(D)
chols = np.linalg.cholesky(A @ A.transpose(0, 2, 1) + np.eye(D))
pk, _ = hip_ops.pack_components(ctx, ctx.asarray(means), ctx.asarray(chols))
logw = ctx.asarray(np.full(K, -np.log(K)))
x = ctx.asarray(rng.normal(size=(N, D)))
tg = ctx.asarray(rng.normal(size=(N, D)))
t_val = timed(lambda: hip_ops.mixture_eval(ctx, pk, logw, x, D, want_ld=True, want_lp=True))
t_grad = timed(lambda: hip_ops.mixture_eval(ctx, pk, logw, x, D, want_ld=True, want_lp=True, want_grad=True))
ld, lp, grad = hip_ops.mixture_eval(ctx, pk, logw, x, D, want_ld=True, want_lp=True, want_grad=True)
t_stein = timed(lambda: hip_ops.stein(ctx, pk, x, ld, grad, lp, tg, D))
print(f"K={K} D={D} N={N} debug={os.environ.get('GMMVI_BG_DEBUG', '0')} f32={os.environ.get('GMMVI_BLOCKED_F32', '0')}: "
      f"values {t_val:.0f} us, values+gradient {t_grad:.0f} us, Stein {t_stein:.0f} us", flush=True)
