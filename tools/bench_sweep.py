"""Developer tool: times the density sweeps of one iteration at a given shape (default: north star, K = 100, D = 20, N = 10 000)
for the geometry taken from the environment (GMMVI_LS*, GMMVI_ME_*): the dual sweep (model log q + gradient + background),
the post-update sweep (log values only) and a 10-component Student-t target evaluation with gradient; checks the results
against the CPU oracle's densities on a subset of the samples.  Driven by tools/sweep_configs.sh on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmmvi_amd.device import get_context
from gmmvi_amd import hip_ops, _lib

K, D, N = int(os.environ.get("TK", 100)), int(os.environ.get("TD", 20)), int(os.environ.get("TN", 10000))
REPS = int(os.environ.get("TREPS", 100))
rng = np.random.default_rng(0)
ctx = get_context()
means_h = rng.normal(size=(K, D)) * 30
covs = np.stack([(lambda a: a @ a.T / D + 3.0 * np.eye(D))(rng.normal(size=(D, D)) * 4) for _ in range(K)])
means = ctx.asarray(means_h)
chols, _ = hip_ops.cholesky(ctx, ctx.asarray(covs))
packed, _ = hip_ops.pack_components(ctx, means, chols)
logw = ctx.asarray(np.full(K, -np.log(K)))
logw2 = ctx.asarray(np.log(rng.dirichlet(np.ones(K))))
comp = rng.integers(0, K, N)
x_h = means_h[comp] + np.einsum("nij,nj->ni", np.linalg.cholesky(covs)[comp], rng.normal(size=(N, D)))
x = ctx.asarray(x_h)
KT = 10
tmeans = ctx.asarray(rng.normal(size=(KT, D)) * 10)
tchols, _ = hip_ops.cholesky(ctx, ctx.asarray(covs[:KT]))
tpacked, _ = hip_ops.pack_components(ctx, tmeans, tchols, family=_lib.STUDENT_T, nu=2.0)
tlogw = ctx.asarray(np.full(KT, -np.log(KT)))


def timed(fn):
    for _ in range(10):
        fn()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(REPS):
        fn()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1) / REPS * 1e3


def kernel_only(fn, reps=30):
    """per-launch HIP-event times of the kernels fn issues (gmmvi_profile_*), us"""
    import ctypes
    ctx.check(ctx.lib.gmmvi_profile_enable(ctx.handle, 1))
    for _ in range(reps):
        fn()
    buf = ctypes.create_string_buffer(1 << 16)
    ctx.check(ctx.lib.gmmvi_profile_report(ctx.handle, buf, len(buf)))
    ctx.check(ctx.lib.gmmvi_profile_enable(ctx.handle, 0))
    out = []
    for line in buf.value.decode().splitlines():
        name, cnt, ms, _ = line.split()
        out.append(f"{name} {1e3 * float(ms) / int(cnt):.1f}")
    return ", ".join(out)


t_dual = timed(lambda: hip_ops.mixture_eval_dual(ctx, packed, logw, logw2, x, D))
t_post = timed(lambda: hip_ops.mixture_eval(ctx, packed, logw, x, D, want_ld=True, want_lp=True))
t_tgt = timed(lambda: hip_ops.mixture_eval(ctx, tpacked, tlogw, x, D, family=_lib.STUDENT_T, nu=2.0, want_lp=True, want_grad=True))

# correctness against the fp64 oracle on the first 512 samples
from oracle import gmm as ogmm
o = ogmm.FullCovGMM(np.ones(K) / K, means_h, covs)
o.chol_cov = chols.numpy().astype(np.float64)
ld, lp, grad, lp2 = hip_ops.mixture_eval_dual(ctx, packed, logw, logw2, x, D)
sub = slice(0, 512)
lq_o, g_o, cld_o = o.log_density_and_grad(x_h[sub].astype(np.float32).astype(np.float64))
err_ld = np.abs(ld.numpy()[:, sub] - cld_o).max() / np.abs(cld_o).max()
err_lp = np.abs(lp.numpy()[sub] - lq_o).max()
err_g = np.abs(grad.numpy()[sub] - g_o).max() / np.abs(g_o).max()
cfg = " ".join(f"{k[6:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("GMMVI_"))
k_dual = kernel_only(lambda: hip_ops.mixture_eval_dual(ctx, packed, logw, logw2, x, D))
k_post = kernel_only(lambda: hip_ops.mixture_eval(ctx, packed, logw, x, D, want_ld=True, want_lp=True))
k_tgt = kernel_only(lambda: hip_ops.mixture_eval(ctx, tpacked, tlogw, x, D, family=_lib.STUDENT_T, nu=2.0, want_lp=True, want_grad=True))
print(f"    kernels: dual [{k_dual}]  post [{k_post}]  target [{k_tgt}]")
print(f"[{cfg}] K={K} D={D} N={N}: dual {t_dual:.1f} us  post {t_post:.1f} us  target {t_tgt:.1f} us | rel err ld {err_ld:.1e} lp {err_lp:.1e} grad {err_g:.1e}",
      flush=True)

if hasattr(ctx.lib, "gmmvi_debug_wg_times"):       # experiment build (-DGMMVI_ME_STAMPS): dispatch ramp of the last dual sweep
    import ctypes
    hip_ops.mixture_eval_dual(ctx, packed, logw, logw2, x, D)
    ctx.sync()
    nwg = int(os.environ.get("TNWG", 471))
    buf = (ctypes.c_longlong * (2 * nwg))()
    ctx.lib.gmmvi_debug_wg_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert ctx.lib.gmmvi_debug_wg_times(buf, 2 * nwg) == 0
    t = np.array(buf[:], dtype=np.int64).reshape(nwg, 2)
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01
    print(f"workgroup starts (us after the first): median {np.median(st):.2f} p90 {np.percentile(st, 90):.2f} max {st.max():.2f}; "
          f"ends: min {en.min():.2f} median {np.median(en):.2f} max {en.max():.2f}; life median {np.median(en - st):.2f} max {(en - st).max():.2f}")
    if hasattr(ctx.lib, "gmmvi_debug_wg_hw"):
        hb = (ctypes.c_ulonglong * nwg)()
        ctx.lib.gmmvi_debug_wg_hw.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert ctx.lib.gmmvi_debug_wg_hw(hb, nwg) == 0
        h = np.array(hb[:], dtype=np.uint64)
        hw, xcc = (h & np.uint64(0xffffffff)).astype(np.int64), (h >> np.uint64(32)).astype(np.int64) & 0xf
        cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 0x1, (hw >> 13) & 0x7          # gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
        place = xcc * 1000 + se * 100 + sh * 50 + cu
        import collections
        cnt = collections.Counter(place.tolist())
        per_cu = np.array([cnt[q] for q in place.tolist()])
        life = en - st
        for c in sorted(set(per_cu.tolist())):
            sel = per_cu == c
            print(f"  workgroups on a CU shared by {c}: {sel.sum()} (CUs {len([1 for v in cnt.values() if v == c])}), life median {np.median(life[sel]):.2f} max {life[sel].max():.2f} us, end median {np.median(en[sel]):.2f} max {en[sel].max():.2f}")
        print("  distinct CUs used:", len(cnt), " XCC histogram:", np.bincount(xcc, minlength=8).tolist())
        order = np.argsort(en)[-8:]
        print("  last to end: " + ", ".join(f"wg{int(i)}(x{i % 157},y{i // 157}) cu{int(place[i])} start {st[i]:.1f} end {en[i]:.1f}" for i in order))
