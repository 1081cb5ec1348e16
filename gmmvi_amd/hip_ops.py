"""Shape-checked Python wrappers over the C ABI (include/gmmvi_hip.h).  Every wrapper validates operand shapes on
the host before the launch, so a kernel never sees a grid/operand mismatch."""
import math

import numpy as np

from . import _lib
from .device import DeviceArray

F32, I32 = np.dtype(np.float32), np.dtype(np.int32)


def _req(a, shape, dtype=F32, name="array"):
    if not isinstance(a, DeviceArray):
        raise TypeError(f"{name}: expected DeviceArray, got {type(a)}")
    if a.dtype != dtype or tuple(a.shape) != tuple(shape):
        raise ValueError(f"{name}: expected {dtype}{tuple(shape)}, got {a.dtype}{a.shape}")
    return a.ptr


def _opt(a, shape, dtype=F32, name="array"):
    return None if a is None else _req(a, shape, dtype, name)


def packed_stride(d):
    s = int(_lib.load().gmmvi_packed_stride(int(d)))
    if s == 0:
        raise _lib.GmmviError(f"dimension {d} is not supported by the register-resident kernels (D <= {_lib.MAX_DIM})")
    return s


def pack_components(ctx, means, chols, family=_lib.GAUSS, nu=0.0, want_inverse=False):
    """-> (packed [K, stride], inv_chols [K,D,D] or None)."""
    k, d = means.shape
    _req(means, (k, d), name="means"); _req(chols, (k, d, d), name="chols")
    packed = ctx.empty((k, packed_stride(d)))
    inv = ctx.empty((k, d, d)) if want_inverse else None
    ctx.check(ctx.lib.gmmvi_pack_components(ctx.handle, family, float(nu), k, d, means.ptr, chols.ptr, packed.ptr,
                                            None if inv is None else inv.ptr))
    return packed, inv


def cholesky(ctx, covs):
    k, d, _ = covs.shape
    _req(covs, (k, d, d), name="covs")
    chols = ctx.empty((k, d, d))
    ok = ctx.empty((k,), np.int32)
    ctx.check(ctx.lib.gmmvi_cholesky(ctx.handle, k, d, covs.ptr, chols.ptr, ok.ptr))
    return chols, ok


def mixture_eval(ctx, packed, logw, x, d, family=_lib.GAUSS, nu=0.0, want_ld=False, want_lp=True, want_grad=False):
    """-> (ld [K,N] | None, lp [N] | None, grad [N,D] | None)."""
    k = packed.shape[0]
    n = x.shape[0]
    _req(packed, (k, packed_stride(d)), name="packed"); _req(logw, (k,), name="logw"); _req(x, (n, d), name="x")
    ld = ctx.empty((k, n)) if want_ld else None
    lp = ctx.empty((n,)) if want_lp else None
    grad = ctx.empty((n, d)) if want_grad else None
    if n > 0:
        ctx.check(ctx.lib.gmmvi_mixture_eval(ctx.handle, family, float(nu), k, d, packed.ptr, logw.ptr, x.ptr, n,
                                             None if ld is None else ld.ptr, None if lp is None else lp.ptr,
                                             None if grad is None else grad.ptr))
    return ld, lp, grad


def mixture_eval_dual(ctx, packed, logw, logw2, x, d, want_ld=True, want_grad=True):
    """One sweep, two mixtures over the same components -> (ld | None, lp, grad | None, lp2)."""
    k = packed.shape[0]
    n = x.shape[0]
    _req(packed, (k, packed_stride(d)), name="packed"); _req(logw, (k,), name="logw"); _req(logw2, (k,), name="logw2")
    _req(x, (n, d), name="x")
    ld = ctx.empty((k, n)) if want_ld else None
    lp, lp2 = ctx.empty((n,)), ctx.empty((n,))
    grad = ctx.empty((n, d)) if want_grad else None
    if n > 0:
        ctx.check(ctx.lib.gmmvi_mixture_eval_dual(ctx.handle, _lib.GAUSS, 0.0, k, d, packed.ptr, logw.ptr, logw2.ptr,
                                                  x.ptr, n, None if ld is None else ld.ptr, lp.ptr,
                                                  None if grad is None else grad.ptr, lp2.ptr))
    return ld, lp, grad, lp2


def target_planar(ctx, prior_std, goals, likelihood_std, x, want_grad=True):
    n, d = x.shape
    g = goals.shape[0]
    _req(prior_std, (d,), name="prior_std"); _req(goals, (g, 2), name="goals"); _req(x, (n, d), name="x")
    lp = ctx.empty((n,))
    grad = ctx.empty((n, d)) if want_grad else None
    if n > 0:
        ctx.check(ctx.lib.gmmvi_target_planar(ctx.handle, d, prior_std.ptr, g, goals.ptr, float(likelihood_std), x.ptr,
                                              n, lp.ptr, None if grad is None else grad.ptr))
    return lp, grad


def sample_components(ctx, means, chols, offsets, n, seed=0, first_index=0, stream_id=0, eps=None):
    """offsets: DeviceArray int32 [K+1] prefix sums with offsets[K] == n.  -> (x [n,D], mapping [n] int32)."""
    k, d = means.shape
    _req(means, (k, d), name="means"); _req(chols, (k, d, d), name="chols"); _req(offsets, (k + 1,), I32, "offsets")
    if eps is not None:
        _req(eps, (n, d), name="eps")
    x = ctx.empty((n, d))
    mapping = ctx.empty((n,), np.int32)
    if n > 0:
        ctx.check(ctx.lib.gmmvi_sample_components(ctx.handle, k, d, means.ptr, chols.ptr, offsets.ptr, n,
                                                  int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_index), int(stream_id),
                                                  None if eps is None else eps.ptr, x.ptr, mapping.ptr))
    return x, mapping


def philox_normals(ctx, seed, first_index, n, d, stream_id=0):
    out = ctx.empty((n, d))
    if n > 0:
        ctx.check(ctx.lib.gmmvi_philox_normals(ctx.handle, int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_index),
                                               int(stream_id), n, d, out.ptr))
    return out


def philox_uniforms(ctx, seed, first_index, n, stream_id=1):
    out = ctx.empty((n,))
    if n > 0:
        ctx.check(ctx.lib.gmmvi_philox_uniforms(ctx.handle, int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_index),
                                                int(stream_id), n, out.ptr))
    return out


def stein(ctx, packed, x, ld, qgrad, bg, tgrad, d, mapping=None, map_offset=0, self_normalized=True,
          own_samples_only=False):
    k = packed.shape[0]
    n = x.shape[0]
    _req(packed, (k, packed_stride(d)), name="packed"); _req(x, (n, d), name="x")
    _req(qgrad, (n, d), name="qgrad"); _req(tgrad, (n, d), name="tgrad")
    if own_samples_only:
        _req(mapping, (n,), I32, "mapping")
    else:
        _req(ld, (k, n), name="ld"); _req(bg, (n,), name="bg")
    flags = (_lib.SELF_NORMALIZED if self_normalized else 0) | (_lib.OWN_SAMPLES_ONLY if own_samples_only else 0)
    h_neg = ctx.empty((k, d, d))
    g_neg = ctx.empty((k, d))
    ctx.check(ctx.lib.gmmvi_stein(ctx.handle, k, d, packed.ptr, x.ptr, n, None if ld is None else ld.ptr, qgrad.ptr,
                                  None if bg is None else bg.ptr, tgrad.ptr,
                                  None if mapping is None else mapping.ptr, int(map_offset), flags, h_neg.ptr,
                                  g_neg.ptr))
    return h_neg, g_neg


def more(ctx, packed, chols, x, ld, logq, bg, tlp, l2, d, mapping=None, map_offset=0, self_normalized=True,
         own_samples_only=False):
    """MORE estimate (gmmvi_more) -> (h_neg [K,D,D], g_neg [K,D])."""
    k = packed.shape[0]
    n = x.shape[0]
    if d >= _lib.MAX_DIM:
        raise ValueError(f"MORE estimator: D = {d} is not supported by the HIP kernels (D <= {_lib.MAX_DIM - 1})")
    _req(packed, (k, packed_stride(d)), name="packed"); _req(chols, (k, d, d), name="chols"); _req(x, (n, d), name="x")
    _req(logq, (n,), name="logq"); _req(tlp, (n,), name="tlp"); _req(l2, (k,), name="l2")
    if own_samples_only:
        _req(mapping, (n,), I32, "mapping")
    else:
        _req(ld, (k, n), name="ld"); _req(bg, (n,), name="bg")
    flags = (_lib.SELF_NORMALIZED if self_normalized else 0) | (_lib.OWN_SAMPLES_ONLY if own_samples_only else 0)
    h_neg = ctx.empty((k, d, d))
    g_neg = ctx.empty((k, d))
    ctx.check(ctx.lib.gmmvi_more(ctx.handle, k, d, packed.ptr, chols.ptr, x.ptr, n, None if ld is None else ld.ptr,
                                 logq.ptr, None if bg is None else bg.ptr, tlp.ptr,
                                 None if mapping is None else mapping.ptr, int(map_offset), flags, l2.ptr, h_neg.ptr,
                                 g_neg.ptr))
    return h_neg, g_neg


def update_components_kl(ctx, means, chols, h_neg, g_neg, stepsizes, temperature, l2_init, last_eta, l2, num_updates,
                         want_info=False, reference=False, want_packed=False):
    """-> (success, kl | None, probes | None[, packed])."""
    k, d = means.shape
    _req(means, (k, d), name="means"); _req(chols, (k, d, d), name="chols")
    _req(h_neg, (k, d, d), name="h_neg"); _req(g_neg, (k, d), name="g_neg"); _req(stepsizes, (k,), name="stepsizes")
    _req(last_eta, (k,), name="last_eta"); _req(l2, (k,), name="l2"); _req(num_updates, (k,), name="num_updates")
    success = ctx.empty((k,), np.int32)
    kl = ctx.empty((k,)) if want_info else None
    probes = ctx.empty((k,), np.int32) if want_info else None
    args = [ctx.handle, k, d, means.ptr, chols.ptr, h_neg.ptr, g_neg.ptr, stepsizes.ptr, float(temperature),
            float(l2_init), last_eta.ptr, l2.ptr, num_updates.ptr, success.ptr, None if kl is None else kl.ptr,
            None if probes is None else probes.ptr]
    if reference:
        ctx.check(ctx.lib.gmmvi_update_components_kl_reference(*args))
        return success, kl, probes
    packed = ctx.empty((k, packed_stride(d))) if want_packed else None
    ctx.check(ctx.lib.gmmvi_update_components_kl(*args, None if packed is None else packed.ptr))
    return (success, kl, probes, packed) if want_packed else (success, kl, probes)


def update_components_plain(ctx, mode, means, chols, h_neg, g_neg, stepsizes, l2_init, l2, num_updates):
    k, d = means.shape
    _req(means, (k, d), name="means"); _req(chols, (k, d, d), name="chols")
    _req(h_neg, (k, d, d), name="h_neg"); _req(g_neg, (k, d), name="g_neg"); _req(stepsizes, (k,), name="stepsizes")
    _req(l2, (k,), name="l2"); _req(num_updates, (k,), name="num_updates")
    success = ctx.empty((k,), np.int32)
    fn = ctx.lib.gmmvi_update_components_direct if mode == "direct" else ctx.lib.gmmvi_update_components_iblr
    ctx.check(fn(ctx.handle, k, d, means.ptr, chols.ptr, h_neg.ptr, g_neg.ptr, stepsizes.ptr, float(l2_init), l2.ptr,
                 num_updates.ptr, success.ptr))
    return success


# ---- dedicated kernels for diagonal mixtures (csrc/diag_sweep.hip): O(D) per (sample, component) pair ----------------------
def diag_packed_stride(d):
    return int(_lib.load().gmmvi_diag_packed_stride(int(d)))


def diag_pack(ctx, means, sigma):
    """(means [K,D], standard deviations [K,D]) -> component blocks [K, diag_packed_stride(D)]."""
    k, d = means.shape
    _req(means, (k, d), name="means"); _req(sigma, (k, d), name="sigma")
    packed = ctx.empty((k, diag_packed_stride(d)))
    ctx.check(ctx.lib.gmmvi_diag_pack(ctx.handle, k, d, means.ptr, sigma.ptr, packed.ptr))
    return packed


def diag_mixture_eval(ctx, packed, logw, x, d, want_ld=False, want_lp=True, want_grad=False, logw2=None):
    """-> (ld [K,N] | None, lp [N] | None, grad [N,D] | None[, lp2 [N] when logw2 is given])."""
    k = packed.shape[0]
    n = x.shape[0]
    _req(packed, (k, diag_packed_stride(d)), name="packed"); _req(logw, (k,), name="logw"); _req(x, (n, d), name="x")
    if logw2 is not None:
        _req(logw2, (k,), name="logw2")
        want_lp = True
    ld = ctx.empty((k, n)) if want_ld else None
    lp = ctx.empty((n,)) if want_lp else None
    grad = ctx.empty((n, d)) if want_grad else None
    lp2 = ctx.empty((n,)) if logw2 is not None else None
    if n > 0:
        ctx.check(ctx.lib.gmmvi_diag_mixture_eval(ctx.handle, k, d, packed.ptr, logw.ptr, None if logw2 is None else logw2.ptr,
                                                  x.ptr, n, None if ld is None else ld.ptr, None if lp is None else lp.ptr,
                                                  None if grad is None else grad.ptr, None if lp2 is None else lp2.ptr))
    return (ld, lp, grad) if logw2 is None else (ld, lp, grad, lp2)


def diag_sample(ctx, means, sigma, offsets, n, seed=0, first_index=0, stream_id=0, eps=None):
    """x = mu_k + sigma_k * eps in component order -> (x [n,D], mapping [n] int32); arguments as sample_components."""
    k, d = means.shape
    _req(means, (k, d), name="means"); _req(sigma, (k, d), name="sigma"); _req(offsets, (k + 1,), I32, "offsets")
    if eps is not None:
        _req(eps, (n, d), name="eps")
    x = ctx.empty((n, d))
    mapping = ctx.empty((n,), np.int32)
    if n > 0:
        ctx.check(ctx.lib.gmmvi_diag_sample(ctx.handle, k, d, means.ptr, sigma.ptr, offsets.ptr, n,
                                            int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_index), int(stream_id),
                                            None if eps is None else eps.ptr, x.ptr, mapping.ptr))
    return x, mapping


def diag_stein(ctx, packed, x, ld, qgrad, bg, tgrad, d, mapping=None, map_offset=0, self_normalized=True,
               own_samples_only=False):
    """Stein estimate of a diagonal mixture -> (h_neg_diag [K,D], g_neg [K,D])."""
    k = packed.shape[0]
    n = x.shape[0]
    _req(packed, (k, diag_packed_stride(d)), name="packed"); _req(x, (n, d), name="x")
    _req(qgrad, (n, d), name="qgrad"); _req(tgrad, (n, d), name="tgrad")
    if own_samples_only:
        _req(mapping, (n,), I32, "mapping")
    else:
        _req(ld, (k, n), name="ld"); _req(bg, (n,), name="bg")
    flags = (_lib.SELF_NORMALIZED if self_normalized else 0) | (_lib.OWN_SAMPLES_ONLY if own_samples_only else 0)
    h_neg = ctx.empty((k, d))
    g_neg = ctx.empty((k, d))
    ctx.check(ctx.lib.gmmvi_diag_stein(ctx.handle, k, d, packed.ptr, x.ptr, n, None if ld is None else ld.ptr, qgrad.ptr,
                                       None if bg is None else bg.ptr, tgrad.ptr, None if mapping is None else mapping.ptr,
                                       int(map_offset), flags, h_neg.ptr, g_neg.ptr))
    return h_neg, g_neg


def diag_embed(ctx, chols_diag):
    """[K,D] sigma -> dense lower-triangular factors [K,D,D] = diag(sigma) for the dense density / sampling kernels."""
    k, d = chols_diag.shape
    _req(chols_diag, (k, d), name="chols_diag")
    dense = ctx.empty((k, d, d))
    ctx.check(ctx.lib.gmmvi_diag_embed(ctx.handle, k, d, chols_diag.ptr, dense.ptr))
    return dense


def diag_extract(ctx, dense):
    """[K,D,D] -> its diagonals [K,D]."""
    k, d, _ = dense.shape
    _req(dense, (k, d, d), name="dense")
    diag = ctx.empty((k, d))
    ctx.check(ctx.lib.gmmvi_diag_extract(ctx.handle, k, d, dense.ptr, diag.ptr))
    return diag


def reciprocal(ctx, a):
    if a.dtype != F32:
        raise ValueError("reciprocal: fp32 only")
    out = ctx.empty(a.shape)
    ctx.check(ctx.lib.gmmvi_reciprocal_f32(ctx.handle, a.ptr, a.size, out.ptr))
    return out


def update_components_diag(ctx, mode, means, chols_diag, h_neg_diag, g_neg, stepsizes, temperature, l2_init, last_eta, l2,
                           num_updates, want_info=False):
    """Diagonal-covariance component update, mode "kl" or "iblr" -> (success, kl | None, probes | None)."""
    k, d = means.shape
    _req(means, (k, d), name="means"); _req(chols_diag, (k, d), name="chols_diag")
    _req(h_neg_diag, (k, d), name="h_neg_diag"); _req(g_neg, (k, d), name="g_neg")
    _req(stepsizes, (k,), name="stepsizes"); _req(l2, (k,), name="l2"); _req(num_updates, (k,), name="num_updates")
    success = ctx.empty((k,), np.int32)
    if mode == "iblr":
        ctx.check(ctx.lib.gmmvi_update_components_diag_iblr(ctx.handle, k, d, means.ptr, chols_diag.ptr, h_neg_diag.ptr,
                                                            g_neg.ptr, stepsizes.ptr, float(l2_init), l2.ptr,
                                                            num_updates.ptr, success.ptr))
        return success, None, None
    if mode != "kl":
        raise ValueError(f"update_components_diag: unknown mode {mode!r}")
    _req(last_eta, (k,), name="last_eta")
    kl = ctx.empty((k,)) if want_info else None
    probes = ctx.empty((k,), np.int32) if want_info else None
    ctx.check(ctx.lib.gmmvi_update_components_diag_kl(ctx.handle, k, d, means.ptr, chols_diag.ptr, h_neg_diag.ptr,
                                                      g_neg.ptr, stepsizes.ptr, float(temperature), float(l2_init),
                                                      last_eta.ptr, l2.ptr, num_updates.ptr, success.ptr,
                                                      None if kl is None else kl.ptr,
                                                      None if probes is None else probes.ptr))
    return success, kl, probes


def mmd_pair_sum(ctx, a, b, inv_bandwidth):
    """sum_i sum_j exp(-sum_d inv_bandwidth[d] (a[i,d] - b[j,d])^2) -> Python float (fp64 sum)."""
    na, d = a.shape
    nb = b.shape[0]
    _req(a, (na, d), name="a"); _req(b, (nb, d), name="b"); _req(inv_bandwidth, (d,), name="inv_bandwidth")
    if na == 0 or nb == 0:
        return 0.0
    # fp64 scratch / result held in fp32-typed device buffers of twice the length (DeviceArray is 4-byte typed)
    scratch = ctx.empty((2 * int(ctx.lib.gmmvi_mmd_scratch_doubles(na, nb)),))
    out = ctx.empty((2,))
    ctx.check(ctx.lib.gmmvi_mmd_pair_sum(ctx.handle, a.ptr, na, b.ptr, nb, d, inv_bandwidth.ptr, scratch.ptr, out.ptr))
    return float(out.numpy().view(np.float64)[0])


def expected_log_ratios(ctx, ld, bg, tlp, logq, beta, logw, self_normalized=True, reward_out=None, want_ess=False):
    k, n = ld.shape
    _req(ld, (k, n), name="ld"); _req(bg, (n,), name="bg"); _req(tlp, (n,), name="tlp"); _req(logq, (n,), name="logq")
    _req(logw, (k,), name="logw")
    if reward_out is not None:
        _req(reward_out, (k,), name="reward_out")
    e = ctx.empty((k,))
    ess = ctx.empty((k,)) if want_ess else None
    ctx.check(ctx.lib.gmmvi_expected_log_ratios(ctx.handle, k, n, ld.ptr, bg.ptr, tlp.ptr, logq.ptr, float(beta),
                                                logw.ptr, 1 if self_normalized else 0, e.ptr,
                                                None if reward_out is None else reward_out.ptr,
                                                None if ess is None else ess.ptr))
    return e, ess


def update_weights(ctx, mode, logw, e, stepsize, beta, want_info=False):
    k = logw.shape[0]
    _req(logw, (k,), name="logw"); _req(e, (k,), name="E"); _req(stepsize, (1,), name="stepsize")
    info = ctx.empty((2,)) if (want_info and mode == "trust-region") else None
    if mode == "trust-region":
        ctx.check(ctx.lib.gmmvi_update_weights_kl(ctx.handle, k, logw.ptr, e.ptr, stepsize.ptr, float(beta),
                                                  None if info is None else info.ptr))
    else:
        ctx.check(ctx.lib.gmmvi_update_weights_direct(ctx.handle, k, logw.ptr, e.ptr, stepsize.ptr, float(beta)))
    return info


def component_stepsize_improvement(ctx, stepsizes, prev, last, mn, mx, inc, dec):
    k = stepsizes.shape[0]
    _req(stepsizes, (k,), name="stepsizes"); _req(prev, (k,), name="prev"); _req(last, (k,), name="last")
    ctx.check(ctx.lib.gmmvi_component_stepsize_improvement(ctx.handle, k, stepsizes.ptr, prev.ptr, last.ptr, float(mn),
                                                           float(mx), float(inc), float(dec)))


def weight_stepsize_improvement(ctx, logw, rewards_last, state, mn, mx, inc, dec):
    k = logw.shape[0]
    _req(logw, (k,), name="logw"); _req(rewards_last, (k,), name="rewards_last"); _req(state, (2,), name="state")
    ctx.check(ctx.lib.gmmvi_weight_stepsize_improvement(ctx.handle, k, logw.ptr, rewards_last.ptr, state.ptr, float(mn),
                                                        float(mx), float(inc), float(dec)))


def combine_partials(ctx, lp_parts, grad_parts, d):
    r, n = lp_parts.shape
    _req(lp_parts, (r, n), name="lp_parts")
    if grad_parts is not None:
        _req(grad_parts, (r, n, d), name="grad_parts")
    lp = ctx.empty((n,))
    grad = ctx.empty((n, d)) if grad_parts is not None else None
    ctx.check(ctx.lib.gmmvi_combine_partials(ctx.handle, r, n, d, lp_parts.ptr,
                                             None if grad_parts is None else grad_parts.ptr, lp.ptr,
                                             None if grad is None else grad.ptr))
    return lp, grad


def gather_rows(ctx, src, idx):
    """dst[i] = src[idx[i]] along axis 0 (idx: DeviceArray int32 or host array)."""
    if not isinstance(idx, DeviceArray):
        idx = ctx.asarray(np.asarray(idx, np.int32), np.int32)
    n = idx.shape[0]
    _req(idx, (n,), I32, "idx")
    inner = src.shape[1:]
    words = math.prod(inner) if inner else 1
    dst = ctx.empty((n,) + tuple(inner), src.dtype)
    if n > 0:
        ctx.check(ctx.lib.gmmvi_gather_rows(ctx.handle, src.ptr, idx.ptr, n, words, dst.ptr))
    return dst


def logaddexp(ctx, a, ca, b, cb):
    """-> log(exp(a + ca) + exp(b + cb)), element-wise (a, b: [n] float32 DeviceArrays; ca, cb: floats)."""
    if a.shape != b.shape or a.dtype != F32 or b.dtype != F32:
        raise ValueError("logaddexp: shape/dtype mismatch")
    out = ctx.empty(a.shape)
    ctx.check(ctx.lib.gmmvi_logaddexp_f32(ctx.handle, out.ptr, a.ptr, float(ca), b.ptr, float(cb), a.size))
    return out


def segment_lse_into(ctx, out, col0, offsets, logw, ld):
    """out[g, col0 + n] = log sum_{j in [offsets[g], offsets[g+1])} exp(logw[j] + ld[j, n]); out: [rows >= G, width] DeviceArray,
    offsets: int32 DeviceArray [G + 1], ld: [Kw, N]."""
    kw, n = ld.shape
    g = offsets.shape[0] - 1
    if out.ndim != 2 or out.shape[0] < g or out.shape[1] < col0 + n or logw.shape != (kw,):
        raise ValueError("segment_lse_into: shape mismatch")
    ctx.check(ctx.lib.gmmvi_segment_lse_f32(ctx.handle, g, offsets.ptr, logw.ptr, ld.ptr, n, out.ptr, out.shape[1], int(col0)))
    return out


def copy_2d(ctx, dst, dst_row, dst_col, src, src_row, src_col, rows, cols):
    """dst[dst_row + r, dst_col + c] = src[src_row + r, src_col + c] (2-D float32 DeviceArrays)."""
    if rows <= 0 or cols <= 0:
        return dst
    if (dst_row + rows > dst.shape[0] or dst_col + cols > dst.shape[1] or src_row + rows > src.shape[0]
            or src_col + cols > src.shape[1]):
        raise ValueError("copy_2d: block out of range")
    ctx.check(ctx.lib.gmmvi_copy_2d_f32(ctx.handle, dst.ptr + (dst_row * dst.shape[1] + dst_col) * 4, dst.shape[1],
                                        src.ptr + (src_row * src.shape[1] + src_col) * 4, src.shape[1], int(rows), int(cols)))
    return dst


def exp_into(ctx, dst, src):
    if dst.shape != src.shape or dst.dtype != F32 or src.dtype != F32:
        raise ValueError("exp_into: shape/dtype mismatch")
    ctx.check(ctx.lib.gmmvi_exp_f32(ctx.handle, dst.ptr, src.ptr, src.size))
    return dst


def copy_batch(ctx, pairs):
    """pairs: list of (dst DeviceArray view, src DeviceArray) of equal size; one launch per group of 8."""
    import ctypes as C
    pairs = [(d, s) for d, s in pairs if s.size > 0]
    for i in range(0, len(pairs), 8):
        grp = pairs[i:i + 8]
        for d, s in grp:
            if d.size != s.size or d.dtype != s.dtype:
                raise ValueError("copy_batch: size/dtype mismatch")
        n = len(grp)
        dst = (C.c_void_p * n)(*[d.ptr for d, _ in grp])
        src = (C.c_void_p * n)(*[s.ptr for _, s in grp])
        nb = (C.c_size_t * n)(*[s.nbytes for _, s in grp])
        ctx.check(ctx.lib.gmmvi_copy_batch(ctx.handle, n, dst, src, nb))


def concat(ctx, parts):
    """One flat fp32 buffer holding the (flattened) parts back to back; a single copy launch (<= 8 parts)."""
    total = sum(int(p.size) for p in parts)
    out = ctx.empty((total,))
    pairs, off = [], 0
    for p in parts:
        n = int(p.size)
        pairs.append((out.rows(off, off + n), p.reshape(-1)))
        off += n
    copy_batch(ctx, pairs)
    return out


def unpack_gathered(ctx, gathered, n_ranks, sizes):
    """Inverse of concat after an all-gather: gathered = [n_ranks][sum(sizes)] -> one [n_ranks * size_j] array per part."""
    import ctypes as C
    chunk = int(sum(sizes))
    if gathered.size != n_ranks * chunk:
        raise ValueError("unpack_gathered: gathered buffer has the wrong size")
    outs = [ctx.empty((n_ranks * int(sz),)) for sz in sizes]
    n = len(sizes)
    words = (C.c_size_t * n)(*[int(sz) for sz in sizes])
    dst = (C.c_void_p * n)(*[o.ptr for o in outs])
    ctx.check(ctx.lib.gmmvi_unpack_gathered(ctx.handle, gathered.ptr, int(n_ranks), chunk, n, words, dst))
    return outs
