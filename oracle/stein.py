"""Oracle restatement of SteinNgEstimator (gmmvi_modules/ng_estimator.py:107-120,146-263).
TEST INFRASTRUCTURE.
"""
import numpy as np
from scipy.linalg import cho_solve
from scipy.special import logsumexp


def _signed_expectation(log_weights, values):
    """ng_estimator.py:146-152 (_stable_expectation): 1/n * sum_n exp(lw_n) * v_n, evaluated through a
    signed log-sum-exp (tfp.math.reduce_weighted_logsumexp with w = sign(v)).  log_weights is [n] or
    [n, 1] (the Hessian call site adds one axis only, :167, so it broadcasts over the LAST matrix axis
    exactly like the reference's expand_dims(.., 1))."""
    n = values.shape[0]
    lw = log_weights.reshape((n,) + (1,) * (values.ndim - 1))
    with np.errstate(divide='ignore'):
        a = lw + np.log(np.abs(values))
    m = np.max(a, axis=0, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    s = np.sum(np.sign(values) * np.exp(a - m), axis=0)
    return (1.0 / n) * s * np.exp(m[0])


def get_rewards_for_comp(index, samples, mapping, component_log_densities, log_ratios, log_ratio_grads,
                         background_densities, only_use_own_samples):
    """ng_estimator.py:107-120."""
    if only_use_own_samples:
        own = np.where(mapping == index)[0]
        return (samples[own], log_ratios[own], log_ratio_grads[own],
                component_log_densities[index][own], component_log_densities[index][own])
    return samples, log_ratios, log_ratio_grads, background_densities, component_log_densities[index]


def expected_gradient_and_hessian_self_normalized(chol_cov, mean, component_log_densities, samples,
                                                  background_densities, log_ratio_grads):
    """ng_estimator.py:171-188; note the double normalisation (:174-176) and the orientation
    H[i, j] = sum_n wg[n, i] * y[n, j] before symmetrising (:184-186).  Diagonal branch (:178-181, chol_cov [D]):
    only the diagonal  h[i] = sum_n wg[n, i] (x[n, i] - mu[i]) / sigma[i]^2  is estimated."""
    lw = component_log_densities - background_densities
    lw = lw - logsumexp(lw)
    w = np.exp(lw)
    iw = w / np.sum(w)
    wg = iw[:, None] * log_ratio_grads
    if chol_cov.ndim == 1:
        prec_times_diff = (1.0 / np.square(chol_cov))[:, None] * (samples - mean).T             # [D, N]
        return np.sum(wg, axis=0), np.sum(prec_times_diff.T * wg, axis=0)
    y = cho_solve((chol_cov, True), (samples - mean).T)          # Sigma^{-1}(x - mu): [D, N]
    h = np.einsum('ni,nj->ij', wg, y.T)
    h = 0.5 * (h + h.T)
    return np.sum(wg, axis=0), h


def expected_gradient_and_hessian_standard(chol_cov, mean, component_log_densities, samples,
                                           background_densities, log_ratio_grads):
    """ng_estimator.py:154-169: plain importance weights, NOT symmetrised (diagonal branch :159-162)."""
    lw = component_log_densities - background_densities
    g = _signed_expectation(lw, log_ratio_grads)
    if chol_cov.ndim == 1:
        prec_times_diff = (1.0 / np.square(chol_cov))[:, None] * (samples - mean).T
        return g, _signed_expectation(lw, prec_times_diff.T * log_ratio_grads)
    y = cho_solve((chol_cov, True), (samples - mean).T)
    prod = y.T[:, None, :] * log_ratio_grads[:, :, None]         # [n, i, j] = g[n, i] * y[n, j]   (:165-166)
    h = _signed_expectation(lw, prod)
    return g, h


def get_expected_hessian_and_grad(model, samples, mapping, background_densities, target_lnpdfs, target_lnpdf_grads,
                                  only_use_own_samples=False, use_self_normalized_importance_weights=True):
    """ng_estimator.py:204-263.  Returns (expected_hessian_neg [K,D,D] -- [K,D] for a diagonal model --, expected_gradient_neg [K,D])."""
    k = model.num_components
    relative_mapping = mapping - (np.max(mapping) if mapping.size else 0) + k - 1        # :244
    model_densities, model_grads, cld = model.log_density_and_grad(samples)            # :246
    log_ratios = target_lnpdfs - model_densities                                        # :247
    log_ratio_grads = target_lnpdf_grads - model_grads                                  # :248
    hs, gs = [], []
    for i in range(k):
        xs, _, gr, bg, my_cld = get_rewards_for_comp(i, samples, relative_mapping, cld, log_ratios,
                                                     log_ratio_grads, background_densities, only_use_own_samples)
        if use_self_normalized_importance_weights:
            g, h = expected_gradient_and_hessian_self_normalized(model.chol_cov[i], model.means[i], my_cld, xs, bg, gr)
        else:
            g, h = expected_gradient_and_hessian_standard(model.chol_cov[i], model.means[i], my_cld, xs, bg, gr)
        hs.append(-h)
        gs.append(-g)
    return np.stack(hs), np.stack(gs)
