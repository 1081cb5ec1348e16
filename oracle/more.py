"""Oracle restatement of MoreNgEstimator (gmmvi_modules/ng_estimator.py:289-376) and the quadratic
ridge regression it uses (optimization/least_squares.py:34-76,103-191).  TEST INFRASTRUCTURE.
"""
import numpy as np
from scipy.special import logsumexp

from .stein import get_rewards_for_comp


def quad_features(x):
    """least_squares.py:113-124: [x_i * x_j (i <= j, row-major upper triangle), x, 1]."""
    n, d = x.shape
    cols = [x[:, i:i + 1] * x[:, i:] for i in range(d)]
    return np.concatenate(cols + [x, np.ones((n, 1), x.dtype)], axis=1)


def fit_quadratic(regularizer, inputs, outputs, weights, sample_mean, sample_chol_cov):
    """least_squares.py:126-191 (whitening always on) + RegressionFunc.fit :34-76 (bias unregularised)."""
    d = inputs.shape[1]
    inv_chol = np.linalg.inv(sample_chol_cov)                                   # :172
    z = (inputs - sample_mean) @ inv_chol.T                                     # :173
    phi = quad_features(z)
    f = phi.shape[1]
    wphi_t = (weights[:, None] * phi).T                                         # :65
    reg = np.eye(f, dtype=phi.dtype) * regularizer
    reg[-1, -1] = 0.0                                                           # :71-73
    params = np.linalg.solve(wphi_t @ phi + reg, wphi_t @ outputs)              # :74-75
    qt = np.zeros((d, d), phi.dtype)
    qt[np.triu_indices(d)] = params[:-(d + 1)]                                  # :177
    quad = -qt - qt.T                                                           # :179
    lin = params[-(d + 1):-1]
    const = params[-1]
    quad = inv_chol.T @ quad @ inv_chol                                         # :185
    t1 = inv_chol.T @ lin
    t2 = quad @ sample_mean
    lin = t1 + t2                                                               # :186-188
    const = const + np.sum(sample_mean * (-0.5 * t2 - t1))                      # :189
    return quad, lin, const


def get_expected_hessian_and_grad(model, l2_regularizers, samples, mapping, background_densities, target_lnpdfs,
                                  only_use_own_samples=False, use_self_normalized_importance_weights=True):
    """ng_estimator.py:296-376."""
    k = model.num_components
    relative_mapping = mapping - (np.max(mapping) if mapping.size else 0) + k - 1
    model_densities, cld = model.log_densities_also_individual(samples)
    log_ratios = target_lnpdfs - model_densities
    dummy_grads = np.zeros_like(samples)
    hs, gs = [], []
    for i in range(k):
        xs, rewards, _, bg, my_cld = get_rewards_for_comp(i, samples, relative_mapping, cld, log_ratios, dummy_grads,
                                                          background_densities, only_use_own_samples)
        lw = my_cld - bg
        if use_self_normalized_importance_weights:
            lw = lw - logsumexp(lw)
            w = np.exp(lw)
            iw = w / np.sum(w)
        else:
            iw = np.exp(lw)
        quad, lin, _ = fit_quadratic(l2_regularizers[i], xs, rewards, iw, model.means[i], model.chol_cov[i])
        hs.append(quad)                                                          # :369-370
        gs.append(quad @ model.means[i] - lin)                                   # :371-373
    return np.stack(hs), np.stack(gs)
