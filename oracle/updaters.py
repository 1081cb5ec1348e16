"""Oracle restatement of the natural-gradient component updaters
(gmmvi_modules/ng_based_component_updater.py:97-141 direct, :160-223 iBLR, :244-524 KL-constrained).
TEST INFRASTRUCTURE.

Cholesky failure: the reference detects it by NaN in TensorFlow's output (:120, :202, :320, :493); NumPy
raises instead, so ``_chol`` maps LinAlgError / non-finite input to ``None`` and callers take the same branch.
"""
import numpy as np
from scipy.linalg import solve_triangular, cho_solve

from .gmm import FLOAT32_MAX


def _chol(a):
    if not np.all(np.isfinite(a)):
        return None
    try:
        c = np.linalg.cholesky(a)
    except np.linalg.LinAlgError:
        return None
    return c if np.all(np.isfinite(c)) else None


def _tri_inv(l):
    return solve_triangular(l, np.eye(l.shape[0], dtype=l.dtype), lower=True)


def _update_l2(wrapper, successes):
    """:135-138 / :217-220 / :520-523 -- note min(1e-6, 10*l2) on failure."""
    wrapper.l2_regularizers = np.where(successes,
                                       np.maximum(0.5 * wrapper.l2_regularizers, wrapper.initial_regularizer),
                                       np.minimum(1e-6, 10 * wrapper.l2_regularizers))


# ---------------------------------------------------------------------------------------------------
# KL-constrained ("T")
# ---------------------------------------------------------------------------------------------------

def kl(eta, old_lin, old_prec, old_inv_chol, reward_lin, reward_quad, kl_const_part, old_mean, eta_in_logspace):
    """:244-333.  Returns (kl, new_mean, new_precision, inv_chol_inv).  Diagonal branch (:304-318, all operands
    [D]): no failure test -- a negative new precision gives a NaN KL, which the search treats as "KL too large"
    (both comparisons false, :410-419) and the final ``kl < float32.max`` test rejects (:488)."""
    if eta_in_logspace:
        eta = np.exp(eta)
    new_lin = (eta * old_lin + reward_lin) / eta
    new_prec = (eta * old_prec + reward_quad) / eta
    if np.ndim(old_prec) == 1:
        with np.errstate(invalid='ignore', divide='ignore'):
            chol_prec = np.sqrt(new_prec)
            new_mean = 1.0 / new_prec * new_lin
            inv_chol_inv = 1.0 / chol_prec
            diff = old_mean - new_mean
            inner = np.sum(np.log(new_prec / old_prec) + old_prec / new_prec) - old_mean.shape[0]
            inner = inner if np.isnan(inner) else max(0.0, inner)              # tf.maximum propagates NaN
            val = 0.5 * (inner + np.sum(np.square(old_inv_chol * diff)))                          # :314-317
        return val, new_mean, new_prec, inv_chol_inv
    c = _chol(new_prec)
    if c is None:                                                    # :320-324
        return FLOAT32_MAX, old_mean, old_prec, old_inv_chol
    new_mean = cho_solve((c, True), new_lin)                         # :326
    new_logdet = -2.0 * np.sum(np.log(np.diag(c)))                   # :327
    cinv = _tri_inv(c)                                               # :328
    trace_term = np.sum(np.square(cinv @ old_inv_chol.T))            # :329  ||C^-1 L^-T||_F^2
    diff = old_mean - new_mean
    val = 0.5 * (kl_const_part - new_logdet + trace_term + np.sum(np.square(old_inv_chol @ diff)))   # :331-332
    return val, new_mean, new_prec, cinv


def bracketing_search(kl_bound, lower_bound, upper_bound, old_lin, old_prec, old_inv_chol, reward_lin, reward_quad,
                      kl_const_part, old_mean, eta_in_logspace=True, trace=None):
    """:335-429.  Returns (new_lower, new_upper) in linear eta space."""
    eta = 0.5 * (upper_bound + lower_bound)
    ub_ok = False
    for _ in range(1000):
        if eta_in_logspace:
            diff = min(np.exp(upper_bound) - np.exp(eta), np.exp(eta) - np.exp(lower_bound))
        else:
            diff = min(upper_bound - eta, eta - lower_bound)
        if diff < 1e-1:
            break
        val = kl(eta, old_lin, old_prec, old_inv_chol, reward_lin, reward_quad, kl_const_part, old_mean,
                 eta_in_logspace)[0]
        if trace is not None:
            trace.append((float(eta), float(val)))
        if abs(kl_bound - val) < 1e-1 * kl_bound:
            lower_bound = upper_bound = eta
            break
        if kl_bound > val:
            upper_bound = eta
            ub_ok = True
        else:
            lower_bound = eta
        eta = 0.5 * (upper_bound + lower_bound)
    if ub_ok:
        lower_bound = upper_bound
    if eta_in_logspace:
        return np.exp(lower_bound), np.exp(upper_bound)
    return lower_bound, upper_bound


def apply_ng_update_kl(wrapper, expected_hessians_neg, expected_gradients_neg, stepsizes, temperature, traces=None):
    """:431-524.  Mutates the wrapper exactly like the reference; returns (successes, etas, kls, n_probes)."""
    model = wrapper.model
    dt = model.dtype
    k, d = model.means.shape
    means, chols, succ, etas, kls, nprobes = [], [], [], [], [], []
    for i in range(k):
        old_chol, old_mean = model.chol_cov[i], model.means[i]
        last_eta = wrapper.last_log_etas[i]
        eps = stepsizes[i]
        reward_quad = expected_hessians_neg[i]
        diag = np.ndim(old_chol) == 1                                               # model.diagonal_covs
        if diag:                                                                    # :447-453
            reward_lin = reward_quad * old_mean - expected_gradients_neg[i]
            old_logdet = 2.0 * np.sum(np.log(old_chol))
            old_inv_chol = 1.0 / old_chol
            old_prec = old_inv_chol ** 2
            old_lin = old_prec * old_mean
        else:
            reward_lin = reward_quad @ old_mean - expected_gradients_neg[i]        # :455
            old_logdet = 2.0 * np.sum(np.log(np.diag(old_chol)))                    # :456
            old_inv_chol = _tri_inv(old_chol)                                       # :457
            old_prec = old_inv_chol.T @ old_inv_chol                                # :458
            old_lin = old_prec @ old_mean                                           # :459
        kl_const = old_logdet - d                                                   # :460
        if last_eta < 0:                                                            # :462-471
            lb, ub = dt(-20.0), dt(80.0)
        else:
            lb, ub = max(dt(0.0), np.log(last_eta) - 3), np.log(last_eta) + 3
        tr = [] if traces is not None else None
        lo, hi = bracketing_search(eps, lb, ub, old_lin, old_prec, old_inv_chol, reward_lin, reward_quad, kl_const,
                                   old_mean, True, trace=tr)
        if traces is not None:
            traces.append(tr)
        eta = max(lo, temperature)                                                  # :476
        success = False
        new_mean, new_chol, this_kl = old_mean, old_chol, -1.0
        if lo == hi:                                                                # :478
            success = True
            this_kl, new_mean, _, cinv = kl(eta, old_lin, old_prec, old_inv_chol, reward_lin, reward_quad,
                                            kl_const, old_mean, False)
            new_cov = np.square(cinv) if diag else cinv.T @ cinv                    # :483-486
            if this_kl < FLOAT32_MAX:
                new_chol = np.sqrt(new_cov) if diag else _chol(new_cov)             # :489-492
                if new_chol is None:
                    success = False
            else:
                success = False
        if success:
            means.append(new_mean); chols.append(new_chol); succ.append(True); etas.append(eta); kls.append(this_kl)
        else:
            means.append(old_mean); chols.append(old_chol); succ.append(False); etas.append(-1.0); kls.append(-1.0)
        nprobes.append(len(tr) if tr is not None else -1)
    succ = np.array(succ)
    model.replace_components(np.stack(means), np.stack(chols))                     # :518
    wrapper.num_received_updates = wrapper.num_received_updates + 1                 # :519
    _update_l2(wrapper, succ)
    wrapper.last_log_etas = np.array(etas, dt)                                      # :524 (stores eta, not log eta)
    return succ, np.array(etas, dt), np.array(kls, dt), np.array(nprobes)


# ---------------------------------------------------------------------------------------------------
# direct ("I") and iBLR ("Y")
# ---------------------------------------------------------------------------------------------------

def apply_ng_update_direct(wrapper, expected_hessians_neg, expected_gradients_neg, stepsizes):
    """:97-141."""
    model = wrapper.model
    k = model.num_components
    means, chols, succ = [], [], []
    for i in range(k):
        old_chol, old_mean = model.chol_cov[i], model.means[i]
        old_inv_chol = _tri_inv(old_chol)
        old_prec = old_inv_chol.T @ old_inv_chol
        old_lin = old_prec @ old_mean
        delta_prec = expected_hessians_neg[i]
        delta_lin = expected_hessians_neg[i] @ old_mean - expected_gradients_neg[i]
        new_lin = old_lin + stepsizes[i] * delta_lin
        new_prec = old_prec + stepsizes[i] * delta_prec
        ok = True
        try:
            new_mean = np.linalg.solve(new_prec, new_lin)
            new_chol = _chol(np.linalg.inv(new_prec))
        except np.linalg.LinAlgError:
            new_chol = None
        if new_chol is None or not np.all(np.isfinite(new_mean)):
            ok, new_mean, new_chol = False, old_mean, old_chol
        means.append(new_mean); chols.append(new_chol); succ.append(ok)
    succ = np.array(succ)
    _update_l2(wrapper, succ)
    model.replace_components(np.stack(means), np.stack(chols))
    wrapper.num_received_updates = wrapper.num_received_updates + 1
    return succ


def apply_ng_update_iblr(wrapper, expected_hessians_neg, expected_gradients_neg, stepsizes):
    """:160-223; the first update of a component leaves its mean alone (:184-186).  Diagonal branch: elementwise
    (:170-174, :188-189, :195-197), failure = NaN in sqrt(1 / new_precision) (:202)."""
    model = wrapper.model
    k = model.num_components
    means, chols, succ = [], [], []
    for i in range(k):
        old_chol, old_mean = model.chol_cov[i], model.means[i]
        h = expected_hessians_neg[i]
        if np.ndim(old_chol) == 1:                                                   # model.diagonal_covs
            correction = stepsizes[i] / 2 * h * old_chol * old_chol * h
            old_prec = (1.0 / old_chol) ** 2
            delta_mean = -expected_gradients_neg[i]
            if wrapper.num_received_updates[i] == 0:
                new_mean = old_mean
            else:
                new_mean = old_mean + stepsizes[i] * old_chol * old_chol * delta_mean
            new_prec = old_prec + stepsizes[i] * (h + correction)
            with np.errstate(invalid='ignore', divide='ignore'):
                new_chol = np.sqrt(1.0 / new_prec)
            ok = not np.any(np.isnan(new_chol))
            if not ok:
                new_mean, new_chol = old_mean, old_chol
            means.append(new_mean); chols.append(new_chol); succ.append(ok)
            continue
        correction = stepsizes[i] / 2 * h @ old_chol @ old_chol.T @ h                # :176-177
        old_inv_chol = _tri_inv(old_chol)
        old_prec = old_inv_chol.T @ old_inv_chol
        delta_prec = h + correction
        delta_mean = -expected_gradients_neg[i]
        if wrapper.num_received_updates[i] == 0:
            new_mean = old_mean
        else:
            new_mean = old_mean + stepsizes[i] * old_chol @ old_chol.T @ delta_mean   # :191-192
        new_prec = old_prec + stepsizes[i] * delta_prec
        ok = True
        try:
            new_chol = _chol(np.linalg.inv(new_prec))
        except np.linalg.LinAlgError:
            new_chol = None
        if new_chol is None:
            ok, new_mean, new_chol = False, old_mean, old_chol
        means.append(new_mean); chols.append(new_chol); succ.append(ok)
    succ = np.array(succ)
    _update_l2(wrapper, succ)
    model.replace_components(np.stack(means), np.stack(chols))
    wrapper.num_received_updates = wrapper.num_received_updates + 1
    return succ
