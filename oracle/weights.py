"""Oracle restatement of the weight updaters (gmmvi_modules/weight_updater.py:56-100,123-141,164-279).
TEST INFRASTRUCTURE.
"""
import numpy as np
from scipy.special import logsumexp

LOG_WEIGHT_FLOOR = -69.07    # weight_updater.py:139, :187 ("lower bound weights to 1e-30")


def get_expected_log_ratios(wrapper, samples, background_mixture_densities, target_lnpdfs, temperature,
                            use_self_normalized_importance_weights=True):
    """:56-75.  Also stores the component rewards (:73-74)."""
    model_densities, cld = wrapper.log_densities_also_individual(samples)
    log_ratios = target_lnpdfs - temperature * model_densities
    if use_self_normalized_importance_weights:
        lw = cld - background_mixture_densities[None, :]
        lw = lw - logsumexp(lw, axis=1, keepdims=True)
        w = np.exp(lw)
        iw = w / np.sum(w, axis=1, keepdims=True)
        expected_log_ratios = iw @ log_ratios
    else:
        n = samples.shape[0]
        lw = cld - background_mixture_densities[None, :]
        with np.errstate(divide='ignore'):
            a = lw + np.log(np.abs(log_ratios))[None, :]
        m = np.max(a, axis=1, keepdims=True)
        expected_log_ratios = (1.0 / n) * np.sum(np.sign(log_ratios)[None, :] * np.exp(a - m), axis=1) * np.exp(m[:, 0])
    wrapper.store_rewards(temperature * wrapper.log_weights + expected_log_ratios)
    return expected_log_ratios


def direct_update(wrapper, expected_log_ratios, stepsize, temperature):
    """:123-141."""
    if wrapper.num_components > 1:
        u = wrapper.log_weights + stepsize / temperature * expected_log_ratios
        nl = u - logsumexp(u)
        nl = np.maximum(nl, LOG_WEIGHT_FLOOR)
        nl = nl - logsumexp(nl)
        wrapper.replace_weights(nl)


def weights_kl(eta, log_weights, component_rewards, temperature):
    """:164-191."""
    u = (eta + 1) / (temperature + eta) * log_weights + 1.0 / (temperature + eta) * component_rewards
    nl = u - logsumexp(u)
    nl = np.maximum(nl, LOG_WEIGHT_FLOOR)
    nl = nl - logsumexp(nl)
    return np.sum(np.exp(nl) * (nl - log_weights)), nl


def weights_bracketing_search(log_weights, expected_log_ratios, kl_bound, temperature, lower_bound=-45.0,
                              upper_bound=45.0):
    """:193-260.  Returns (kl, eta, new_log_weights)."""
    log_eta = 0.5 * (upper_bound + lower_bound)
    ub_ok = False
    kl, eta = -1.0, -1.0
    new_lw = log_weights
    for _ in range(50):
        eta = np.exp(log_eta)
        if abs(np.exp(upper_bound) - np.exp(lower_bound)) < 1e-1:
            break
        kl, new_lw = weights_kl(eta, log_weights, expected_log_ratios, temperature)
        if abs(kl_bound - kl) < 1e-1 * kl_bound:
            lower_bound = upper_bound                 # :242 (sets lb = ub, not = log_eta)
            break
        if kl_bound > kl:
            upper_bound = log_eta
            ub_ok = True
        else:
            lower_bound = log_eta
        log_eta = 0.5 * (upper_bound + lower_bound)
    if lower_bound == upper_bound:
        return kl, eta, new_lw
    if ub_ok:
        kl, new_lw = weights_kl(np.exp(upper_bound), log_weights, expected_log_ratios, temperature)
        return kl, np.exp(upper_bound), new_lw
    return -1.0, -1.0, log_weights


def trust_region_update(wrapper, expected_log_ratios, kl_bound, temperature):
    """:262-279."""
    if wrapper.num_components > 1:
        kl, eta, nl = weights_bracketing_search(wrapper.log_weights, expected_log_ratios, kl_bound, temperature)
        wrapper.replace_weights(nl)
        return kl, eta
    return -1.0, -1.0
