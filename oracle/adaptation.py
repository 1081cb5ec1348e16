"""Oracle restatement of VipsComponentAdaptation (gmmvi_modules/component_adaptation.py:104-300).
TEST INFRASTRUCTURE.  Randomness (tf.random.uniform / tf.random.shuffle in the reference, :208, sample_db.py:151)
comes from a NumPy Generator handed in by the caller so that the product's host code and the oracle can be
driven by the same stream.
"""
import numpy as np
from scipy.special import logsumexp

from .gmm import FLOAT32_MAX


class FixedComponentAdaptation:
    """:88-101."""
    def adapt_number_of_components(self, iteration):
        pass


class VipsComponentAdaptation:
    def __init__(self, wrapper, sample_db, target, prior_mean, initial_cov, del_iters, add_iters, max_components,
                 thresholds_for_add_heuristic, min_weight_for_del_heuristic, num_database_samples,
                 num_prior_samples, rng):
        self.model = wrapper
        self.sample_db = sample_db
        self.target = target
        d = wrapper.num_dimensions
        if prior_mean is not None and initial_cov is not None:
            self.prior_mean = np.broadcast_to(np.asarray(prior_mean, float), (d,)).copy()      # :152-155
            self.prior_var = np.broadcast_to(np.asarray(initial_cov, float), (d,)).copy()
        else:
            self.prior_mean = self.prior_var = None
        self.num_prior_samples = num_prior_samples
        self.del_iters = del_iters
        self.add_iters = add_iters
        self.max_components = max_components
        self.num_db_samples = num_database_samples
        self.num_calls_to_add_heuristic = 0
        self.thresholds = np.asarray(thresholds_for_add_heuristic, float)
        self.min_weight_for_del_heuristic = min_weight_for_del_heuristic
        self.filter_delay = int(np.floor(del_iters / 3))                                         # :172
        xs = np.arange(-self.filter_delay, self.filter_delay, dtype=float)                       # :174
        sd = del_iters / 8.0
        kern = np.exp(-0.5 * (xs / sd) ** 2) / (sd * np.sqrt(2 * np.pi))
        self.kernel = kern / kern.sum()                                                          # :175
        self.rng = rng

    def prior_entropy(self):
        """DiagonalGMM with one component of variance initial_cov (:156); entropy of a diagonal Gaussian
        (models/diagonal_gmm.py:33-34: 0.5 D (log 2 pi + 1) + sum log chol)."""
        d = self.model.num_dimensions
        return 0.5 * d * (np.log(2 * np.pi) + 1) + np.sum(np.log(np.sqrt(self.prior_var)))

    def adapt_number_of_components(self, iteration):
        """:177-190."""
        if iteration > self.del_iters:
            self.delete_bad_components()
        if iteration > 1 and iteration % self.add_iters == 0:
            if self.model.num_components < self.max_components:
                self.add_new_component()

    def add_at_best_location(self, samples, target_lnpdfs):
        """:192-226."""
        it = self.num_calls_to_add_heuristic % len(self.thresholds)
        model_ld = self.model.log_density(samples)
        a = self.rng.random()                                                                     # :208
        if self.prior_var is not None:
            des_entropy = self.model.get_average_entropy() * a + self.prior_entropy() * (1 - a)
        else:
            des_entropy = self.model.get_average_entropy()
        max_ld = np.max(model_ld)
        rewards = target_lnpdfs - np.maximum(max_ld - self.thresholds[it], model_ld)
        new_mean = samples[np.argmax(rewards)]
        d = self.model.num_dimensions
        h_unscaled = 0.5 * d * (np.log(2.0 * np.pi) + 1)
        c = np.exp((2 * (des_entropy - h_unscaled)) / d)
        new_cov = c * np.ones(d) if self.model.diagonal_covs else c * np.eye(d)                       # :220-223
        self.model.add_component(1e-29, new_mean, new_cov, [self.thresholds[it]], [des_entropy])

    def add_new_component(self):
        """:228-259."""
        self.num_calls_to_add_heuristic += 1
        samples, lp = self.sample_db.get_random_sample(self.num_db_samples, self.rng)
        if self.num_prior_samples > 0:
            prior = self.prior_mean + np.sqrt(self.prior_var) * self.rng.standard_normal(
                (self.num_prior_samples, self.model.num_dimensions))
            self.sample_db.num_samples_written += self.num_prior_samples
            samples = np.concatenate([samples, prior])
            lp = np.concatenate([lp, self.target.log_density(prior)])
        self.add_at_best_location(samples, lp)

    def delete_bad_components(self):
        """:261-300."""
        ks = self.kernel.size
        rh = self.model.reward_history
        wh = self.model.weight_history
        cur = np.mean(rh[:, -ks:] * self.kernel[None, :], axis=1)
        old = np.mean(rh[:, -ks - self.del_iters:-self.del_iters] * self.kernel[None, :], axis=1)
        old = old - np.max(cur)
        cur = cur - np.max(cur)
        with np.errstate(divide='ignore', invalid='ignore'):
            improvements = (cur - old) / np.abs(old)
        max_actual = np.max(wh[:, -ks - self.del_iters:-1], axis=1)
        window = rh[:, -ks - self.del_iters:]
        max_greedy = np.max(np.exp(window - logsumexp(window, axis=0, keepdims=True)), axis=1)
        max_weights = np.maximum(max_actual, max_greedy)
        is_stagnating = improvements <= 0.4
        is_low_weight = max_weights < self.min_weight_for_del_heuristic
        is_old_enough = rh[:, -self.del_iters] != -FLOAT32_MAX
        bad = np.where(is_stagnating & is_low_weight & is_old_enough)[0]
        for idx in sorted(bad.tolist(), reverse=True):
            self.model.remove_component(idx)
        return bad
