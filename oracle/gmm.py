"""Oracle restatement of the full-covariance and diagonal GMMs and their wrapper.  TEST INFRASTRUCTURE.

Follows models/full_cov_gmm.py:29-68, models/diagonal_gmm.py:6-59, models/gmm.py:124-216,249-300,340-418 and
models/gmm_wrapper.py:33-182 (paths relative to /root/reference/src/gmmvi).
Gradients that the reference obtains by reverse-mode AD (models/gmm.py:294-300) are written
analytically here and verified by central finite differences in tests/test_oracle_gmm.py.
"""
import numpy as np
from scipy.linalg import solve_triangular
from scipy.special import logsumexp

from . import philox

FLOAT32_MIN = float(np.finfo(np.float32).min)   # tf.float32.min == -3.4028235e38 (gmm_wrapper.py:72)
FLOAT32_MAX = float(np.finfo(np.float32).max)


def gaussian_log_pdf_from_chol(x, mean, chol):
    """log N(x; mean, chol chol^T) for x[N,D]  (models/full_cov_gmm.py:41-47)."""
    d = x.shape[1]
    z = solve_triangular(chol, (x - mean).T, lower=True)
    return -0.5 * np.sum(z * z, axis=0) - np.sum(np.log(np.diag(chol))) - 0.5 * d * np.log(2 * np.pi)


class FullCovGMM:
    """models/full_cov_gmm.py:6-68 + models/gmm.py:5-418, NumPy state instead of tf.Variable."""
    diagonal_covs = False

    def __init__(self, weights, means, covs, dtype=np.float64):
        self.dtype = dtype
        self.diagonal_covs = False
        means = np.asarray(means, dtype=dtype)
        self.num_dimensions = means.shape[1]
        self.means = means.copy()
        self.chol_cov = np.stack([np.linalg.cholesky(np.asarray(c, dtype=dtype)) for c in covs])  # full_cov_gmm.py:23
        self.log_weights = np.log(np.asarray(weights, dtype=dtype))
        self.replace_weights(self.log_weights)                                                   # gmm.py:34

    # ---- properties -------------------------------------------------------------------------
    @property
    def num_components(self):
        return self.log_weights.shape[0]

    @property
    def weights(self):
        return np.exp(self.log_weights)                                                           # gmm.py:171

    @property
    def covs(self):
        return self.chol_cov @ np.transpose(self.chol_cov, (0, 2, 1))                            # full_cov_gmm.py:31

    # ---- densities --------------------------------------------------------------------------
    def component_log_densities(self, samples):
        """[K, N]  (full_cov_gmm.py:56-62)."""
        samples = np.asarray(samples, dtype=self.dtype)
        k, d = self.means.shape
        out = np.empty((k, samples.shape[0]), dtype=self.dtype)
        for i in range(k):
            z = solve_triangular(self.chol_cov[i], (samples - self.means[i]).T, lower=True)
            const = -0.5 * np.sum(np.log(np.square(np.diag(self.chol_cov[i])))) - 0.5 * d * np.log(2 * np.pi)
            out[i] = -0.5 * np.sum(z * z, axis=0) + const
        return out

    def log_densities_also_individual(self, samples):
        """gmm.py:183-201."""
        cld = self.component_log_densities(samples)
        return logsumexp(cld + self.log_weights[:, None], axis=0), cld

    def log_density(self, samples):
        """gmm.py:203-216."""
        return self.log_densities_also_individual(samples)[0]

    def log_density_and_grad(self, samples):
        """gmm.py:274-300.  grad_x log q = -sum_k r_k Sigma_k^{-1} (x - mu_k), r = softmax_k(log w_k + ld_k)."""
        samples = np.asarray(samples, dtype=self.dtype)
        logq, cld = self.log_densities_also_individual(samples)
        resp = np.exp(cld + self.log_weights[:, None] - logq[None, :])           # [K, N]
        grad = np.zeros_like(samples)
        for i in range(self.num_components):
            z = solve_triangular(self.chol_cov[i], (samples - self.means[i]).T, lower=True)
            y = solve_triangular(self.chol_cov[i], z, lower=True, trans='T')     # Sigma^{-1}(x-mu), [D, N]
            grad -= (resp[i][None, :] * y).T
        return logq, grad, cld

    def component_marginal_log_densities(self, samples, dim):
        """full_cov_gmm.py:49-54."""
        var = self.covs[:, dim, dim]
        diffs = samples[None, :, dim] - self.means[:, dim, None]
        return -0.5 * diffs * diffs / var[:, None] - 0.5 * np.log(var)[:, None] - 0.5 * np.log(2 * np.pi)

    def marginal_log_density(self, samples, dim):
        """gmm.py:218-234."""
        return logsumexp(self.component_marginal_log_densities(samples, dim) + self.log_weights[:, None], axis=0)

    # ---- entropies --------------------------------------------------------------------------
    def gaussian_entropy(self, chol):
        """full_cov_gmm.py:33-34."""
        return 0.5 * self.num_dimensions * (np.log(2 * np.pi) + 1) + np.sum(np.log(np.diag(chol)))

    def component_entropies(self):
        """gmm.py:249-260."""
        return np.array([self.gaussian_entropy(c) for c in self.chol_cov], dtype=self.dtype)

    def get_average_entropy(self):
        """gmm.py:262-272."""
        return float(np.sum(np.exp(self.log_weights) * self.component_entropies()))

    # ---- sampling ---------------------------------------------------------------------------
    def sample_from_components_no_shuffle(self, samples_per_component, eps):
        """gmm.py:361-386 + full_cov_gmm.py:36-39 with the normals supplied: x = mu_k + L_k eps.
        eps is [sum(n_k), D] in component order."""
        n_k = np.asarray(samples_per_component, dtype=np.int64)
        mapping = np.repeat(np.arange(self.num_components, dtype=np.int32), n_k)
        eps = np.asarray(eps, dtype=self.dtype)
        x = self.means[mapping] + np.einsum('nij,nj->ni', self.chol_cov[mapping], eps)
        return x, mapping

    def sample_categorical_from_uniform(self, u):
        """gmm.py:124-137: argmax(u < cumsum(weights)) -> first threshold exceeding u."""
        thresholds = np.cumsum(self.weights)
        return np.argmax(u[:, None] < thresholds[None, :], axis=1).astype(np.int32)

    def sample_with(self, u, eps):
        """gmm.py:139-163: categorical draw, then samples grouped by component (component order),
        returned together with the *unsorted* component indices exactly as the reference does."""
        comp = self.sample_categorical_from_uniform(np.asarray(u, dtype=self.dtype))
        counts = np.bincount(comp, minlength=self.num_components)
        x, _ = self.sample_from_components_no_shuffle(counts, eps)
        return x, comp

    def sample(self, num_samples, seed, first_index):
        """Philox-driven GMM.sample (stream ids in oracle/philox.py)."""
        u = philox.uniform01(seed, first_index, num_samples, philox.STREAM_CATEGORICAL)
        eps = philox.normals(seed, first_index, num_samples, self.num_dimensions, philox.STREAM_MIXTURE_NORMALS)
        return self.sample_with(u, eps)

    # ---- mutation ---------------------------------------------------------------------------
    def replace_weights(self, new_log_weights):
        """gmm.py:173-181."""
        new_log_weights = np.asarray(new_log_weights, dtype=self.dtype)
        self.log_weights = new_log_weights - logsumexp(new_log_weights)

    def replace_components(self, new_means, new_chols):
        """gmm.py:401-418."""
        self.means = np.asarray(new_means, dtype=self.dtype).copy()
        self.chol_cov = np.asarray(new_chols, dtype=self.dtype).copy()

    def add_component(self, initial_weight, initial_mean, initial_cov):
        """full_cov_gmm.py:64-68."""
        self.means = np.concatenate([self.means, np.asarray(initial_mean, self.dtype)[None]], axis=0)
        self.chol_cov = np.concatenate(
            [self.chol_cov, np.linalg.cholesky(np.asarray(initial_cov, self.dtype))[None]], axis=0)
        self.replace_weights(np.concatenate([self.log_weights, [np.log(self.dtype(initial_weight))]]))

    def remove_component(self, idx):
        """gmm.py:388-398."""
        self.replace_weights(np.delete(self.log_weights, idx))
        self.means = np.delete(self.means, idx, axis=0)
        self.chol_cov = np.delete(self.chol_cov, idx, axis=0)


class DiagonalGMM(FullCovGMM):
    """models/diagonal_gmm.py:6-59: chol_cov is [K, D] (square roots of the diagonal covariance entries)."""

    def __init__(self, weights, means, covs, dtype=np.float64):
        self.dtype = dtype
        means = np.asarray(means, dtype=dtype)
        self.num_dimensions = means.shape[1]
        self.means = means.copy()
        self.chol_cov = np.sqrt(np.asarray(covs, dtype=dtype))                                    # diagonal_gmm.py:23
        self.log_weights = np.log(np.asarray(weights, dtype=dtype))
        self.replace_weights(self.log_weights)
        self.diagonal_covs = True                                                                # :28

    @property
    def covs(self):
        return np.square(self.chol_cov)                                                          # :36-38

    @staticmethod
    def diagonal_gaussian_log_pdf(dim, mean, chol, x):
        """:30-34."""
        const = -0.5 * dim * np.log(2 * np.pi) - np.sum(np.log(chol))
        return const - 0.5 * np.sum(np.square((1.0 / chol)[None, :] * (mean[None, :] - x)), axis=1)

    def component_log_densities(self, samples):
        """:47-53 -> [K, N]."""
        samples = np.asarray(samples, dtype=self.dtype)
        return np.stack([self.diagonal_gaussian_log_pdf(self.num_dimensions, self.means[i], self.chol_cov[i], samples)
                         for i in range(self.num_components)])

    def log_density_and_grad(self, samples):
        """gmm.py:274-300 (reverse-mode AD upstream): grad = -sum_k r_k (x - mu_k) / sigma_k^2."""
        samples = np.asarray(samples, dtype=self.dtype)
        logq, cld = self.log_densities_also_individual(samples)
        resp = np.exp(cld + self.log_weights[:, None] - logq[None, :])
        grad = np.zeros_like(samples)
        for i in range(self.num_components):
            grad -= resp[i][:, None] * (samples - self.means[i]) / np.square(self.chol_cov[i])[None, :]
        return logq, grad, cld

    def component_marginal_log_densities(self, samples, dim):
        var = self.covs[:, dim]
        diffs = samples[None, :, dim] - self.means[:, dim, None]
        return -0.5 * diffs * diffs / var[:, None] - 0.5 * np.log(var)[:, None] - 0.5 * np.log(2 * np.pi)

    def gaussian_entropy(self, chol):
        """:40-41."""
        return 0.5 * self.num_dimensions * (np.log(2 * np.pi) + 1) + np.sum(np.log(chol))

    def sample_from_components_no_shuffle(self, samples_per_component, eps):
        """gmm.py:361-386 + diagonal_gmm.py:43-45: x = mu_k + sigma_k * eps."""
        n_k = np.asarray(samples_per_component, dtype=np.int64)
        mapping = np.repeat(np.arange(self.num_components, dtype=np.int32), n_k)
        eps = np.asarray(eps, dtype=self.dtype)
        return self.means[mapping] + self.chol_cov[mapping] * eps, mapping

    def add_component(self, initial_weight, initial_mean, initial_cov):
        """:55-59."""
        self.means = np.concatenate([self.means, np.asarray(initial_mean, self.dtype)[None]], axis=0)
        self.chol_cov = np.concatenate([self.chol_cov, np.sqrt(np.asarray(initial_cov, self.dtype))[None]], axis=0)
        self.replace_weights(np.concatenate([self.log_weights, [np.log(self.dtype(initial_weight))]]))


class GmmWrapper:
    """models/gmm_wrapper.py:4-182: per-component learner metadata beside the model."""

    def __init__(self, model, initial_stepsize, initial_regularizer, max_reward_history_length):
        self.model = model
        dt = model.dtype
        k = model.num_components
        self.initial_regularizer = initial_regularizer
        self.initial_last_eta = -1
        self.initial_stepsize = initial_stepsize
        self.max_reward_history_length = max_reward_history_length
        self.l2_regularizers = initial_regularizer * np.ones(k, dt)                       # :68
        self.last_log_etas = self.initial_last_eta * np.ones(k, dt)                       # :69
        self.num_received_updates = np.zeros(k, dt)                                       # :70
        self.stepsizes = initial_stepsize * np.ones(k, dt)                                # :71
        self.reward_history = FLOAT32_MIN * np.ones((k, max_reward_history_length), dt)   # :72
        self.weight_history = FLOAT32_MIN * np.ones((k, max_reward_history_length), dt)   # :74
        self.unique_component_ids = np.arange(k, dtype=np.int32)                          # :76
        self.max_component_id = int(self.unique_component_ids.max())                      # :77
        self.adding_thresholds = -np.ones(k, dt)                                          # :79
        self.initial_entropies = model.component_entropies()                              # :80

    def __getattr__(self, name):                                                          # :83-88
        return getattr(self.__dict__['model'], name)

    def add_component(self, initial_weight, initial_mean, initial_cov, adding_threshold, initial_entropy):
        """:90-127."""
        dt = self.model.dtype
        self.model.add_component(initial_weight, initial_mean, initial_cov)
        self.max_component_id += 1
        self.unique_component_ids = np.append(self.unique_component_ids, np.int32(self.max_component_id))
        self.l2_regularizers = np.append(self.l2_regularizers, dt(self.initial_regularizer))
        self.last_log_etas = np.append(self.last_log_etas, dt(self.initial_last_eta))
        self.num_received_updates = np.append(self.num_received_updates, dt(0))
        self.stepsizes = np.append(self.stepsizes, dt(self.initial_stepsize))
        h = self.max_reward_history_length
        self.reward_history = np.concatenate([self.reward_history, FLOAT32_MIN * np.ones((1, h), dt)], axis=0)
        self.weight_history = np.concatenate([self.weight_history, initial_weight * np.ones((1, h), dt)], axis=0)
        self.adding_thresholds = np.append(self.adding_thresholds, np.asarray(adding_threshold, dt).reshape(-1))
        self.initial_entropies = np.append(self.initial_entropies, np.asarray(initial_entropy, dt).reshape(-1))

    def remove_component(self, idx):
        """:129-148."""
        self.model.remove_component(idx)
        for name in ('unique_component_ids', 'l2_regularizers', 'last_log_etas', 'num_received_updates',
                     'stepsizes', 'adding_thresholds', 'initial_entropies'):
            setattr(self, name, np.delete(getattr(self, name), idx, axis=0))
        self.reward_history = np.delete(self.reward_history, idx, axis=0)
        self.weight_history = np.delete(self.weight_history, idx, axis=0)

    def store_rewards(self, rewards):
        """:150-158."""
        self.reward_history = np.concatenate([self.reward_history[:, 1:], np.asarray(rewards)[:, None]], axis=1)

    def update_stepsizes(self, new_stepsizes):
        """:160-168."""
        self.stepsizes = np.asarray(new_stepsizes, self.model.dtype).copy()

    def replace_weights(self, new_log_weights):
        """:170-182."""
        self.model.replace_weights(new_log_weights)
        self.weight_history = np.concatenate([self.weight_history[:, 1:], self.model.weights[:, None]], axis=1)
