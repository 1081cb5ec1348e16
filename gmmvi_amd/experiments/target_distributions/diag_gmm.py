"""Diagonal-covariance GMM target (reference: src/gmmvi/experiments/target_distributions/diag_gmm.py:9-45): the
target IS a DiagonalGMM; its log-density (and, unlike the reference's autodiff, its analytic gradient) comes from the
fused density kernel on the embedded factors."""
import numpy as np

from ...models.diagonal_gmm import DiagonalGMM
from .lnpdf import LNPDF


class DIAGGMM_LNPDF(LNPDF):
    def __init__(self, target_weights, target_means, target_covs):
        super().__init__(use_log_density_and_grad=True, safe_for_tf_graph=True)
        self.target_weights = np.asarray(target_weights, np.float32)
        self.target_means = np.asarray(target_means, np.float32)                 # :15-16
        self.target_covs = np.asarray(target_covs, np.float32)
        self.gmm = DiagonalGMM(self.target_weights, self.target_means, self.target_covs)   # :17

    def log_density(self, x):
        return self.gmm.log_density(x)                                             # :19-21

    def log_density_and_grad(self, x):
        lp, grad, _ = self.gmm.log_density_and_grad(x)
        return lp, grad

    def get_num_dimensions(self):
        return int(self.target_means.shape[1])

    def can_sample(self):
        return True

    def sample(self, n):
        return self.gmm.sample(n)                                                  # :29-30


def make_target(num_dimensions):
    """:33-45: 10 components, means U(-50, 50)^D, covariance diagonals U(0, 10)^D (global NumPy RNG as upstream)."""
    num_true_components = 10
    weights = np.ones(num_true_components) / num_true_components
    means = np.empty((num_true_components, num_dimensions))
    covs = np.empty((num_true_components, num_dimensions))
    for i in range(num_true_components):
        means[i] = 100 * (np.random.random(num_dimensions) - 0.5)
        covs[i] = 10 * np.random.random(num_dimensions)
    return DIAGGMM_LNPDF(weights.astype(np.float32), means.astype(np.float32), covs.astype(np.float32))
