"""Mixture of multivariate Student-t target (reference: src/gmmvi/experiments/target_distributions/
student_t_mixture.py:12-199): same fused kernel as the GMM target with the Student-t family switch."""
import numpy as np

from ... import _lib
from .gmm import GMM_LNPDF


class StudentTMixture_LNPDF(GMM_LNPDF):
    _family = _lib.STUDENT_T

    def __init__(self, target_weights, target_means, target_covs, alpha=2):
        self.alpha = alpha
        self._nu = float(alpha)
        super().__init__(target_weights, target_means, target_covs)

    def marginal_log_density(self, x, dim):
        """student_t_mixture.py:46-64 (host; plots only)."""
        from scipy.special import logsumexp
        from scipy import stats
        x = np.asarray(x.numpy() if hasattr(x, "numpy") else x, np.float64)
        ld = np.stack([stats.t(df=self.alpha, loc=self.target_means[c, dim],
                               scale=np.sqrt(self.target_covs[c, dim, dim])).logpdf(x[:, dim])
                       for c in range(len(self.target_weights))])
        lw = np.log(self.target_weights / self.target_weights.sum())
        return logsumexp(ld + lw[:, None], axis=0)

    def sample(self, n, rng=None):
        rng = np.random.default_rng() if rng is None else rng
        w = self.target_weights.astype(np.float64); w /= w.sum()
        comp = rng.choice(len(w), size=n, p=w)
        chols = self._chols_dev.numpy()
        d = self.get_num_dimensions()
        eps = rng.standard_normal((n, d))
        g = rng.chisquare(self.alpha, size=n) / self.alpha
        return (self.target_means[comp] + np.einsum('nij,nj->ni', chols[comp], eps) / np.sqrt(g)[:, None]).astype(np.float32)


def make_target_parameters(num_dimensions, harder_setting):
    """The parameter law of student_t_mixture.py:153-169 (global NumPy RNG, as the reference): host arrays only."""
    s, num_components = (25, 20) if harder_setting else (20, 10)
    weights = np.ones(num_components) / num_components
    means = np.empty((num_components, num_dimensions))
    covs = np.empty((num_components, num_dimensions, num_dimensions))
    for i in range(num_components):
        means[i] = np.random.uniform(0, 1, num_dimensions) * (2 * s) - s
        a = 0.1 * num_dimensions * np.random.normal(0, 1, (num_dimensions, num_dimensions))
        covs[i] = np.linalg.inv(a.T @ a + np.eye(num_dimensions))
    return weights, means, covs


def make_target(num_dimensions, harder_setting, use_matlab_target=False):
    """student_t_mixture.py:138-194 (the MATLAB known-answer data of the reference is not shipped, :171-193)."""
    if use_matlab_target:
        raise ValueError("the MATLAB target data is not shipped with the reference (student_t_mixture.py:171-193)")
    return StudentTMixture_LNPDF(*make_target_parameters(num_dimensions, harder_setting))
