"""Oracle restatement of the three in-scope target log-densities and their gradients.  TEST INFRASTRUCTURE.

Reference: experiments/target_distributions/gmm.py:28-40,123-162, student_t_mixture.py:34-68,138-169,
planar_robot.py:29-66 (the arithmetic itself lives in tensorflow-probability==0.20.1:
``MixtureSameFamily``, ``MultivariateNormalTriL``, ``MultivariateStudentTLinearOperator``,
``MultivariateNormalDiag`` -- not vendored; restated from their published densities and checked
against scipy.stats in tests/test_oracle_targets.py).  Gradients (reference: GradientTape,
gmmvi_modules/sample_selector.py:74-77) are analytic and checked by finite differences.
"""
import numpy as np
from scipy.linalg import solve_triangular
from scipy.special import logsumexp, gammaln


class GmmTarget:
    """GMM_LNPDF (gmm.py:12-40): log p(x) = LSE_c(log pi_c + log N(x; m_c, S_c))."""
    family = "gauss"

    def __init__(self, weights, means, covs, dtype=np.float64):
        self.dtype = dtype
        self.weights = np.asarray(weights, dtype)
        self.means = np.asarray(means, dtype)
        self.covs = np.asarray(covs, dtype)
        self.chols = np.stack([np.linalg.cholesky(c) for c in self.covs])        # gmm.py:36 scale_tril
        self.log_weights = np.log(self.weights) - logsumexp(np.log(self.weights))  # Categorical(logits=log w)

    def get_num_dimensions(self):
        return self.means.shape[1]

    def _components(self, x):
        x = np.asarray(x, self.dtype)
        c, d = self.means.shape
        ld = np.empty((c, x.shape[0]), self.dtype)
        ys = np.empty((c, x.shape[0], d), self.dtype)
        for i in range(c):
            z = solve_triangular(self.chols[i], (x - self.means[i]).T, lower=True)
            ys[i] = solve_triangular(self.chols[i], z, lower=True, trans='T').T
            ld[i] = -0.5 * np.sum(z * z, axis=0) - np.sum(np.log(np.diag(self.chols[i]))) - 0.5 * d * np.log(2 * np.pi)
        return ld, ys

    def log_density(self, x):
        ld, _ = self._components(x)
        return logsumexp(ld + self.log_weights[:, None], axis=0)

    def log_density_and_grad(self, x):
        ld, ys = self._components(x)
        lp = logsumexp(ld + self.log_weights[:, None], axis=0)
        resp = np.exp(ld + self.log_weights[:, None] - lp[None])
        return lp, -np.einsum('cn,cnd->nd', resp, ys)


class StudentTMixtureTarget:
    """StudentTMixture_LNPDF (student_t_mixture.py:12-68): mixture of multivariate Student-t, df = alpha,
    location m_c, *scale* operator T_c = chol(target_covs_c) (student_t_mixture.py:40-44):
      log t(x) = lgamma((v+D)/2) - lgamma(v/2) - D/2 log(v pi) - sum log diag T - (v+D)/2 log1p(|T^-1 (x-m)|^2 / v)."""
    family = "student_t"

    def __init__(self, weights, means, covs, alpha=2, dtype=np.float64):
        self.dtype = dtype
        self.alpha = float(alpha)
        self.weights = np.asarray(weights, dtype)
        self.means = np.asarray(means, dtype)
        self.covs = np.asarray(covs, dtype)
        self.chols = np.stack([np.linalg.cholesky(c) for c in self.covs])
        self.log_weights = np.log(self.weights) - logsumexp(np.log(self.weights))

    def get_num_dimensions(self):
        return self.means.shape[1]

    def _components(self, x):
        x = np.asarray(x, self.dtype)
        c, d = self.means.shape
        v = self.alpha
        ld = np.empty((c, x.shape[0]), self.dtype)
        gs = np.empty((c, x.shape[0], d), self.dtype)
        for i in range(c):
            z = solve_triangular(self.chols[i], (x - self.means[i]).T, lower=True)
            q = np.sum(z * z, axis=0)
            y = solve_triangular(self.chols[i], z, lower=True, trans='T').T
            ld[i] = (gammaln(0.5 * (v + d)) - gammaln(0.5 * v) - 0.5 * d * np.log(v * np.pi)
                     - np.sum(np.log(np.diag(self.chols[i]))) - 0.5 * (v + d) * np.log1p(q / v))
            gs[i] = -((v + d) / (v + q))[:, None] * y
        return ld, gs

    def log_density(self, x):
        ld, _ = self._components(x)
        return logsumexp(ld + self.log_weights[:, None], axis=0)

    def log_density_and_grad(self, x):
        ld, gs = self._components(x)
        lp = logsumexp(ld + self.log_weights[:, None], axis=0)
        resp = np.exp(ld + self.log_weights[:, None] - lp[None])
        return lp, np.einsum('cn,cnd->nd', resp, gs)


class PlanarRobotTarget:
    """PlanarRobot (planar_robot.py:11-66): diag-Gaussian prior on joint angles plus the max over goal
    Gaussians of the end-effector position (unit link lengths)."""
    family = "planar"

    def __init__(self, num_links=10, num_goals=4, prior_std=2e-1, likelihood_std=1e-2, dtype=np.float64):
        self.dtype = dtype
        self.num_links = num_links
        stds = prior_std * np.ones(num_links)
        stds[0] = 1.0                                                            # planar_robot.py:32-33
        self.prior_stds = stds.astype(dtype)
        self.likelihood_std = float(likelihood_std)
        if num_goals == 1:
            self.goals = np.array([[7., 0.]], dtype)
        elif num_goals == 4:
            self.goals = np.array([[7., 0.], [-7., 0.], [0., 7.], [0., -7.]], dtype)   # :40
        else:
            raise ValueError

    def get_num_dimensions(self):
        return self.num_links

    def forward_kinematics(self, theta):
        """planar_robot.py:58-64."""
        c = np.cumsum(np.asarray(theta, self.dtype), axis=1)
        return np.stack([np.cos(c).sum(axis=1), np.sin(c).sum(axis=1)], axis=1)

    def log_density_and_grad(self, theta):
        theta = np.asarray(theta, self.dtype)
        d = self.num_links
        s = self.prior_stds
        prior = -0.5 * np.sum((theta / s) ** 2, axis=1) - np.sum(np.log(s)) - 0.5 * d * np.log(2 * np.pi)
        gprior = -theta / s ** 2
        c = np.cumsum(theta, axis=1)
        sinc, cosc = np.sin(c), np.cos(c)
        px, py = cosc.sum(axis=1), sinc.sum(axis=1)
        # d px / d theta_j = -sum_{i>=j} sin c_i ; d py / d theta_j = sum_{i>=j} cos c_i
        dpx = -np.cumsum(sinc[:, ::-1], axis=1)[:, ::-1]
        dpy = np.cumsum(cosc[:, ::-1], axis=1)[:, ::-1]
        ls = self.likelihood_std
        ll = np.stack([-0.5 * ((px - g[0]) ** 2 + (py - g[1]) ** 2) / ls ** 2 - 2 * np.log(ls) - np.log(2 * np.pi)
                       for g in self.goals], axis=0)                              # [G, N]
        best = np.argmax(ll, axis=0)                                             # planar_robot.py:52-56 reduce_max
        g = self.goals[best]
        glik = -((px - g[:, 0])[:, None] * dpx + (py - g[:, 1])[:, None] * dpy) / ls ** 2
        return prior + ll[best, np.arange(theta.shape[0])], gprior + glik

    def log_density(self, theta):
        return self.log_density_and_grad(theta)[0]


# ---- constructors (laws of the reference, NumPy Generator instead of the global/TF RNG) ------------

def make_gmm_target(num_dimensions, rng, num_components=10, dtype=np.float64):
    """gmm.py:123-145: means 100*(U-0.5), cov = A^T A + I, A = 0.1 * N(0, std=D)^{DxD}."""
    w = np.ones(num_components) / num_components
    means = 100.0 * (rng.random((num_components, num_dimensions)) - 0.5)
    covs = np.empty((num_components, num_dimensions, num_dimensions))
    for i in range(num_components):
        a = 0.1 * rng.normal(0.0, num_dimensions, (num_dimensions, num_dimensions))
        covs[i] = a.T @ a + np.eye(num_dimensions)
    return GmmTarget(w, means, covs, dtype)


def make_gmm_target_with_scale(num_dimensions, num_components, scale, rng, dtype=np.float64):
    """gmm.py:148-162: A ~ N(0, sqrt(scale))."""
    w = np.ones(num_components) / num_components
    means = 100.0 * (rng.random((num_components, num_dimensions)) - 0.5)
    covs = np.empty((num_components, num_dimensions, num_dimensions))
    for i in range(num_components):
        a = rng.normal(0.0, np.sqrt(scale), (num_dimensions, num_dimensions))
        covs[i] = a.T @ a + np.eye(num_dimensions)
    return GmmTarget(w, means, covs, dtype)


def make_stm_target(num_dimensions, rng, harder_setting=False, dtype=np.float64):
    """student_t_mixture.py:153-169: means U(-s, s), cov = inv(A^T A + I), A = 0.1*D*N(0,1)^{DxD}, df 2."""
    s, num_components = (25, 20) if harder_setting else (20, 10)
    w = np.ones(num_components) / num_components
    means = rng.random((num_components, num_dimensions)) * (2 * s) - s
    covs = np.empty((num_components, num_dimensions, num_dimensions))
    for i in range(num_components):
        a = 0.1 * num_dimensions * rng.normal(0.0, 1.0, (num_dimensions, num_dimensions))
        covs[i] = np.linalg.inv(a.T @ a + np.eye(num_dimensions))
    return StudentTMixtureTarget(w, means, covs, alpha=2, dtype=dtype)
