"""GMM target (reference: src/gmmvi/experiments/target_distributions/gmm.py:12-237): the mixture log-density and its
gradient come from the same fused density kernel that evaluates the model."""
import numpy as np

from ... import _lib, hip_ops
from ...device import get_context
from .lnpdf import LNPDF


class GMM_LNPDF(LNPDF):
    _family = _lib.GAUSS
    _nu = 0.0

    def __init__(self, target_weights, target_means, target_covs):
        super().__init__(use_log_density_and_grad=True, safe_for_tf_graph=True)
        self.ctx = get_context()
        self.target_weights = np.asarray(target_weights, np.float32)
        self.target_means = np.asarray(target_means, np.float32)
        self.target_covs = np.asarray(target_covs, np.float32)
        self._init_device()

    def _init_device(self):
        ctx = self.ctx
        chols, ok = hip_ops.cholesky(ctx, ctx.asarray(self.target_covs))                 # gmm.py:36 / stm :44
        if not ok.numpy().all():
            raise ValueError("target covariances must be positive definite")
        self._means_dev = ctx.asarray(self.target_means)
        self._chols_dev = chols
        self._packed, _ = hip_ops.pack_components(ctx, self._means_dev, chols, family=self._family, nu=self._nu)
        lw = np.log(self.target_weights.astype(np.float64))
        lw = lw - np.log(np.sum(np.exp(lw - lw.max()))) - lw.max()                       # Categorical(logits=log w)
        self._logw = ctx.asarray(lw.astype(np.float32))

    def get_num_dimensions(self):
        return int(self.target_means.shape[1])

    def _fast_path_target(self):
        """Descriptor for the single-call iteration (optimization/fused.py)."""
        return {"kind": 0, "family": self._family, "nu": float(self._nu), "K": int(self.target_means.shape[0]),
                "packed": self._packed.ptr, "logw": self._logw.ptr}

    def log_density(self, x):
        _, lp, _ = hip_ops.mixture_eval(self.ctx, self._packed, self._logw, self.ctx.asarray(x),
                                        self.get_num_dimensions(), family=self._family, nu=self._nu)
        return lp

    def log_density_and_grad(self, x):
        _, lp, grad = hip_ops.mixture_eval(self.ctx, self._packed, self._logw, self.ctx.asarray(x),
                                           self.get_num_dimensions(), family=self._family, nu=self._nu,
                                           want_grad=True)
        return lp, grad

    def marginal_log_density(self, x, dim):
        """gmm.py:42-61 (host; plots only)."""
        from scipy.special import logsumexp
        x = np.asarray(x.numpy() if hasattr(x, "numpy") else x, np.float64)
        mu = self.target_means[:, dim].astype(np.float64)
        var = self.target_covs[:, dim, dim].astype(np.float64)
        ld = -0.5 * (x[None, :, dim] - mu[:, None]) ** 2 / var[:, None] - 0.5 * np.log(2 * np.pi * var)[:, None]
        lw = np.log(self.target_weights / self.target_weights.sum())
        return logsumexp(ld + lw[:, None], axis=0)

    def can_sample(self):
        return True

    def sample(self, n, rng=None):
        rng = np.random.default_rng() if rng is None else rng
        w = self.target_weights.astype(np.float64); w /= w.sum()
        comp = rng.choice(len(w), size=n, p=w)
        chols = self._chols_dev.numpy()
        eps = rng.standard_normal((n, self.get_num_dimensions()))
        return (self.target_means[comp] + np.einsum('nij,nj->ni', chols[comp], eps)).astype(np.float32)

    def expensive_metrics(self, model, samples) -> dict:
        """gmm.py:85-121: number of detected modes (the marginal plots of the reference need matplotlib and are
        produced only when it is importable)."""
        means = model.means.numpy()
        dists = np.min(np.linalg.norm(self.target_means[:, None, :] - means[None, :, :], axis=2), axis=1)
        num_detected = int(np.sum(dists < np.linalg.norm(6.0 * np.ones(model.num_dimensions))))
        print(f"Found {num_detected} components.")
        return {"num_detected_modes": num_detected}


def make_target_parameters(num_dimensions):
    """The parameter law of gmm.py:123-145 (global NumPy RNG, as the reference): host arrays only."""
    num_true_components = 10
    weights = np.ones(num_true_components) / num_true_components
    means = np.empty((num_true_components, num_dimensions))
    covs = np.empty((num_true_components, num_dimensions, num_dimensions))
    for i in range(num_true_components):
        means[i] = 100 * (np.random.random(num_dimensions) - 0.5)
        a = 0.1 * np.random.normal(0, num_dimensions, (num_dimensions, num_dimensions))
        covs[i] = a.T @ a + np.eye(num_dimensions)
    return weights, means, covs


def make_target(num_dimensions):
    """gmm.py:123-145."""
    return GMM_LNPDF(*make_target_parameters(num_dimensions))


def make_target_with_scale_parameters(num_dimensions, num_components, scale):
    """The parameter law of gmm.py:148-162: host arrays only."""
    weights = np.ones(num_components) / num_components
    means = np.empty((num_components, num_dimensions))
    covs = np.empty((num_components, num_dimensions, num_dimensions))
    for i in range(num_components):
        means[i] = 100 * (np.random.random(num_dimensions) - 0.5)
        a = np.random.normal(0, np.sqrt(scale), (num_dimensions, num_dimensions))
        covs[i] = a.T @ a + np.eye(num_dimensions)
    return weights, means, covs


def make_target_with_scale(num_dimensions, num_components, scale):
    """gmm.py:148-162."""
    return GMM_LNPDF(*make_target_with_scale_parameters(num_dimensions, num_components, scale))
