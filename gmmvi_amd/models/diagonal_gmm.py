"""DiagonalGMM on the MI355X (reference: src/gmmvi/models/diagonal_gmm.py:6-59).

``chol_cov`` is the reference's [K, D] array of standard deviations.  The density / gradient / sampling kernels are
the dense ones, run on the embedded factors L_k = diag(sigma_k) (``hip_ops.diag_embed``; the zeros off the diagonal
contribute exact zeros); the component updates have their own elementwise kernels (csrc/diag.hip).
"""
import numpy as np

from .. import hip_ops
from .gmm import GMM


class DiagonalGMM(GMM):
    """Parameters (diagonal_gmm.py:19): weights [K], means [K,D], covs [K,D] (diagonal covariance entries)."""

    def __init__(self, weights, means, covs, ctx=None):
        from ..device import get_context
        ctx = ctx if ctx is not None else get_context()
        means = ctx.asarray(means)
        covs_host = np.asarray(covs.numpy() if hasattr(covs, "numpy") else covs, np.float32)
        if covs_host.ndim != 2 or covs_host.shape != tuple(means.shape):
            raise ValueError(f"covs must be [K,D]; got {covs_host.shape} for means {means.shape}")
        if not np.all(covs_host > 0):
            raise ValueError("initial covariance entries must be positive")
        w = np.asarray(weights.numpy() if hasattr(weights, "numpy") else weights, dtype=np.float64)
        self._dense = None
        super().__init__(np.log(w).astype(np.float32), means, np.sqrt(covs_host), ctx)          # :21-27
        self.diagonal_covs = True                                                                # :28

    # ---- dense view for the kernels -------------------------------------------------------------------------------
    def _invalidate(self):
        super()._invalidate()
        self._dense = None

    @property
    def dense_chol(self):
        if self._dense is None:
            self._dense = hip_ops.diag_embed(self.ctx, self.chol_cov)
        return self._dense

    @property
    def packed(self):
        if self._packed is None:
            self._packed, _ = hip_ops.pack_components(self.ctx, self.means, self.dense_chol)
        return self._packed

    def _kernel_chol(self):
        return self.dense_chol

    # ---- reference API ------------------------------------------------------------------------------------------------
    @property
    def covs(self):
        """:36-38 (host array)."""
        return np.square(self.chol_cov.numpy())

    def gaussian_entropy(self, chol):
        """:40-41."""
        return 0.5 * self.num_dimensions * (np.log(2 * np.pi) + 1) + np.sum(np.log(np.asarray(chol)))

    def component_log_densities(self, samples):
        """:47-53 -> [K, N]."""
        ld, _, _ = hip_ops.mixture_eval(self.ctx, self.packed, self.log_weights, self._x(samples), self.num_dimensions,
                                        want_ld=True, want_lp=False)
        return ld

    def component_log_density(self, index, samples):
        return self.component_log_densities(samples).rows(int(index), int(index) + 1).reshape(-1)

    def component_marginal_log_densities(self, samples, dim):
        x = np.asarray(samples.numpy() if hasattr(samples, "numpy") else samples)
        var = self.covs[:, dim]
        diffs = x[None, :, dim] - self.means.numpy()[:, dim, None]
        return -0.5 * diffs * diffs / var[:, None] - 0.5 * np.log(var)[:, None] - 0.5 * np.log(2 * np.pi)

    def add_component(self, initial_weight, initial_mean, initial_cov):
        """:55-59."""
        d = self.num_dimensions
        cov = np.asarray(initial_cov, np.float32).reshape(d)
        if not np.all(cov > 0):
            raise ValueError("add_component: covariance entries must be positive")
        self.means = self.ctx.asarray(np.concatenate([self.means.numpy(),
                                                      np.asarray(initial_mean, np.float32).reshape(1, d)]))
        self.chol_cov = self.ctx.asarray(np.concatenate([self.chol_cov.numpy(), np.sqrt(cov)[None]]))
        self._invalidate()
        self.replace_weights(np.concatenate([self.log_weights.numpy().astype(np.float64),
                                             [np.log(np.float64(initial_weight))]]))
