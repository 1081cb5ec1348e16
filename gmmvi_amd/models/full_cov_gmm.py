"""FullCovGMM on the MI355X (reference: src/gmmvi/models/full_cov_gmm.py:6-68)."""
import numpy as np

from .. import hip_ops
from ..device import DeviceArray
from .gmm import GMM


class FullCovGMM(GMM):
    """A Gaussian mixture model with full covariance matrices.

    Parameters (as in the reference, full_cov_gmm.py:19): weights [K], means [K,D], covs [K,D,D]; array-likes,
    NumPy arrays or DeviceArrays.
    """

    def __init__(self, weights, means, covs, ctx=None):
        from ..device import get_context
        ctx = ctx if ctx is not None else get_context()
        means = ctx.asarray(means)
        covs = ctx.asarray(covs)
        if covs.ndim != 3 or covs.shape[0] != means.shape[0] or covs.shape[1] != means.shape[1]:
            raise ValueError(f"covs must be [K,D,D]; got {covs.shape} for means {means.shape}")
        chols, ok = hip_ops.cholesky(ctx, covs)                                   # full_cov_gmm.py:23
        if not ok.numpy().all():
            raise ValueError("initial covariance matrices must be positive definite")
        w = np.asarray(weights.numpy() if hasattr(weights, "numpy") else weights, dtype=np.float64)
        super().__init__(np.log(w).astype(np.float32), means, chols, ctx)
        self.diagonal_covs = False

    @property
    def covs(self):
        """full_cov_gmm.py:29-31 (host array; metrics / dumps only)."""
        l = self.chol_cov.numpy()
        return l @ np.transpose(l, (0, 2, 1))

    def gaussian_entropy(self, chol):
        """full_cov_gmm.py:33-34."""
        chol = np.asarray(chol)
        return 0.5 * self.num_dimensions * (np.log(2 * np.pi) + 1) + np.sum(np.log(np.diag(chol)))

    def component_entropies(self):
        """gmm.py:249-261 for all components at once (one row-wise reduction instead of a Python loop over K: 0.8 ms per call at
        K = 170, called by every add of the adaptive configurations)."""
        diag = np.ascontiguousarray(np.diagonal(self.chol_cov.numpy(), axis1=1, axis2=2))
        return (0.5 * self.num_dimensions * (np.log(2 * np.pi) + 1) + np.sum(np.log(diag), axis=1)).astype(np.float32)

    def component_log_densities(self, samples):
        """full_cov_gmm.py:56-62 -> [K, N]."""
        ld, _, _ = hip_ops.mixture_eval(self.ctx, self.packed, self.log_weights, self._x(samples), self.num_dimensions,
                                        want_ld=True, want_lp=False)
        return ld

    def component_log_density(self, index, samples):
        """full_cov_gmm.py:41-47."""
        return self.component_log_densities(samples).rows(int(index), int(index) + 1).reshape(-1)

    def component_marginal_log_densities(self, samples, dim):
        """full_cov_gmm.py:49-54 (host; used by the targets' marginal plots)."""
        x = np.asarray(samples.numpy() if hasattr(samples, "numpy") else samples)
        var = self.covs[:, dim, dim]
        diffs = x[None, :, dim] - self.means.numpy()[:, dim, None]
        return -0.5 * diffs * diffs / var[:, None] - 0.5 * np.log(var)[:, None] - 0.5 * np.log(2 * np.pi)

    def add_component(self, initial_weight, initial_mean, initial_cov, pairs=None):
        """full_cov_gmm.py:64-68.  Device-side appends (nothing is read back); ``initial_mean`` may be a DeviceArray [1, D].
        ``pairs`` (GmmWrapper.add_component): the appends are queued there and the caller launches them together; the log
        weights are then left UNnormalised for the caller to finish (``_renormalised``) behind that launch."""
        d = self.num_dimensions
        cov = np.asarray(initial_cov, np.float32).reshape(d, d)
        if np.count_nonzero(cov - np.diag(np.diagonal(cov))) == 0:
            # a diagonal covariance (what the add heuristic passes, component_adaptation.py:220-223): its factor is the square
            # root of the diagonal -- exactly what the Cholesky kernel returns, without the launch and the status read-back
            if not np.all(np.diagonal(cov) > 0):
                raise ValueError("add_component: covariance is not positive definite")
            chol = self.ctx.asarray(np.diag(np.sqrt(np.diagonal(cov))).reshape(1, d, d))
        else:
            chol, ok = hip_ops.cholesky(self.ctx, self.ctx.asarray(cov.reshape(1, d, d)))
            if not ok.numpy().all():
                raise ValueError("add_component: covariance is not positive definite")
        if not isinstance(initial_mean, DeviceArray):
            initial_mean = self.ctx.asarray(np.asarray(initial_mean, np.float32).reshape(1, d))
        self.means = self._append_rows(self.means, initial_mean.reshape((1, d)), pairs)
        self.chol_cov = self._append_rows(self.chol_cov, chol, pairs)
        self._invalidate()
        new_lw = self.ctx.asarray(np.array([np.log(np.float64(initial_weight))], np.float32))
        appended = self._append_rows(self.log_weights, new_lw, pairs)
        self.log_weights = appended if pairs is not None else self._renormalised(appended)
