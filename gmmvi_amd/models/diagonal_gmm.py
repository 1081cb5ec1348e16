"""DiagonalGMM on the MI355X (reference: src/gmmvi/models/diagonal_gmm.py:6-59).

``chol_cov`` is the reference's [K, D] array of standard deviations.  Densities, gradients, the background density and
sampling run on dedicated O(D)-per-pair kernels (csrc/diag_sweep.hip: component blocks [mu | 1/sigma | 1/sigma^2 | c]); the
component updates have their own elementwise kernels (csrc/diag.hip).  ``dense_chol`` (the embedded factors diag(sigma)) is
still available for callers of the dense entry points.
"""
import numpy as np

from .. import hip_ops
from ..device import DeviceArray
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
        self._dense_packed = None
        super().__init__(np.log(w).astype(np.float32), means, np.sqrt(covs_host), ctx)          # :21-27
        self.diagonal_covs = True                                                                # :28

    # ---- dense view for the kernels -------------------------------------------------------------------------------
    def _invalidate(self):
        super()._invalidate()
        self._dense = None
        self._dense_packed = None

    @property
    def dense_chol(self):
        if self._dense is None:
            self._dense = hip_ops.diag_embed(self.ctx, self.chol_cov)
        return self._dense

    @property
    def dense_packed(self):
        """Component blocks of the DENSE kernels on the embedded factors (the Stein estimate of the register-path dimensions)."""
        if self._dense_packed is None:
            self._dense_packed, _ = hip_ops.pack_components(self.ctx, self.means, self.dense_chol)
        return self._dense_packed

    @property
    def packed(self):
        """Component blocks of the diagonal kernels [K, diag_packed_stride(D)]."""
        if self._packed is None:
            self._packed = hip_ops.diag_pack(self.ctx, self.means, self.chol_cov)
        return self._packed

    # ---- densities on the diagonal kernels (gmm.py:183-216, 274-300 with diagonal_gmm.py:47-53) ----------------------------
    def log_densities_also_individual(self, samples):
        ld, lp, _ = hip_ops.diag_mixture_eval(self.ctx, self.packed, self.log_weights, self._x(samples),
                                              self.num_dimensions, want_ld=True, want_lp=True)
        return lp, ld

    def log_density(self, samples):
        _, lp, _ = hip_ops.diag_mixture_eval(self.ctx, self.packed, self.log_weights, self._x(samples),
                                             self.num_dimensions, want_lp=True)
        return lp

    def eval_with_background(self, samples, log_background_weights):
        """Background density over the model's own components and the model's log_density_and_grad in one sweep (see
        GMM.eval_with_background); the model part is cached for the next log_density_and_grad(samples)."""
        x = self._x(samples)
        packed = self.packed
        ld, lp, grad, bg = hip_ops.diag_mixture_eval(self.ctx, packed, self.log_weights, x, self.num_dimensions, want_ld=True,
                                                     want_lp=True, want_grad=True, logw2=log_background_weights)
        self._eval_cache = (x.ptr, x.shape, packed, self.log_weights.ptr, (lp, grad, ld))
        return bg

    def log_density_and_grad(self, samples):
        x = self._x(samples)
        c = self._eval_cache
        if c is not None:
            self._eval_cache = None
            if c[0] == x.ptr and c[1] == x.shape and c[2] is self._packed and c[3] == self.log_weights.ptr:
                return c[4]
        ld, lp, grad = hip_ops.diag_mixture_eval(self.ctx, self.packed, self.log_weights, x, self.num_dimensions,
                                                 want_ld=True, want_lp=True, want_grad=True)
        return lp, grad, ld

    def sample_from_components_no_shuffle(self, samples_per_component, seed=None, first_index=0, eps=None,
                                          stream_id=None):
        """gmm.py:361-386 with diagonal_gmm.py:43-45: x = mu + sigma * eps (same Philox counters as the dense kernel)."""
        from .gmm import STREAM_COMPONENT_NORMALS
        stream_id = STREAM_COMPONENT_NORMALS if stream_id is None else stream_id
        n_k = np.asarray(samples_per_component, dtype=np.int64).reshape(-1)
        if n_k.shape[0] != self.num_components:
            raise ValueError("samples_per_component must have one entry per component")
        offsets = np.concatenate([[0], np.cumsum(n_k)]).astype(np.int32)
        n = int(offsets[-1])
        offsets_dev = self.ctx.cached_const(("offsets", offsets.tobytes()), lambda: self.ctx.asarray(offsets, np.int32))
        return hip_ops.diag_sample(self.ctx, self.means, self.chol_cov, offsets_dev, n,
                                   seed=self.seed if seed is None else seed, first_index=first_index, stream_id=stream_id,
                                   eps=None if eps is None else self.ctx.asarray(eps))

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
        ld, _, _ = hip_ops.diag_mixture_eval(self.ctx, self.packed, self.log_weights, self._x(samples), self.num_dimensions,
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
        if not isinstance(initial_mean, DeviceArray):
            initial_mean = self.ctx.asarray(np.asarray(initial_mean, np.float32).reshape(1, d))
        self.means = self._append_rows(self.means, initial_mean.reshape((1, d)))
        self.chol_cov = self._append_rows(self.chol_cov, self.ctx.asarray(np.sqrt(cov)[None]))
        self._invalidate()
        new_lw = self.ctx.asarray(np.array([np.log(np.float64(initial_weight))], np.float32))
        self.log_weights = self._renormalised(self._append_rows(self.log_weights, new_lw))
