"""Host-side mirror of the reference's GMM base class (reference: src/gmmvi/models/gmm.py:5-418).

Parameters live in HBM as ``DeviceArray``s (log_weights [K], means [K,D], chol_cov [K,D,D] -- [K,D] for DiagonalGMM); every density /
sampling method launches the HIP kernels of libgmmvi_hip.so.  Method names, argument order and return tuples are
the reference's; returned arrays are device resident and offer ``.numpy()``.
"""
import numpy as np

from .. import _lib, hip_ops
from ..device import DeviceArray, get_context

STREAM_COMPONENT_NORMALS = 0   # oracle/philox.py documents the stream layout
STREAM_CATEGORICAL = 1
STREAM_MIXTURE_NORMALS = 2


class GMM:
    def __init__(self, log_weights, means, chol_covs, ctx=None):
        self.ctx = ctx if ctx is not None else get_context()
        self.means = self.ctx.asarray(means)
        self.chol_cov = self.ctx.asarray(chol_covs)
        self.diagonal_covs = self.chol_cov.ndim == 2
        self.num_dimensions = int(self.means.shape[1])
        self._const_log_det = 0.5 * self.num_dimensions * np.log(2 * np.pi)
        self.log_weights = self.ctx.asarray(log_weights)
        self._packed = None
        self._eval_cache = None       # (samples ptr, shape, packed) -> (lp, grad, ld) of a fused evaluation
        self.seed = 0                 # Philox key for sample(); the runner sets it from the config seed
        self._sample_counter = 0      # index into the mixture-sampling Philox streams
        self.replace_weights(self.log_weights)                                   # gmm.py:34

    # ---- packed parameter blocks for the kernels (rebuilt lazily after every component change) -----------
    def _invalidate(self):
        self._packed = None
        self._eval_cache = None

    @property
    def packed(self):
        if self._packed is None:
            self._packed, _ = hip_ops.pack_components(self.ctx, self.means, self.chol_cov)
        return self._packed

    def _kernel_chol(self):
        """Dense [K,D,D] factors the sampling kernel reads (DiagonalGMM: the embedded diag(sigma))."""
        return self.chol_cov

    # ---- properties -------------------------------------------------------------------------------------
    @property
    def num_components(self):
        return int(self.log_weights.shape[0])

    @property
    def weights(self):
        """gmm.py:165-171 (host array: used for metrics only)."""
        return np.exp(self.log_weights.numpy())

    @property
    def covs(self):
        raise NotImplementedError

    # ---- densities (gmm.py:183-216, 274-300) ----------------------------------------------------------------
    def _x(self, samples):
        return self.ctx.asarray(samples)

    def component_log_densities(self, samples):
        raise NotImplementedError

    def log_densities_also_individual(self, samples):
        ld, lp, _ = hip_ops.mixture_eval(self.ctx, self.packed, self.log_weights, self._x(samples),
                                         self.num_dimensions, want_ld=True, want_lp=True)
        return lp, ld

    def log_density(self, samples):
        _, lp, _ = hip_ops.mixture_eval(self.ctx, self.packed, self.log_weights, self._x(samples),
                                        self.num_dimensions, want_lp=True)
        return lp

    def density(self, samples):
        return np.exp(self.log_density(samples).numpy())

    def eval_with_background(self, samples, log_background_weights):
        """Background density log sum_k c_k N(x; mu_k, Sigma_k) (sample_db.py:221-227) AND the model's
        log_density_and_grad of the same samples in one sweep over the (sample, component) pairs.  The model part is
        cached and handed out by the next log_density_and_grad(samples) call (gmmvi_modules/ng_estimator.py:246)."""
        x = self._x(samples)
        packed = self.packed
        ld, lp, grad, bg = hip_ops.mixture_eval_dual(self.ctx, packed, self.log_weights, log_background_weights, x,
                                                     self.num_dimensions)
        self._eval_cache = (x.ptr, x.shape, packed, self.log_weights.ptr, (lp, grad, ld))
        return bg

    def log_density_and_grad(self, samples):
        """-> (log q [N], grad_x log q [N,D], component log densities [K,N])."""
        c = self._eval_cache
        if c is not None:
            x = self._x(samples)
            if c[0] == x.ptr and c[1] == x.shape and c[2] is self._packed and c[3] == self.log_weights.ptr:
                self._eval_cache = None
                return c[4]
        ld, lp, grad = hip_ops.mixture_eval(self.ctx, self.packed, self.log_weights, self._x(samples),
                                            self.num_dimensions, want_ld=True, want_lp=True, want_grad=True)
        return lp, grad, ld

    def marginal_log_density(self, samples, dimension):
        """gmm.py:218-234 (plots only: evaluated on the host from downloaded parameters)."""
        from scipy.special import logsumexp
        ld = self.component_marginal_log_densities(samples, dimension)
        return logsumexp(ld + self.log_weights.numpy()[:, None], axis=0)

    # ---- entropies (gmm.py:249-272) ---------------------------------------------------------------------------
    def gaussian_entropy(self, chol):
        raise NotImplementedError

    def component_entropies(self):
        chols = self.chol_cov.numpy()
        return np.array([self.gaussian_entropy(c) for c in chols], dtype=np.float32)

    def get_average_entropy(self):
        return float(np.sum(self.weights * self.component_entropies()))

    # ---- sampling -----------------------------------------------------------------------------------------------
    def sample_from_components_no_shuffle(self, samples_per_component, seed=None, first_index=0, eps=None,
                                          stream_id=STREAM_COMPONENT_NORMALS):
        """gmm.py:361-386: component-ordered samples and their mapping.  ``samples_per_component`` is a host
        int array [K]; normals come from the Philox stream (seed, first_index + n) or from ``eps`` [N,D]."""
        n_k = np.asarray(samples_per_component, dtype=np.int64).reshape(-1)
        if n_k.shape[0] != self.num_components:
            raise ValueError("samples_per_component must have one entry per component")
        offsets = np.concatenate([[0], np.cumsum(n_k)]).astype(np.int32)
        n = int(offsets[-1])
        offsets_dev = self.ctx.cached_const(("offsets", offsets.tobytes()),
                                            lambda: self.ctx.asarray(offsets, np.int32))
        return hip_ops.sample_components(self.ctx, self.means, self._kernel_chol(), offsets_dev, n,
                                         seed=self.seed if seed is None else seed, first_index=first_index,
                                         stream_id=stream_id,
                                         eps=None if eps is None else self.ctx.asarray(eps))

    def sample_from_component(self, index, num_samples):
        n_k = np.zeros(self.num_components, np.int64)
        n_k[index] = num_samples
        first = self._sample_counter
        self._sample_counter += int(num_samples)
        return self.sample_from_components_no_shuffle(n_k, first_index=first, stream_id=STREAM_MIXTURE_NORMALS)[0]

    def sample_categorical(self, num_samples):
        """gmm.py:124-137: first cumulative-weight threshold exceeding a uniform draw."""
        u = hip_ops.philox_uniforms(self.ctx, self.seed, self._sample_counter, num_samples, STREAM_CATEGORICAL).numpy()
        thresholds = np.cumsum(np.exp(self.log_weights.numpy().astype(np.float64)))
        return np.argmax(u.astype(np.float64)[:, None] < thresholds[None, :], axis=1).astype(np.int32)

    def sample(self, num_samples):
        """gmm.py:139-163: (samples grouped by component, component index of every *draw*)."""
        num_samples = int(num_samples)
        comp = self.sample_categorical(num_samples)
        counts = np.bincount(comp, minlength=self.num_components)
        first = self._sample_counter
        self._sample_counter += num_samples
        x, _ = self.sample_from_components_no_shuffle(counts, first_index=first, stream_id=STREAM_MIXTURE_NORMALS)
        return x, comp

    # ---- mutation -----------------------------------------------------------------------------------------------
    def replace_weights(self, new_log_weights):
        """gmm.py:173-181: normalises by log-sum-exp."""
        from scipy.special import logsumexp
        lw = np.asarray(new_log_weights.numpy() if isinstance(new_log_weights, DeviceArray) else new_log_weights,
                        dtype=np.float64)
        self.log_weights = self.ctx.asarray((lw - logsumexp(lw)).astype(np.float32))

    def replace_components(self, new_means, new_chols):
        """gmm.py:401-418."""
        self.means = self.ctx.asarray(new_means)
        self.chol_cov = self.ctx.asarray(new_chols)
        self._invalidate()

    def _renormalised(self, log_weights_dev):
        """gmm.py:173-181 on the device (fp64 log-sum-exp): no read-back of the weights."""
        out = self.ctx.empty(log_weights_dev.shape)
        self.ctx.check(self.ctx.lib.gmmvi_normalize_logw(self.ctx.handle, log_weights_dev.ptr, int(log_weights_dev.shape[0]), out.ptr))
        return out

    def _append_rows(self, arr, new_row, pairs=None):
        """[K, ...] device array with one more row (device-side copy, nothing read back).  ``pairs``: the two copies are queued
        there for ONE hip_ops.copy_batch launch of the caller (adding a component appends to seven arrays)."""
        inner = tuple(arr.shape[1:])
        new_row = self.ctx.asarray(new_row) if not isinstance(new_row, DeviceArray) else new_row
        if pairs is None:
            return hip_ops.concat(self.ctx, [arr, new_row]).reshape((arr.shape[0] + 1,) + inner)
        k = arr.shape[0]
        out = self.ctx.empty((k + 1,) + inner, arr.dtype)
        pairs.append((out.rows(0, k).reshape(-1), arr.reshape(-1)))
        pairs.append((out.rows(k, k + 1).reshape(-1), new_row.reshape(-1)))
        return out

    def remove_component(self, idx):
        """gmm.py:388-398 (device-side gathers: nothing is read back)."""
        idx = int(idx)
        keep = np.delete(np.arange(self.num_components, dtype=np.int32), idx)
        keep_dev = self.ctx.asarray(keep, np.int32)
        self.log_weights = self._renormalised(hip_ops.gather_rows(self.ctx, self.log_weights, keep_dev))
        self.means = hip_ops.gather_rows(self.ctx, self.means, keep_dev)
        self.chol_cov = hip_ops.gather_rows(self.ctx, self.chol_cov, keep_dev)
        self._invalidate()
