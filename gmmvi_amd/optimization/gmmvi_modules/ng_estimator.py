"""Natural-gradient estimators (reference: src/gmmvi/optimization/gmmvi_modules/ng_estimator.py:10-376)."""
import numpy as np

from ... import hip_ops


class NgEstimator:
    def __init__(self, temperature, model, requires_gradient: bool, only_use_own_samples: bool,
                 use_self_normalized_importance_weights: bool):
        self._model = model
        self._temperature = temperature
        self._requires_gradients = requires_gradient
        self._only_use_own_samples = only_use_own_samples
        self._use_self_normalized_importance_weights = use_self_normalized_importance_weights

    @staticmethod
    def build_from_config(config, temperature, gmm_wrapper):
        """ng_estimator.py:47-65."""
        if config["ng_estimator_type"] == "Stein":
            return SteinNgEstimator(temperature=temperature, model=gmm_wrapper, **config['ng_estimator_config'])
        elif config["ng_estimator_type"] == "MORE":
            return MoreNgEstimator(temperature=temperature, model=gmm_wrapper, **config['ng_estimator_config'])
        raise ValueError(f"config['ng_estimator_type'] is '{config['ng_estimator_type']}' which is an unknown type")

    @property
    def requires_gradients(self) -> bool:
        return self._requires_gradients

    def get_expected_hessian_and_grad(self, samples, mapping, background_densities, target_lnpdfs,
                                      target_lnpdfs_grads):
        raise NotImplementedError


class SteinNgEstimator(NgEstimator):
    """ng_estimator.py:123-263: one fused density+gradient pass (K1-K3) and one MFMA contraction over the samples
    per component (K6-K8) instead of the reference's K-step loop with [N,D,D] temporaries."""

    def __init__(self, temperature, model, only_use_own_samples: bool, use_self_normalized_importance_weights: bool):
        super().__init__(temperature, model, True, only_use_own_samples, use_self_normalized_importance_weights)
        self.last_model_densities = None

    def get_expected_hessian_and_grad(self, samples, mapping, background_densities, target_lnpdfs,
                                      target_lnpdfs_grads):
        """-> (expected_hessian_neg [K,D,D], expected_gradient_neg [K,D])."""
        m = self._model
        ctx = m.ctx
        x = ctx.asarray(samples)
        bg = ctx.asarray(background_densities)
        tgrad = ctx.asarray(target_lnpdfs_grads)
        k, d = m.num_components, m.num_dimensions
        model_densities, model_grads, ld = m.log_density_and_grad(x)                     # ng_estimator.py:246
        self.last_model_densities = model_densities
        map_dev, map_offset = None, 0
        if self._only_use_own_samples:
            map_dev = ctx.asarray(mapping, np.int32)
            # relative_mapping = mapping - max(mapping) + K - 1 (:244); the newest sample belongs to the newest
            # DB component, whose index the host already knows
            host = getattr(m, "_mapping_max_hint", None)
            mx = int(host) if host is not None else int(np.asarray(map_dev.numpy()).max())
            map_offset = k - 1 - mx
        if m.diagonal_covs:
            # :159-162 / :178-181: h[i] = E_w[g[i] (x[i] - mu[i]) / sigma[i]^2], g = E_w[g].
            from ... import _lib
            if d > _lib.blocked_above():
                # O(D) per pair (csrc/diag_sweep.hip): D = 300, K = 64, N = 2e4: 0.56 ms against 3.7 ms on the embedded factors
                return hip_ops.diag_stein(ctx, m.packed, x, ld, model_grads, bg, tgrad, d, mapping=map_dev,
                                          map_offset=map_offset,
                                          self_normalized=self._use_self_normalized_importance_weights,
                                          own_samples_only=self._only_use_own_samples)
            # register-path dimensions: the matrix-core moment kernel on the embedded factors diag(sigma) is faster than the
            # elementwise kernel's few workgroups (D = 20: 30 against 198 us); the diagonal of its estimate is the estimate
            # (the symmetrisation leaves the diagonal alone)
            h_neg, g_neg = hip_ops.stein(ctx, m.dense_packed, x, ld, model_grads, bg, tgrad, d, mapping=map_dev,
                                         map_offset=map_offset,
                                         self_normalized=self._use_self_normalized_importance_weights,
                                         own_samples_only=self._only_use_own_samples)
            return hip_ops.diag_extract(ctx, h_neg), g_neg
        return hip_ops.stein(ctx, m.packed, x, ld, model_grads, bg, tgrad, d, mapping=map_dev, map_offset=map_offset,
                             self_normalized=self._use_self_normalized_importance_weights,
                             own_samples_only=self._only_use_own_samples)


class MoreNgEstimator(NgEstimator):
    """ng_estimator.py:266-376 (MORE, codename letter "Z"): importance-weighted quadratic ridge regression of the
    rewards on the samples whitened by each component (least_squares.py:126-191), as one f32-MFMA Gram contraction
    and one fp64 Cholesky solve per component (csrc/more.hip): register-resident up to D = 21, tiled above (D <= 63; components of a blocked-path dimension are re-packed for the call)."""

    def __init__(self, temperature, model, only_use_own_samples: bool, initial_l2_regularizer: float,
                 use_self_normalized_importance_weights: bool):
        super().__init__(temperature, model, True, only_use_own_samples,                   # :291 (True, as upstream)
                         use_self_normalized_importance_weights)
        l2 = model.l2_regularizers.numpy()
        if not np.all(l2 == np.float32(initial_l2_regularizer)):                          # :293
            raise ValueError("model.l2_regularizers must equal initial_l2_regularizer")
        if model.diagonal_covs:
            raise ValueError("MoreNgEstimator needs a full-covariance model (the reference's QuadFunc whitening, "
                             "least_squares.py:126-191, has no diagonal branch)")
        from ... import _lib
        if model.num_dimensions >= _lib.MAX_DIM:
            raise ValueError(f"MoreNgEstimator: the HIP kernels support D <= {_lib.MAX_DIM - 1} (DESIGN.md section 7)")
        self.last_model_densities = None

    def get_expected_hessian_and_grad(self, samples, mapping, background_densities, target_lnpdfs,
                                      target_lnpdfs_grads=None):
        """-> (expected_hessian_neg [K,D,D], expected_gradient_neg [K,D]); the gradients are not used (:296-299)."""
        m = self._model
        ctx = m.ctx
        x = ctx.asarray(samples)
        bg = ctx.asarray(background_densities)
        tlp = ctx.asarray(target_lnpdfs)
        k, d = m.num_components, m.num_dimensions
        model_densities, ld = m.log_densities_also_individual(x)                          # :344
        self.last_model_densities = model_densities
        map_dev, map_offset = None, 0
        if self._only_use_own_samples:
            map_dev = ctx.asarray(mapping, np.int32)
            host = getattr(m, "_mapping_max_hint", None)
            mx = int(host) if host is not None else int(np.asarray(map_dev.numpy()).max())
            map_offset = k - 1 - mx                                                        # :342
        return hip_ops.more(ctx, m.packed, m.chol_cov, x, ld, model_densities, bg, tlp, m.l2_regularizers, d,
                            mapping=map_dev, map_offset=map_offset,
                            self_normalized=self._use_self_normalized_importance_weights,
                            own_samples_only=self._only_use_own_samples)
