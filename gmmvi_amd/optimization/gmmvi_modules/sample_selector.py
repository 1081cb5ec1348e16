"""Sample selectors (reference: src/gmmvi/optimization/gmmvi_modules/sample_selector.py:30-339)."""
import numpy as np

from ... import hip_ops
from ...device import DeviceArray
from ...models.gmm import STREAM_COMPONENT_NORMALS


class SampleSelector:
    def __init__(self, target_distribution, model, sample_db):
        self.target_distribution = target_distribution
        self.model = model
        self.sample_db = sample_db

    @staticmethod
    def build_from_config(config, gmm_wrapper, sample_db, target_distribution):
        """sample_selector.py:38-64."""
        if config["sample_selector_type"] == "component-based":
            return VipsSampleSelector(target_distribution, gmm_wrapper, sample_db, **config['sample_selector_config'])
        elif config["sample_selector_type"] == "mixture-based":
            return LinSampleSelector(target_distribution, gmm_wrapper, sample_db, **config['sample_selector_config'])
        raise ValueError(
            f"config['sample_selector_type'] is '{config['sample_selector_type']}' which is an unknown type")

    def target_uld(self, samples):
        return self.target_distribution.log_density(samples)

    def get_target_grads(self, samples):
        """sample_selector.py:69-78 -> (gradient, target).  Built-in targets evaluate log-density and gradient in one
        fused kernel; user targets must provide log_density_and_grad (there is no autodiff here)."""
        target, gradient = self.target_distribution.log_density_and_grad(samples)
        ctx = self.model.ctx
        return ctx.asarray(gradient), ctx.asarray(target)

    def select_samples(self):
        raise NotImplementedError


class VipsSampleSelector(SampleSelector):
    """sample_selector.py:103-219 ("M")."""

    def __init__(self, target_distribution, model, sample_db, desired_samples_per_component: int,
                 ratio_reused_samples_to_desired: float):
        super().__init__(target_distribution, model, sample_db)
        self.desired_samples_per_component = int(desired_samples_per_component)
        self.reused_samples_per_component = int(np.floor(ratio_reused_samples_to_desired
                                                         * desired_samples_per_component))
        self.eps_override = None       # callable(n, d) -> host normals, for bit-reproducible parity runs
        self.fuse_background = True    # fused background + model density pass when nothing is reused

    def get_effective_samples(self, model_densities, oldsamples_pdf):
        """sample_selector.py:140-158: ESS_k = 1 / sum_n softmax_n(ld[k,n] - bg[n])^2 (device reduction)."""
        ctx = self.model.ctx
        ld = ctx.asarray(model_densities); bg = ctx.asarray(oldsamples_pdf)
        k, n = ld.shape
        zeros_n = ctx.zeros((n,)); zeros_k = ctx.zeros((k,))
        _, ess = hip_ops.expected_log_ratios(ctx, ld, bg, zeros_n, zeros_n, 1.0, zeros_k, True, want_ess=True)
        return ess

    def sample_where_needed(self, samples, oldsamples_pdf, num_desired_samples=None):
        """sample_selector.py:160-202 -> (new_samples, new_target_lnpdfs, new_target_grads, mapping)."""
        if num_desired_samples is None:
            num_desired_samples = self.desired_samples_per_component
        k = self.model.num_components
        if samples.shape[0] == 0:
            n_eff = np.zeros(k, np.int64)
        else:
            ld = self.model.component_log_densities(samples)
            n_eff = np.floor(self.get_effective_samples(ld, oldsamples_pdf).numpy()).astype(np.int64)
        n_add = np.maximum(1, num_desired_samples - n_eff)
        first = int(self.sample_db.num_samples_written)
        eps = None
        if self.eps_override is not None:
            eps = self.eps_override(int(n_add.sum()), self.model.num_dimensions)
        new_samples, mapping = self.model.sample_from_components_no_shuffle(
            n_add, first_index=first, eps=eps, stream_id=STREAM_COMPONENT_NORMALS)
        new_target_grads, new_target_lnpdfs = self.get_target_grads(new_samples)
        key = n_add.tobytes()
        if getattr(self, "_mapping_key", None) != key:
            self._mapping_key = key
            self._last_mapping_host = np.repeat(np.arange(k, dtype=np.int32), n_add)
        self._last_counts = n_add
        return new_samples, new_target_lnpdfs, new_target_grads, mapping

    def select_samples(self):
        """sample_selector.py:204-219 -> (samples, mapping, sample_dist_densities, target_lnpdfs, target_grads)."""
        n_reuse = self.reused_samples_per_component * self.model.num_components
        oldsamples_pdf, samples, _, _, _ = self.sample_db.get_newest_samples(n_reuse)
        num_reused = samples.shape[0]
        new_samples, new_lp, new_grads, mapping = self.sample_where_needed(samples, oldsamples_pdf)
        self.sample_db.add_samples(new_samples, self.model.means, self.model.chol_cov, new_lp, new_grads, mapping,
                                   mapping_host=self._last_mapping_host, packed=self.model.packed,
                                   counts=self._last_counts)
        num_new = new_samples.shape[0]
        # nothing reused: the active samples come from the current components, so the background mixture shares its
        # components with the model and both are evaluated in one sweep
        fuse = self.model.model if (num_reused == 0 and self.fuse_background) else None
        oldsamples_pdf, samples, mapping, target_lnpdfs, target_grads = \
            self.sample_db.get_newest_samples(num_reused + num_new, fuse_with_model=fuse)
        return samples, mapping, oldsamples_pdf, target_lnpdfs, target_grads


class LinSampleSelector(SampleSelector):
    """sample_selector.py:221-339 ("P")."""

    def __init__(self, target_distribution, model, sample_db, desired_samples_per_component: int,
                 ratio_reused_samples_to_desired: float):
        super().__init__(target_distribution, model, sample_db)
        self.desired_samples_per_component = int(desired_samples_per_component)
        self.reused_samples_per_component = int(np.floor(ratio_reused_samples_to_desired
                                                         * desired_samples_per_component))

    def get_effective_samples(self, model_densities, oldsamples_pdf):
        """sample_selector.py:258-277 (mixture-level ESS; one value)."""
        from scipy.special import logsumexp
        lw = np.asarray(model_densities, np.float64) - np.asarray(oldsamples_pdf, np.float64)
        lw = lw - logsumexp(lw)
        return 1.0 / np.sum(np.exp(lw) ** 2)

    def sample_where_needed(self):
        """sample_selector.py:279-325 -> (new_samples, mapping, num_reused_samples)."""
        n_reuse = self.reused_samples_per_component * self.model.num_components
        oldsamples_pdf, old_samples, _, _, _ = self.sample_db.get_newest_samples(n_reuse)
        num_reused = old_samples.shape[0]
        if num_reused == 0:
            n_eff = 0
        else:
            n_eff = int(np.floor(self.get_effective_samples(self.model.log_density(old_samples).numpy(),
                                                            oldsamples_pdf.numpy())))
        n_add = max(1, self.desired_samples_per_component - n_eff)
        new_samples, mapping = self.model.sample(n_add)
        return new_samples, mapping, num_reused

    def select_samples(self):
        """sample_selector.py:327-339."""
        new_samples, mapping, num_reused = self.sample_where_needed()
        new_grads, new_lp = self.get_target_grads(new_samples)
        mapping = np.asarray(mapping, np.int32)
        self.sample_db.add_samples(new_samples, self.model.means, self.model.chol_cov, new_lp, new_grads,
                                   self.model.ctx.asarray(mapping, np.int32), mapping_host=mapping,
                                   packed=self.model.packed)
        n_iter = num_reused + new_samples.shape[0]
        oldsamples_pdf, samples, mapping, target_lnpdfs, target_grads = self.sample_db.get_newest_samples(n_iter)
        return samples, mapping, oldsamples_pdf, target_lnpdfs, target_grads
