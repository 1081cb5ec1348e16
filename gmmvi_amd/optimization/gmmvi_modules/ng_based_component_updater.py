"""Component updaters (reference: src/gmmvi/optimization/gmmvi_modules/ng_based_component_updater.py:8-527)."""
from ... import hip_ops


class NgBasedComponentUpdater:
    def __init__(self, model, temperature: float):
        self.model = model
        self.temperature = temperature
        self.last_success = None
        self.last_info = None
        self.want_info = False

    @staticmethod
    def build_from_config(config, gmm_wrapper):
        """ng_based_component_updater.py:33-57."""
        t = config["ng_based_updater_type"]
        if t == "trust-region":
            return KLConstrainedNgBasedComponentUpdater(gmm_wrapper, temperature=config['temperature'],
                                                        **config["ng_based_updater_config"])
        elif t == "direct":
            return DirectNgBasedComponentUpdater(gmm_wrapper, temperature=config['temperature'],
                                                 **config["ng_based_updater_config"])
        elif t == "iBLR":
            return NgBasedComponentUpdaterIblr(gmm_wrapper, temperature=config['temperature'],
                                               **config["ng_based_updater_config"])
        raise ValueError(f"config['ng_based_updater_type'] is '{t}' which is an unknown type")

    def apply_NG_update(self, expected_hessians_neg, expected_gradients_neg, stepsizes):
        raise NotImplementedError


class DirectNgBasedComponentUpdater(NgBasedComponentUpdater):
    """:83-141."""

    def apply_NG_update(self, expected_hessians_neg, expected_gradients_neg, stepsizes):
        m = self.model
        ctx = m.ctx
        if m.diagonal_covs:
            raise NotImplementedError("the reference's direct updater has no diagonal branch (:106 inverts chol_cov)")
        self.last_success = hip_ops.update_components_plain(
            ctx, "direct", m.means, m.chol_cov, ctx.asarray(expected_hessians_neg), ctx.asarray(expected_gradients_neg),
            ctx.asarray(stepsizes), m.initial_regularizer, m.l2_regularizers, m.num_received_updates)
        m.model._invalidate()


class NgBasedComponentUpdaterIblr(NgBasedComponentUpdater):
    """:144-223."""

    def apply_NG_update(self, expected_hessians_neg, expected_gradients_neg, stepsizes):
        m = self.model
        ctx = m.ctx
        if m.diagonal_covs:                                                               # :170-174, :188-197
            self.last_success, _, _ = hip_ops.update_components_diag(
                ctx, "iblr", m.means, m.chol_cov, ctx.asarray(expected_hessians_neg),
                ctx.asarray(expected_gradients_neg), ctx.asarray(stepsizes), 0.0, m.initial_regularizer, None,
                m.l2_regularizers, m.num_received_updates)
            m.model._invalidate()
            return
        self.last_success = hip_ops.update_components_plain(
            ctx, "iblr", m.means, m.chol_cov, ctx.asarray(expected_hessians_neg), ctx.asarray(expected_gradients_neg),
            ctx.asarray(stepsizes), m.initial_regularizer, m.l2_regularizers, m.num_received_updates)
        m.model._invalidate()


class KLConstrainedNgBasedComponentUpdater(NgBasedComponentUpdater):
    """:226-524: the per-component bracketing search runs on the device, one wavefront per component, with the
    reference's stop rules; means / Cholesky factors / last etas / l2 regularisers are updated in place."""

    def apply_NG_update(self, expected_hessians_neg, expected_gradients_neg, stepsizes):
        m = self.model
        ctx = m.ctx
        if m.diagonal_covs:                                                               # :447-453, :304-318
            succ, kl, probes = hip_ops.update_components_diag(
                ctx, "kl", m.means, m.chol_cov, ctx.asarray(expected_hessians_neg), ctx.asarray(expected_gradients_neg),
                ctx.asarray(stepsizes), self.temperature, m.initial_regularizer, m.last_log_etas, m.l2_regularizers,
                m.num_received_updates, want_info=self.want_info)
            self.last_success = succ
            self.last_info = (kl, probes)
            m.model._invalidate()
            return
        succ, kl, probes, packed = hip_ops.update_components_kl(
            ctx, m.means, m.chol_cov, ctx.asarray(expected_hessians_neg), ctx.asarray(expected_gradients_neg),
            ctx.asarray(stepsizes), self.temperature, m.initial_regularizer, m.last_log_etas, m.l2_regularizers,
            m.num_received_updates, want_info=self.want_info, want_packed=True)
        self.last_success = succ
        self.last_info = (kl, probes)
        m.model._packed = packed          # the kernel emitted the parameter blocks of the updated components
