"""Weight updaters (reference: src/gmmvi/optimization/gmmvi_modules/weight_updater.py:7-281)."""
import numpy as np

from ... import hip_ops
from ...device import DeviceArray


class WeightUpdater:
    def __init__(self, model, temperature: float, use_self_normalized_importance_weights: bool):
        self.model = model
        self.temperature = temperature
        self.use_self_normalized_importance_weights = use_self_normalized_importance_weights
        self.last_info = None
        self.want_info = False

    @staticmethod
    def build_from_config(config, gmm_wrapper):
        """weight_updater.py:34-54."""
        t = config["weight_updater_type"]
        if t == "direct":
            return DirectWeightUpdater(gmm_wrapper, temperature=config['temperature'], **config["weight_updater_config"])
        elif t == "trust-region":
            return TrustRegionBasedWeightUpdater(gmm_wrapper, temperature=config['temperature'],
                                                 **config["weight_updater_config"])
        raise ValueError(f"config['weight_updater_type'] is '{t}' which is an unknown type")

    def _get_expected_log_ratios(self, samples, background_mixture_densities, target_lnpdfs):
        """weight_updater.py:56-75: density pass with the *updated* components, importance-weighted expectation of
        log p~ - beta log q per component, and the reward store (:73-74) -- the ELR kernel writes the rewards straight
        into the wrapper's ring-buffer slot."""
        m = self.model
        ctx = m.ctx
        model_densities, ld = m.log_densities_also_individual(samples)                 # :57
        e, _ = hip_ops.expected_log_ratios(ctx, ld, ctx.asarray(background_mixture_densities),
                                           ctx.asarray(target_lnpdfs), model_densities, self.temperature,
                                           m.log_weights, self.use_self_normalized_importance_weights,
                                           reward_out=m.next_reward_slot())
        m.commit_rewards()
        return e

    def update_weights(self, samples, background_mixture_densities, target_lnpdfs, stepsize):
        """weight_updater.py:77-100."""
        e = self._get_expected_log_ratios(samples, background_mixture_densities, target_lnpdfs)
        self._update_weights_from_expected_log_ratios(e, stepsize)

    def _stepsize_dev(self, stepsize):
        ctx = self.model.ctx
        if isinstance(stepsize, DeviceArray):
            return stepsize.reshape(-1).rows(0, 1)
        return ctx.asarray(np.array([float(stepsize)], np.float32))

    def _update_weights_from_expected_log_ratios(self, expected_log_ratios, stepsize):
        raise NotImplementedError


class DirectWeightUpdater(WeightUpdater):
    """weight_updater.py:106-141."""

    def _update_weights_from_expected_log_ratios(self, expected_log_ratios, stepsize):
        m = self.model
        if m.num_components > 1:                                                          # :136
            hip_ops.update_weights(m.ctx, "direct", m.log_weights, m.ctx.asarray(expected_log_ratios),
                                   self._stepsize_dev(stepsize), self.temperature)
            m.record_weights()                                                            # gmm_wrapper.py:182


class TrustRegionBasedWeightUpdater(WeightUpdater):
    """weight_updater.py:144-279: 50-step log-eta bracketing on the categorical KL, one wavefront on the device."""

    def _update_weights_from_expected_log_ratios(self, expected_log_ratios, kl_bound):
        m = self.model
        if m.num_components > 1:                                                          # :275
            self.last_info = hip_ops.update_weights(m.ctx, "trust-region", m.log_weights,
                                                    m.ctx.asarray(expected_log_ratios), self._stepsize_dev(kl_bound),
                                                    self.temperature, want_info=self.want_info)
            m.record_weights()
