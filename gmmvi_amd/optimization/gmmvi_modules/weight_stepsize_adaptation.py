"""Weight stepsize adaptation (reference: src/gmmvi/optimization/gmmvi_modules/weight_stepsize_adaptation.py:7-156)."""
import numpy as np

from ... import hip_ops
from ...device import get_context

FLOAT32_MIN = float(np.finfo(np.float32).min)


class WeightStepsizeAdaptation:
    def __init__(self, initial_stepsize, ctx=None):
        self.ctx = ctx if ctx is not None else get_context()
        # state[0] = stepsize, state[1] = previous ELBO proxy (elbo_history[-1], :139)
        self._state = self.ctx.asarray(np.array([initial_stepsize, FLOAT32_MIN], np.float32))

    @property
    def stepsize(self):
        return self._state.rows(0, 1)

    @staticmethod
    def build_from_config(config, gmm_wrapper):
        """:26-48 (underscore in "improvement_based")."""
        t = config["weight_stepsize_adapter_type"]
        if t == "fixed":
            return FixedWeightStepsizeAdaptation(**config['weight_stepsize_adapter_config'])
        elif t == "decaying":
            return DecayingWeightStepsizeAdaptation(**config['weight_stepsize_adapter_config'])
        elif t == "improvement_based":
            return ImprovementBasedWeightStepsizeAdaptation(gmm_wrapper, **config['weight_stepsize_adapter_config'])
        raise ValueError(f"config['weight_stepsize_adapter_type'] is '{t}' which is an unknown type")

    def _update_stepsize(self):
        pass

    def update_stepsize(self):
        """:53-61 -> device scalar [1]."""
        self._update_stepsize()
        return self.stepsize


class FixedWeightStepsizeAdaptation(WeightStepsizeAdaptation):
    """:64-72."""


class DecayingWeightStepsizeAdaptation(WeightStepsizeAdaptation):
    """:75-105."""
    def __init__(self, initial_stepsize, annealing_exponent):
        super().__init__(initial_stepsize)
        self.initial_stepsize = float(initial_stepsize)
        self.annealing_exponent = float(annealing_exponent)
        self.num_weight_updates = 0.0

    def _update_stepsize(self):
        s = self.initial_stepsize / (1.0 + self.num_weight_updates ** self.annealing_exponent)
        self._state.set(np.array([s, FLOAT32_MIN], np.float32))
        self.num_weight_updates += 1.0


class ImprovementBasedWeightStepsizeAdaptation(WeightStepsizeAdaptation):
    """:108-156: ELBO proxy sum_k w_k R_k[-1] - sum_k w_k log w_k against its previous value, on the device."""
    def __init__(self, model, initial_stepsize, min_stepsize, max_stepsize, stepsize_inc_factor, stepsize_dec_factor):
        super().__init__(initial_stepsize, model.ctx)
        self.model = model
        self.min_stepsize = min_stepsize
        self.max_stepsize = max_stepsize
        self.stepsize_inc_factor = stepsize_inc_factor
        self.stepsize_dec_factor = stepsize_dec_factor

    def _update_stepsize(self):
        m = self.model
        hip_ops.weight_stepsize_improvement(m.ctx, m.log_weights, m.reward_slot(0), self._state, self.min_stepsize,
                                            self.max_stepsize, self.stepsize_inc_factor, self.stepsize_dec_factor)
