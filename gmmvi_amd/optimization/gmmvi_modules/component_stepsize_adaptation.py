"""Component stepsize adaptation (reference: src/gmmvi/optimization/gmmvi_modules/component_stepsize_adaptation.py:7-188)."""
import numpy as np

from ... import hip_ops


class ComponentStepsizeAdaptation:
    def __init__(self, gmm_wrapper, initial_stepsize: float):
        self.gmm_wrapper = gmm_wrapper
        self.initial_stepsize = initial_stepsize
        if not np.allclose(gmm_wrapper.stepsizes.numpy(), initial_stepsize):          # tf.assert_equal, :28
            raise ValueError("gmm_wrapper.stepsizes must equal initial_stepsize")

    @staticmethod
    def build_from_config(config, gmm_wrapper):
        """:30-52 (note the hyphen in "improvement-based")."""
        t = config["component_stepsize_adapter_type"]
        if t == "improvement-based":
            return ImprovementBasedComponentStepsizeAdaptation(gmm_wrapper, **config["component_stepsize_adapter_config"])
        elif t == "decaying":
            return DecayingComponentStepsizeAdaptation(gmm_wrapper, **config["component_stepsize_adapter_config"])
        elif t == "fixed":
            return FixedComponentStepsizeAdaptation(gmm_wrapper, **config["component_stepsize_adapter_config"])
        raise ValueError(f"config['component_stepsize_adapter_type'] is '{t}' which is an unknown type")

    def update_stepsize(self, current_stepsizes):
        raise NotImplementedError


class FixedComponentStepsizeAdaptation(ComponentStepsizeAdaptation):
    """:69-92."""
    def update_stepsize(self, current_stepsizes):
        return current_stepsizes


class DecayingComponentStepsizeAdaptation(ComponentStepsizeAdaptation):
    """:95-130: initial / (1 + num_received_updates ** annealing_exponent); O(K) host arithmetic, one upload."""
    def __init__(self, gmm_wrapper, annealing_exponent: float, initial_stepsize: float):
        super().__init__(gmm_wrapper, initial_stepsize)
        self.annealing_exponent = annealing_exponent

    def update_stepsize(self, current_stepsizes):
        n = self.gmm_wrapper.num_received_updates.numpy().astype(np.float64)
        new = self.initial_stepsize / (1 + np.power(n, self.annealing_exponent))
        return self.gmm_wrapper.ctx.asarray(new.astype(np.float32))


class ImprovementBasedComponentStepsizeAdaptation(ComponentStepsizeAdaptation):
    """:133-188: x1.15 if the last reward improved, x0.85 otherwise (clipped); updates the device array in place."""
    def __init__(self, gmm_wrapper, initial_stepsize: float, min_stepsize: float, max_stepsize: float,
                 stepsize_inc_factor: float, stepsize_dec_factor: float):
        super().__init__(gmm_wrapper, initial_stepsize)
        self.min_stepsize = min_stepsize
        self.max_stepsize = max_stepsize
        self.stepsize_inc_factor = stepsize_inc_factor
        self.stepsize_dec_factor = stepsize_dec_factor

    def update_stepsize(self, current_stepsizes):
        w = self.gmm_wrapper
        steps = w.ctx.asarray(current_stepsizes)
        hip_ops.component_stepsize_improvement(w.ctx, steps, w.reward_slot(1), w.reward_slot(0), self.min_stepsize,
                                               self.max_stepsize, self.stepsize_inc_factor, self.stepsize_dec_factor)
        return steps
