"""Adapting the number of components (reference: src/gmmvi/optimization/gmmvi_modules/component_adaptation.py:14-302).

Adjacent to the hot path: host Python over the device buffers; the only heavy step -- the model
log-density of up to 100 000 candidate samples -- runs in the fused density kernel.
"""
import numpy as np
from scipy.special import logsumexp

FLOAT32_MAX = float(np.finfo(np.float32).max)


class ComponentAdaptation:
    def __init__(self):
        pass

    @staticmethod
    def build_from_config(config, gmm_wrapper, sample_db, target_distribution, prior_mean, initial_cov):
        """:45-82."""
        if config["num_component_adapter_type"] == "adaptive":
            return VipsComponentAdaptation(gmm_wrapper, sample_db, target_distribution, prior_mean, initial_cov,
                                           **config["num_component_adapter_config"])
        elif config["num_component_adapter_type"] == "fixed":
            return FixedComponentAdaptation(**config["num_component_adapter_config"])
        raise ValueError(f"config['num_component_adapter_type'] is '{config['num_component_adapter_type']}' "
                         f"which is an unknown type")

    def adapt_number_of_components(self, iteration):
        raise NotImplementedError


class FixedComponentAdaptation(ComponentAdaptation):
    """:88-101."""
    def adapt_number_of_components(self, iteration):
        pass


class VipsComponentAdaptation(ComponentAdaptation):
    """:104-300."""

    def __init__(self, model, sample_db, target_lnpdf, prior_mean, initial_cov, del_iters: int, add_iters: int,
                 max_components: int, thresholds_for_add_heuristic, min_weight_for_del_heuristic: float,
                 num_database_samples: int, num_prior_samples: int):
        super().__init__()
        self.model = model
        d = model.num_dimensions
        if (prior_mean is not None) and (initial_cov is not None):
            self.prior_mean = np.broadcast_to(np.asarray(prior_mean, np.float64), (d,)).copy()     # :152-155
            self.prior_var = np.broadcast_to(np.asarray(initial_cov, np.float64), (d,)).copy()
        else:
            self.prior_mean = self.prior_var = None
        self.num_prior_samples = num_prior_samples
        self.target_lnpdf = target_lnpdf
        self.sample_db = sample_db
        self.del_iters = del_iters
        self.add_iters = add_iters
        self.max_components = max_components
        self.num_db_samples = num_database_samples
        self.num_calls_to_add_heuristic = 0
        self.thresholds_for_addHeuristic = np.asarray(thresholds_for_add_heuristic, np.float32)
        self.min_weight_for_del_heuristic = min_weight_for_del_heuristic
        self.reward_improvements = np.zeros(0, np.float32)
        self.filter_delay = int(np.floor(self.del_iters / 3))                                        # :172
        xs = np.arange(-self.filter_delay, self.filter_delay, dtype=np.float64)                     # :174
        sd = self.del_iters / 8.0
        kern = np.exp(-0.5 * (xs / sd) ** 2) / (sd * np.sqrt(2 * np.pi))
        self.kernel = (kern / kern.sum()).astype(np.float32)                                         # :175
        self.rng = np.random.default_rng(getattr(model, "seed", 0))   # tf.random.uniform / shuffle stand-in

    def _prior_entropy(self):
        """entropy of the diagonal prior (:156, models/diagonal_gmm.py:33-34)."""
        d = self.model.num_dimensions
        return 0.5 * d * (np.log(2 * np.pi) + 1) + np.sum(np.log(np.sqrt(self.prior_var)))

    def adapt_number_of_components(self, iteration):
        """:177-190."""
        iteration = int(iteration)
        add_due = iteration > 1 and iteration % self.add_iters == 0
        # The candidate draw of the add heuristic (host generator, index upload, two gathers) depends neither on the deletions nor
        # on anything the iteration's launches still compute, while the deletion rule has to wait for their rewards: draw first,
        # so that the host works while the GPU finishes.  Same generator sequence as the reference order (the deletion rule draws
        # nothing); only when K sits at max_components -- where a deletion decides whether the add happens at all -- the order
        # of the reference is kept.
        drawn = None
        if add_due and self.model.num_components < self.max_components:
            drawn = self.select_samples_for_adding_heuristic()
        if iteration > self.del_iters:
            self.delete_bad_components()
        if add_due:
            if self.model.num_components < self.max_components:
                self.add_new_component(drawn)

    def add_at_best_location(self, samples, target_lnpdfs):
        """:192-226.  The candidate search runs on the device (gmmvi_add_heuristic_argmax: the same fp64 arithmetic, first
        maximum): neither the 10^5 candidate densities nor the chosen sample are read back."""
        m = self.model
        ctx = m.ctx
        it = self.num_calls_to_add_heuristic % len(self.thresholds_for_addHeuristic)
        samples = ctx.asarray(samples) if not hasattr(samples, "ptr") else samples
        target_lnpdfs = ctx.asarray(np.asarray(target_lnpdfs, np.float32)) if not hasattr(target_lnpdfs, "ptr") else target_lnpdfs
        model_log_densities = m.log_density(samples)
        init_weight = 1e-29
        a = self.rng.random()                                                                        # :208
        if self.prior_var is not None:
            des_entropy = m.get_average_entropy() * a + self._prior_entropy() * (1 - a)
        else:
            des_entropy = m.get_average_entropy()
        best = ctx.empty((1,), np.int32)
        ctx.check(ctx.lib.gmmvi_add_heuristic_argmax(ctx.handle, model_log_densities.ptr, target_lnpdfs.ptr,
                                                     int(model_log_densities.shape[0]),
                                                     float(self.thresholds_for_addHeuristic[it]), best.ptr))
        from ... import hip_ops
        new_mean = hip_ops.gather_rows(ctx, samples, best)                                           # [1, D] on the device
        d = m.num_dimensions
        h_unscaled = 0.5 * d * (np.log(2.0 * np.pi) + 1)
        c = np.exp((2 * (des_entropy - h_unscaled)) / d)
        new_cov = c * np.ones(d) if m.diagonal_covs else c * np.eye(d)                               # :220-223
        m.add_component(init_weight, new_mean, new_cov, [self.thresholds_for_addHeuristic[it]], [des_entropy])

    def select_samples_for_adding_heuristic(self):
        """:228-249."""
        self.num_calls_to_add_heuristic += 1
        samples, target_lnpdfs = self.sample_db.get_random_sample(self.num_db_samples, self.rng)
        prior_samples = np.zeros((0, self.model.num_dimensions), np.float32)
        if self.num_prior_samples > 0:
            prior_samples = (self.prior_mean + np.sqrt(self.prior_var) * self.rng.standard_normal(
                (self.num_prior_samples, self.model.num_dimensions))).astype(np.float32)
            self.sample_db.num_samples_written.assign_add(self.num_prior_samples)
        return samples, target_lnpdfs, prior_samples

    def add_new_component(self, drawn=None):
        """:251-259 (``drawn``: the candidates, when adapt_number_of_components selected them ahead of the deletion step)."""
        samples, target_lnpdfs, prior_samples = drawn if drawn is not None else self.select_samples_for_adding_heuristic()
        if self.num_prior_samples > 0:
            ctx = self.model.ctx
            plp = self.target_lnpdf.log_density(ctx.asarray(prior_samples))
            samples = ctx.asarray(np.concatenate([samples.numpy(), prior_samples]))
            target_lnpdfs = ctx.asarray(np.concatenate([target_lnpdfs.numpy(),
                                                        np.asarray(plp.numpy() if hasattr(plp, "numpy") else plp)]).astype(np.float32))
        self.add_at_best_location(samples, target_lnpdfs)

    def deletion_criteria(self):
        """The three per-component quantities of :261-296: relative improvement of the smoothed reward, the largest weight
        (actual or greedy) over the window, and whether the component is old enough."""
        m = self.model
        ks = self.kernel.size
        win = ks + self.del_iters
        rh = m.reward_window(win).astype(np.float64)          # last ks+del_iters columns of reward_history
        wh = m.weight_window(win).astype(np.float64)
        kern = self.kernel.astype(np.float64)
        cur = np.mean(rh[:, -ks:] * kern[None, :], axis=1)
        old = np.mean(rh[:, :ks] * kern[None, :], axis=1)     # [-ks-del_iters : -del_iters]
        old = old - np.max(cur)
        cur = cur - np.max(cur)
        with np.errstate(divide='ignore', invalid='ignore'):
            reward_improvements = (cur - old) / np.abs(old)
        max_actual_weights = np.max(wh[:, :-1], axis=1)                                              # :286
        with np.errstate(over='ignore', invalid='ignore'):
            max_greedy_weights = np.max(np.exp(rh - logsumexp(rh, axis=0, keepdims=True)), axis=1)   # :287-289
        max_weights = np.maximum(max_actual_weights, max_greedy_weights)
        is_old_enough = rh[:, -self.del_iters] != -FLOAT32_MAX                                       # :294
        return reward_improvements, max_weights, is_old_enough

    def bad_components(self):
        """:291-296: indices of the components the heuristic deletes now."""
        reward_improvements, max_weights, is_old_enough = self.deletion_criteria()
        self.reward_improvements = reward_improvements.astype(np.float32)
        is_stagnating = reward_improvements <= 0.4
        is_low_weight = max_weights < self.min_weight_for_del_heuristic
        return np.where(is_stagnating & is_low_weight & is_old_enough)[0]

    def delete_bad_components(self):
        """:261-300."""
        bad = self.bad_components()
        for idx in sorted(bad.tolist(), reverse=True):                                               # :298-300
            self.model.remove_component(idx)
        return bad
