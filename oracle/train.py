"""Oracle restatement of the orchestrator: GMMVI.train_iter / _run_updates (optimization/gmmvi.py:105-174),
the initial-mixture law (experiments/setup_experiment.py:88-160) and the runner's ELBO estimate
(gmmvi_runner.py:83-100,131-135).  TEST INFRASTRUCTURE.
"""
import time
import numpy as np

from . import gmm as ogmm
from . import philox, sample_db as odb, stein, more, updaters, weights as oweights, stepsizes, adaptation


def construct_initial_mixture(num_dimensions, num_initial_components, prior_mean, prior_scale, initial_cov, rng,
                              dtype=np.float64, use_diagonal_covs=False):
    """setup_experiment.py:88-160, NumPy Generator instead of np.random global."""
    prior_mean = np.broadcast_to(np.asarray(prior_mean, float), (num_dimensions,))
    prior_scale = np.broadcast_to(np.asarray(prior_scale, float), (num_dimensions,))
    if use_diagonal_covs:                                                                        # :129-141
        prior = prior_scale ** 2
        cov = prior if initial_cov is None else np.asarray(initial_cov, float) * np.ones(num_dimensions)
        means = np.zeros((num_initial_components, num_dimensions))
        for i in range(num_initial_components):
            means[i] = prior_mean if num_initial_components == 1 else \
                prior_mean + np.sqrt(prior) * rng.standard_normal(num_dimensions)
        w = np.ones(num_initial_components, np.float32) / num_initial_components
        covs = np.broadcast_to(cov.astype(np.float32), (num_initial_components, num_dimensions))
        return ogmm.DiagonalGMM(w, means.astype(np.float32), covs, dtype=dtype)
    prior = np.diag(prior_scale ** 2)
    cov = prior if initial_cov is None else np.asarray(initial_cov, float) * np.eye(num_dimensions)
    w = np.ones(num_initial_components) / num_initial_components
    means = np.zeros((num_initial_components, num_dimensions))
    for i in range(num_initial_components):
        if num_initial_components == 1:
            means[i] = prior_mean
        else:
            means[i] = prior_mean + np.linalg.cholesky(prior) @ rng.standard_normal(num_dimensions)
    # the reference stores means/covs as float32 before building the model (:131-133, :159-160)
    means = means.astype(np.float32)
    covs = np.broadcast_to(cov.astype(np.float32), (num_initial_components, num_dimensions, num_dimensions))
    return ogmm.FullCovGMM(w.astype(np.float32), means, covs, dtype=dtype)


class OracleGMMVI:
    """One object = GMMVI + its modules, SAMTRON-style by default (codename letters in brackets)."""

    def __init__(self, target, model, *, temperature=1.0, seed=0,
                 desired_samples_per_component=100, ratio_reused_samples_to_desired=0.0,      # [M]
                 sample_selector="component-based",
                 ng_estimator="Stein", only_use_own_samples=False, use_self_normalized_importance_weights=True,
                 initial_l2_regularizer=1e-12,
                 updater="trust-region",                                                           # [T]
                 component_stepsize="improvement-based",
                 component_stepsize_config=None,
                 weight_updater="trust-region", weight_stepsize="improvement_based", weight_stepsize_config=None,
                 adaptive=None, max_database_size=10000000, keep_samples=True,
                 max_reward_history_length=10000, host_rng=None):
        cs = dict(initial_stepsize=1.0, min_stepsize=0.001, max_stepsize=1.0, stepsize_inc_factor=1.15,
                  stepsize_dec_factor=0.85)
        cs.update(component_stepsize_config or {})
        ws = dict(initial_stepsize=1.0, min_stepsize=0.0001, max_stepsize=1.0, stepsize_inc_factor=1.15,
                  stepsize_dec_factor=0.85)
        ws.update(weight_stepsize_config or {})
        self.temperature = temperature
        self.target = target
        self.model = ogmm.GmmWrapper(model, cs["initial_stepsize"], initial_l2_regularizer, max_reward_history_length)
        self.sample_db = odb.SampleDB(model.num_dimensions, model.diagonal_covs, keep_samples, max_database_size, model.dtype)
        sel = odb.VipsSampleSelector if sample_selector == "component-based" else odb.LinSampleSelector
        self.sample_selector = sel(target, self.model, self.sample_db, desired_samples_per_component,
                                   ratio_reused_samples_to_desired, seed)
        self.ng_estimator = ng_estimator
        self.only_use_own_samples = only_use_own_samples
        self.snis = use_self_normalized_importance_weights
        self.updater = updater
        self.component_stepsize = component_stepsize
        self.cs = cs
        self.weight_updater = weight_updater
        if weight_stepsize == "improvement_based":
            self.weight_stepsize_adapter = stepsizes.WeightStepsizeImprovement(**ws)
        elif weight_stepsize == "fixed":
            self.weight_stepsize_adapter = stepsizes.WeightStepsizeFixed(ws["initial_stepsize"])
        else:
            self.weight_stepsize_adapter = stepsizes.WeightStepsizeDecaying(ws["initial_stepsize"],
                                                                            ws["annealing_exponent"])
        self.host_rng = host_rng if host_rng is not None else np.random.default_rng(seed)
        if adaptive:
            self.num_component_adapter = adaptation.VipsComponentAdaptation(
                self.model, self.sample_db, target, rng=self.host_rng, **adaptive)
        else:
            self.num_component_adapter = adaptation.FixedComponentAdaptation()
        self.num_updates = 0
        self.seed = seed
        self.last = {}

    def _component_stepsizes(self):
        if self.component_stepsize == "improvement-based":
            return stepsizes.component_stepsize_improvement(
                self.model.stepsizes, self.model.reward_history, self.cs["min_stepsize"], self.cs["max_stepsize"],
                self.cs["stepsize_inc_factor"], self.cs["stepsize_dec_factor"])
        if self.component_stepsize == "decaying":
            return stepsizes.component_stepsize_decaying(self.model.num_received_updates, self.cs["initial_stepsize"],
                                                         self.cs["annealing_exponent"])
        return stepsizes.component_stepsize_fixed(self.model.stepsizes)

    def run_updates(self, samples, mapping, bg, tlp, tgrad):
        """gmmvi.py:163-174."""
        m = self.model
        m.update_stepsizes(self._component_stepsizes())                                          # :165-166
        if self.ng_estimator == "Stein":
            h_neg, g_neg = stein.get_expected_hessian_and_grad(m, samples, mapping, bg, tlp, tgrad,
                                                               self.only_use_own_samples, self.snis)   # :167-168
        else:
            h_neg, g_neg = more.get_expected_hessian_and_grad(m, m.l2_regularizers, samples, mapping, bg, tlp,
                                                              self.only_use_own_samples, self.snis)
        info = {"h_neg": h_neg, "g_neg": g_neg, "stepsizes": m.stepsizes.copy()}
        if self.updater == "trust-region":                                                        # :169
            succ, etas, kls, nprobes = updaters.apply_ng_update_kl(m, h_neg, g_neg, m.stepsizes, self.temperature,
                                                                   traces=[])
            info.update(success=succ, etas=etas, kls=kls, n_probes=nprobes)
        elif self.updater == "direct":
            info.update(success=updaters.apply_ng_update_direct(m, h_neg, g_neg, m.stepsizes))
        else:
            info.update(success=updaters.apply_ng_update_iblr(m, h_neg, g_neg, m.stepsizes))
        wstep = self.weight_stepsize_adapter.update_stepsize(m)                                   # :172
        elr = oweights.get_expected_log_ratios(m, samples, bg, tlp, self.temperature, self.snis)  # :173
        if self.weight_updater == "trust-region":
            oweights.trust_region_update(m, elr, wstep, self.temperature)
        else:
            oweights.direct_update(m, elr, wstep, self.temperature)
        self.num_updates += 1                                                                     # :174
        info.update(weight_stepsize=wstep, expected_log_ratios=elr)
        return info

    def train_iter(self):
        """gmmvi.py:146-161."""
        samples, mapping, bg, tlp, tgrad = self.sample_selector.select_samples()
        info = self.run_updates(samples, mapping, bg, tlp, tgrad)
        self.num_component_adapter.adapt_number_of_components(self.num_updates)
        info.update(samples=samples, mapping=mapping, background=bg, target_lnpdfs=tlp, target_grads=tgrad)
        self.last = info
        return info

    def elbo(self, num_samples, seed, first_index=0):
        """gmmvi_runner.py:83-100,131-135: ELBO = mean log p~(x) + beta * (-mean log q(x)), x ~ q."""
        x, _ = self.model.model.sample(num_samples, seed, first_index)
        entropy = -np.mean(self.model.log_density(x))
        mean_reward = np.mean(self.target.log_density(x))
        return mean_reward + self.temperature * entropy, entropy, mean_reward


def time_cpu_baseline(make_algo, warmup=1, iters=3):
    """Median wall time of train_iter for bench.py's cpu_baseline leg ("port": this restatement)."""
    algo = make_algo()
    for _ in range(warmup):
        algo.train_iter()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        algo.train_iter()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), algo
