"""Experiment set-up: target distribution + initial mixture from a config dictionary.

Drop-in for the reference's ``gmmvi.experiments.setup_experiment`` (src/gmmvi/experiments/setup_experiment.py:10-160): same
three functions, same arguments.  What is contractual and therefore reproduced exactly: the experiment names (:46-86), the
``GmmWrapper`` arguments (:36-41) and the *law and draw order* of the initial means (:119-155: one ``standard_normal`` draw of
``num_dimensions`` values per component from the global NumPy generator, scaled by sqrt(prior), no draw for a single
component), because ``start_seed`` must give the reference's initial mixture.  The targets live in a name -> factory table.
"""
import importlib

import numpy as np

from ..models.diagonal_gmm import DiagonalGMM
from ..models.full_cov_gmm import FullCovGMM
from ..models.gmm_wrapper import GmmWrapper

_MAX_REWARD_HISTORY = 10000        # setup_experiment.py:40-41

# experiment name (exact, or prefix when the key ends with "*") -> (module below target_distributions, factory, takes the config)
_TARGETS = {
    "PlanarRobot4": ("planar_robot", "make_four_goal", False),
    "PlanarRobot1": ("planar_robot", "make_single_goal", False),
    "STM": ("student_t_mixture", "make_target", True),
    "GMM*": ("gmm", "make_target", True),
    "DIAGGMM*": ("diag_gmm", "make_target", True),
}


def _lookup_target(experiment):
    entry = _TARGETS.get(experiment)
    if entry is None:
        # prefixes, longest first ("DIAGGMM..." must not be taken for "GMM...": it does not start with it, but keep the order explicit)
        for key in sorted((k for k in _TARGETS if k.endswith("*")), key=len, reverse=True):
            if experiment.startswith(key[:-1]):
                entry = _TARGETS[key]
                break
    return entry


def get_target_lnpdf(experiment, environment_config, seed):
    """:46-86.  In scope here: PlanarRobot1/4, STM, GMM*, DIAGGMM*; the reference's other benchmark posteriors (logistic
    regression, BNN, Talos) plug in as ``config['target_fn']`` through the LNPDF interface."""
    entry = _lookup_target(experiment)
    if entry is None:
        raise ValueError(f"get_target_lnpdf() was called with unknown experiment name: {experiment} "
                         f"(in scope: PlanarRobot1/4, STM, GMM*, DIAGGMM*; pass other targets as config['target_fn'])")
    module_name, factory_name, takes_config = entry
    module = importlib.import_module(f"{__package__}.target_distributions.{module_name}")
    factory = getattr(module, factory_name)
    return factory(**environment_config) if takes_config else factory()


def init_experiment(config: dict):
    """:10-43 -> (target LNPDF, GmmWrapper around the initial mixture)."""
    if "environment_config" in config:
        target = get_target_lnpdf(experiment=config["environment_name"], environment_config=config["environment_config"],
                                  seed=config["seed"])
    elif "target_fn" in config:
        target = config.pop("target_fn")
    else:
        raise ValueError("No target distribution was specified")
    mixture = construct_initial_mixture(num_dimensions=target.get_num_dimensions(), **config["model_initialization"])
    l2 = config["ng_estimator_config"].get("initial_l2_regularizer", 1e-12)
    wrapper = GmmWrapper(mixture, config["component_stepsize_adapter_config"]["initial_stepsize"], l2,
                         max_reward_history_length=_MAX_REWARD_HISTORY)
    return target, wrapper


def _per_dimension(value, num_dimensions):
    return value * np.ones(num_dimensions) if np.isscalar(value) else np.asarray(value)


def _initial_means(prior_mean, prior_variances, num_components):
    """[K, D] fp32.  A single component sits on the prior mean; otherwise every component draws ``D`` standard normals (global
    NumPy generator, one call per component, component order) scaled by the prior's standard deviations (:131-133, :151-155:
    the full-covariance branch multiplies by chol(diag(prior)) = diag(sqrt(prior)), the same numbers)."""
    means = np.zeros((num_components, prior_mean.shape[0]), dtype=np.float32)
    if num_components == 1:
        means[0] = prior_mean
        return means
    scale = np.sqrt(prior_variances)
    for k in range(num_components):
        means[k] = prior_mean + scale * np.random.standard_normal([prior_mean.shape[0]])
    return means


def construct_initial_mixture(num_dimensions, num_initial_components, prior_mean, prior_scale, use_diagonal_covs,
                              initial_cov=None):
    """:88-160: equal weights, means ~ N(prior_mean, diag(prior_scale²)), every covariance ``initial_cov``·I (the prior
    covariance when ``initial_cov`` is None); ``DiagonalGMM`` ([K, D] variances) or ``FullCovGMM`` ([K, D, D])."""
    prior_mean = _per_dimension(prior_mean, num_dimensions)
    prior_variances = np.array(_per_dimension(prior_scale, num_dimensions)) ** 2
    weights = np.ones(num_initial_components, dtype=np.float32) / num_initial_components
    means = _initial_means(prior_mean, prior_variances, num_initial_components)
    variances = prior_variances if initial_cov is None else initial_cov * np.ones(num_dimensions)
    if use_diagonal_covs:
        covs = np.tile(np.asarray(variances, np.float32), (num_initial_components, 1))
        return DiagonalGMM(weights, means, covs)
    cov = (np.diag(variances) if initial_cov is None else initial_cov * np.eye(num_dimensions)).astype(np.float32)
    covs = np.tile(cov, (num_initial_components, 1, 1))
    return FullCovGMM(weights, means, covs)
