"""Experiment set-up (reference: src/gmmvi/experiments/setup_experiment.py:10-160)."""
import numpy as np

from ..models.diagonal_gmm import DiagonalGMM
from ..models.full_cov_gmm import FullCovGMM
from ..models.gmm_wrapper import GmmWrapper


def init_experiment(config: dict):
    """setup_experiment.py:10-43 -> (target LNPDF, GmmWrapper)."""
    if "environment_config" in config.keys():
        target_fn = get_target_lnpdf(experiment=config["environment_name"],
                                     environment_config=config["environment_config"], seed=config["seed"])
    elif "target_fn" in config.keys():
        target_fn = config.pop("target_fn")
    else:
        raise ValueError("No target distribution was specified")
    gmm = construct_initial_mixture(num_dimensions=target_fn.get_num_dimensions(), **config["model_initialization"])
    initial_l2_regularizer = config["ng_estimator_config"].get('initial_l2_regularizer', 1e-12)
    gmm_wrapper = GmmWrapper(gmm, config["component_stepsize_adapter_config"]["initial_stepsize"],
                             initial_l2_regularizer, max_reward_history_length=10000)                # :40-41
    return target_fn, gmm_wrapper


def get_target_lnpdf(experiment, environment_config, seed):
    """setup_experiment.py:46-86: the three targets on the hot-path scope; the other benchmark posteriors of the
    reference (logistic regression, BNN, Talos) plug in through LNPDF (config key "target_fn")."""
    if experiment == "PlanarRobot4":
        from .target_distributions.planar_robot import make_four_goal
        return make_four_goal()
    elif experiment == "PlanarRobot1":
        from .target_distributions.planar_robot import make_single_goal
        return make_single_goal()
    elif experiment == "STM":
        from .target_distributions.student_t_mixture import make_target
        return make_target(**environment_config)
    elif experiment.startswith("GMM"):
        from .target_distributions.gmm import make_target
        return make_target(**environment_config)
    elif experiment.startswith("DIAGGMM"):
        from .target_distributions.diag_gmm import make_target
        return make_target(**environment_config)
    raise ValueError(f"get_target_lnpdf() was called with unknown experiment name: {experiment} "
                     f"(in scope: PlanarRobot1/4, STM, GMM*, DIAGGMM*; pass other targets as config['target_fn'])")


def construct_initial_mixture(num_dimensions, num_initial_components, prior_mean, prior_scale, use_diagonal_covs,
                              initial_cov=None):
    """setup_experiment.py:88-160: equal weights, means ~ N(prior_mean, prior_scale^2) from the global NumPy RNG,
    covariance initial_cov * I (or the prior covariance)."""
    if np.isscalar(prior_mean):
        prior_mean = prior_mean * np.ones(num_dimensions)
    if np.isscalar(prior_scale):
        prior_scale = prior_scale * np.ones(num_dimensions)
    weights = np.ones(num_initial_components, dtype=np.float32) / num_initial_components
    means = np.zeros((num_initial_components, num_dimensions), dtype=np.float32)
    if use_diagonal_covs:                                                                       # :129-141
        prior = np.array(prior_scale) ** 2
        initial_cov = prior if initial_cov is None else initial_cov * np.ones(num_dimensions)
        covs = np.ones((num_initial_components, num_dimensions), dtype=np.float32)
        for i in range(num_initial_components):
            if num_initial_components == 1:
                means[i] = prior_mean
            else:
                means[i] = prior_mean + np.sqrt(prior) * np.random.standard_normal([num_dimensions])
            covs[i] = initial_cov
        return DiagonalGMM(weights, means, covs)
    prior = np.diag(np.array(prior_scale) ** 2)
    initial_cov = prior if initial_cov is None else initial_cov * np.eye(num_dimensions)
    covs = np.ones((num_initial_components, num_dimensions, num_dimensions), dtype=np.float32)
    for i in range(num_initial_components):
        if num_initial_components == 1:
            means[i] = prior_mean
        else:
            means[i] = prior_mean + np.linalg.cholesky(prior) @ np.random.standard_normal([num_dimensions, 1])[:, 0]
        covs[i] = initial_cov
    return FullCovGMM(weights, means, covs)
