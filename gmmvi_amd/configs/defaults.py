"""Default hyper-parameters: the values of the reference's configs/module_configs/**.yml and
configs/experiment_configs/*.yml (in-scope experiments), keyed by codename letter / experiment id."""

MODULE_DEFAULTS = {
    # ng_estimator/*.yml
    "Z": {"ng_estimator_type": "MORE",
          "ng_estimator_config": {"initial_l2_regularizer": 1e-12, "only_use_own_samples": False,
                                  "use_self_normalized_importance_weights": True}},
    "S": {"ng_estimator_type": "Stein",
          "ng_estimator_config": {"only_use_own_samples": False, "use_self_normalized_importance_weights": True}},
    # component_adaptation/*.yml
    "A": {"num_component_adapter_type": "adaptive",
          "num_component_adapter_config": {"del_iters": 100, "add_iters": 30, "max_components": 1000,
                                           "thresholds_for_add_heuristic": [5000., 1000.0, 500.0, 200.0, 100.0, 50.0],
                                           "min_weight_for_del_heuristic": 1.0e-6, "num_database_samples": 100000,
                                           "num_prior_samples": 0}},
    "E": {"num_component_adapter_type": "fixed", "num_component_adapter_config": {}},
    # sample_selector/*.yml
    "P": {"sample_selector_type": "mixture-based",
          "sample_selector_config": {"desired_samples_per_component": 100, "ratio_reused_samples_to_desired": 0.}},
    "M": {"sample_selector_type": "component-based",
          "sample_selector_config": {"desired_samples_per_component": 100, "ratio_reused_samples_to_desired": 2.}},
    # ng_based_component_updater/*.yml
    "I": {"ng_based_updater_type": "direct", "ng_based_updater_config": {}},
    "Y": {"ng_based_updater_type": "iBLR", "ng_based_updater_config": {}},
    "T": {"ng_based_updater_type": "trust-region", "ng_based_updater_config": {}},
    # component_stepsize_adaptation/*.yml
    "F": {"component_stepsize_adapter_type": "fixed", "component_stepsize_adapter_config": {"initial_stepsize": 1.0e-5}},
    "D": {"component_stepsize_adapter_type": "decaying",
          "component_stepsize_adapter_config": {"initial_stepsize": 1., "annealing_exponent": 0.55}},
    "R": {"component_stepsize_adapter_type": "improvement-based",
          "component_stepsize_adapter_config": {"initial_stepsize": 1., "min_stepsize": 0.001, "max_stepsize": 1.,
                                                "stepsize_inc_factor": 1.15, "stepsize_dec_factor": 0.85}},
    # weight_updater/*.yml
    "U": {"weight_updater_type": "direct", "weight_updater_config": {"use_self_normalized_importance_weights": True}},
    "O": {"weight_updater_type": "trust-region",
          "weight_updater_config": {"use_self_normalized_importance_weights": True}},
    # weight_stepsize_adaptation/*.yml  ("G" ships with annealing_exponent: TODO in the reference)
    "X": {"weight_stepsize_adapter_type": "fixed", "weight_stepsize_adapter_config": {"initial_stepsize": 1.}},
    "G": {"weight_stepsize_adapter_type": "decaying",
          "weight_stepsize_adapter_config": {"initial_stepsize": 1., "annealing_exponent": "TODO"}},
    "N": {"weight_stepsize_adapter_type": "improvement_based",
          "weight_stepsize_adapter_config": {"initial_stepsize": 1., "min_stepsize": 0.0001, "max_stepsize": 1.,
                                             "stepsize_inc_factor": 1.15, "stepsize_dec_factor": 0.85}},
}


def _experiment(name, env_cfg, n_init, prior_scale, initial_cov, log_interval, max_db=10000000):
    return {"start_seed": 10000, "environment_name": name, "environment_config": env_cfg,
            "model_initialization": {"use_diagonal_covs": False, "num_initial_components": n_init, "prior_mean": 0.,
                                     "prior_scale": prior_scale, "initial_cov": initial_cov},
            "gmmvi_runner_config": {"log_metrics_interval": log_interval},
            "use_sample_database": True, "max_database_size": max_db, "temperature": 1.}


EXPERIMENT_DEFAULTS = {
    "stm20": _experiment("STM", {"num_dimensions": 20, "harder_setting": False, "use_matlab_target": False},
                         20, 100., 300., 1000),
    "stm300": _experiment("STM", {"num_dimensions": 300, "harder_setting": True, "use_matlab_target": False},
                          20, 100., 300., 50, max_db=100000),
    "gmm20": _experiment("GMM", {"num_dimensions": 20}, 1, 31.63, 1000., 100),
    "gmm100": _experiment("GMM", {"num_dimensions": 100}, 1, 31.63, 1000., 100),
    "planar_robot_4": _experiment("PlanarRobot4", {}, 300, [1.] + [0.2] * 9, [0.0625] + [0.0025] * 9, 10),
}
