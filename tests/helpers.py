"""Shared builders for the GPU-vs-oracle end-to-end tests and for generating tests/golden/*.npz."""
import numpy as np

from oracle import targets as otargets, train as otrain


def samtron_config(desired_samples, reuse_ratio=0.0, initial_stepsize=0.1, adaptive=None, updater="trust-region",
                   weight_updater="trust-region", snis=True, own=False, wstep=1.0, estimator="Stein", diag=False,
                   selector="component-based", max_database_size=10000000):
    """SAMTRON-style config dict with the keys of the reference's example_config.yml."""
    cfg = {
        "temperature": 1.0, "use_sample_database": True, "max_database_size": max_database_size, "seed": 0,
        "model_initialization": {"use_diagonal_covs": bool(diag), "prior_mean": 0., "initial_cov": 1.0},
        "ng_estimator_type": estimator,
        "ng_estimator_config": dict({"only_use_own_samples": own, "use_self_normalized_importance_weights": snis},
                                    **({"initial_l2_regularizer": 1e-12} if estimator == "MORE" else {})),
        "sample_selector_type": selector,
        "sample_selector_config": {"desired_samples_per_component": desired_samples,
                                   "ratio_reused_samples_to_desired": reuse_ratio},
        "ng_based_updater_type": updater, "ng_based_updater_config": {},
        "component_stepsize_adapter_type": "improvement-based",
        "component_stepsize_adapter_config": {"initial_stepsize": initial_stepsize, "min_stepsize": 0.001,
                                              "max_stepsize": 1.0, "stepsize_inc_factor": 1.15,
                                              "stepsize_dec_factor": 0.85},
        "weight_stepsize_adapter_type": "improvement_based",
        "weight_stepsize_adapter_config": {"initial_stepsize": wstep, "min_stepsize": 0.0001, "max_stepsize": 1.0,
                                           "stepsize_inc_factor": 1.15, "stepsize_dec_factor": 0.85},
        "weight_updater_type": weight_updater,
        "weight_updater_config": {"use_self_normalized_importance_weights": snis},
    }
    if adaptive:
        cfg["num_component_adapter_type"] = "adaptive"
        cfg["num_component_adapter_config"] = dict(adaptive)
    else:
        cfg["num_component_adapter_type"] = "fixed"
        cfg["num_component_adapter_config"] = {}
    return cfg


def make_oracle_target(kind, d, seed):
    rng = np.random.default_rng(seed)
    if kind == "stm":
        return otargets.make_stm_target(d, rng)
    if kind == "gmm":
        return otargets.make_gmm_target(d, rng, num_components=4)
    if kind == "diaggmm":                      # target_distributions/diag_gmm.py:33-45 law, 4 components
        means = 100 * (rng.random((4, d)) - 0.5) * 0.2
        covs = 10 * rng.random((4, d)) + 0.5
        return otargets.GmmTarget(np.ones(4) / 4, means, [np.diag(c) for c in covs])
    if kind == "gauss":
        a = rng.normal(size=(d, d))
        return otargets.GmmTarget([1.0], [rng.normal(size=d) * 3], [a @ a.T + np.eye(d)])
    if kind == "planar":
        return otargets.PlanarRobotTarget(d, 4)
    raise ValueError(kind)


def init_params(kind, d, k, seed):
    """(prior_scale, initial_cov) following the reference's experiment configs, scaled for small tests."""
    if kind == "planar":
        return [1.0] + [0.2] * (d - 1), [0.0625] + [0.0025] * (d - 1)
    if kind == "stm":
        return 10.0, 30.0
    if kind == "gmm":
        return 30.0, 100.0
    if kind == "diaggmm":
        return 10.0, 20.0
    return 5.0, 10.0


def make_oracle(kind, d, k, s, seed, cfg, dtype=np.float64, init=None):
    """``init`` = (prior_scale, initial_cov) of an experiment config where the scaled-down test defaults are not wanted."""
    tgt = make_oracle_target(kind, d, seed)
    ps, ic = init if init is not None else init_params(kind, d, k, seed)
    model = otrain.construct_initial_mixture(d, k, 0.0, ps, ic, np.random.default_rng(seed + 1), dtype=dtype,
                                             use_diagonal_covs=cfg["model_initialization"]["use_diagonal_covs"])
    algo = otrain.OracleGMMVI(
        tgt, model, temperature=cfg["temperature"], seed=seed,
        desired_samples_per_component=cfg["sample_selector_config"]["desired_samples_per_component"],
        ratio_reused_samples_to_desired=cfg["sample_selector_config"]["ratio_reused_samples_to_desired"],
        ng_estimator=cfg["ng_estimator_type"],
        only_use_own_samples=cfg["ng_estimator_config"]["only_use_own_samples"],
        use_self_normalized_importance_weights=cfg["ng_estimator_config"]["use_self_normalized_importance_weights"],
        updater=cfg["ng_based_updater_type"] if cfg["ng_based_updater_type"] != "iBLR" else "iblr",
        component_stepsize_config=cfg["component_stepsize_adapter_config"],
        weight_updater=cfg["weight_updater_type"],
        weight_stepsize_config=cfg["weight_stepsize_adapter_config"],
        adaptive=(dict(cfg["num_component_adapter_config"], prior_mean=0.0, initial_cov=ic)
                  if cfg["num_component_adapter_type"] == "adaptive" else None),
        max_reward_history_length=400, sample_selector=cfg["sample_selector_type"],
        max_database_size=cfg["max_database_size"],
        host_rng=np.random.default_rng(seed))
    return algo


def make_device(kind, d, k, s, seed, cfg, oracle_algo, init=None):
    """Device GMMVI initialised with exactly the oracle's target / initial mixture / seed."""
    from gmmvi_amd.models.full_cov_gmm import FullCovGMM
    from gmmvi_amd.models.gmm_wrapper import GmmWrapper
    from gmmvi_amd.optimization.gmmvi import GMMVI
    from gmmvi_amd.experiments.target_distributions.gmm import GMM_LNPDF
    from gmmvi_amd.experiments.target_distributions.student_t_mixture import StudentTMixture_LNPDF
    from gmmvi_amd.experiments.target_distributions.planar_robot import PlanarRobot
    t = oracle_algo.target
    if kind == "stm":
        tgt = StudentTMixture_LNPDF(t.weights, t.means, t.covs, alpha=2)
    elif kind in ("gmm", "gauss"):
        tgt = GMM_LNPDF(t.weights, t.means, t.covs)
    elif kind == "diaggmm":
        from gmmvi_amd.experiments.target_distributions.diag_gmm import DIAGGMM_LNPDF
        tgt = DIAGGMM_LNPDF(t.weights, t.means, np.stack([np.diag(c) for c in t.covs]))
    else:
        tgt = PlanarRobot(d, 4)
    om = oracle_algo.model.model
    if om.diagonal_covs:
        from gmmvi_amd.models.diagonal_gmm import DiagonalGMM
        model = DiagonalGMM(om.weights, om.means.astype(np.float32), om.covs.astype(np.float32))
    else:
        model = FullCovGMM(om.weights, om.means.astype(np.float32), om.covs.astype(np.float32))
    model.seed = seed
    wrapper = GmmWrapper(model, cfg["component_stepsize_adapter_config"]["initial_stepsize"], 1e-12, 400)
    cfg = dict(cfg)
    ps, ic = init if init is not None else init_params(kind, d, k, seed)
    cfg["model_initialization"] = dict(cfg["model_initialization"], prior_mean=0.0, initial_cov=ic)
    g = GMMVI.build_from_config(cfg, tgt, wrapper)
    if cfg["num_component_adapter_type"] == "adaptive":
        g.num_component_adapter.rng = np.random.default_rng(seed)
    return g


# ---- long-horizon cases (tests/golden/make_long_golden.py, tests/test_hip_long_horizon.py) ---------------------------
EXAMPLE5_ADAPTIVE = {"del_iters": 100, "add_iters": 60, "max_components": 1000,            # examples/5...:16, adaptive.yml
                     "thresholds_for_add_heuristic": [5000., 1000., 500., 200., 100., 50.],
                     "min_weight_for_del_heuristic": 1e-6, "num_database_samples": 100000, "num_prior_samples": 0}

EXAMPLE6_ADAPTIVE = dict(EXAMPLE5_ADAPTIVE, del_iters=10, add_iters=1)                       # examples/6...:20

LONG_CASES = {
    # name: kind, D, K, samples / component, seed, iterations, checkpoint interval, adaptive config, (prior scale, initial cov)
    # BASELINE configs[1] ("C2": 20-D Student-t mixture, K = 50 fixed, N = 5000) with the stm20.yml initial mixture
    "c2": dict(kind="stm", d=20, k=50, s=100, seed=3, iters=120, every=10, adaptive=None, init=(100.0, 300.0)),
    # BASELINE configs[0] = examples/5_samtron_20D_student-T.py:13-30: K = 45 adaptive (add every 60, delete after 100),
    # 200 samples per component, reuse ratio 0, initial stepsize 0.1, weight stepsize 1
    "c1": dict(kind="stm", d=20, k=45, s=200, seed=5, iters=260, every=10, adaptive=EXAMPLE5_ADAPTIVE,
               init=(100.0, 300.0)),
    # the same run at the length the example states (examples/5_samtron_20D_student-T.py:30: 1501 iterations): the converged tail,
    # where the stepsizes saturate and the deletion rule sees flat reward histories.  Its first 260 iterations ARE long_c1; behind
    # them K / ids are compared in lock-step as long as they agree and statistically afterwards (test_hip_long_horizon.py)
    "c1_full": dict(kind="stm", d=20, k=45, s=200, seed=5, iters=1501, every=50, adaptive=EXAMPLE5_ADAPTIVE,
                    init=(100.0, 300.0), chaotic=True, lockstep=260),
    # BASELINE configs[3]'s example as it is run, examples/6_samtron_planar4.py:19-26: planar-4 target, 100 initial components
    # from the planar_robot_4.yml prior, a component ADDED EVERY iteration, deletions from iteration 11 on (del_iters 10),
    # 100 samples per component, reuse ratio 0, weight stepsize 5 (clipped by the rule's max_stepsize as upstream does)
    "c4": dict(kind="planar", d=10, k=100, s=100, seed=7, iters=140, every=10, adaptive=EXAMPLE6_ADAPTIVE,
               init=([1.0] + [0.2] * 9, [0.0625] + [0.0025] * 9), wstep=5.0, min_deleted=5, chaotic=True),
}


def long_case_config(case):
    return samtron_config(case["s"], initial_stepsize=0.1, adaptive=case["adaptive"], wstep=case.get("wstep", 1.0))


def make_long_oracle(case, dtype=np.float64):
    return make_oracle(case["kind"], case["d"], case["k"], case["s"], case["seed"], long_case_config(case), dtype=dtype,
                       init=case["init"])


def make_long_device(case, oracle_algo):
    return make_device(case["kind"], case["d"], case["k"], case["s"], case["seed"], long_case_config(case), oracle_algo,
                       init=case["init"])


def score_elbo(target, log_weights, means, chols, temperature=1.0, num_samples=20000, seed=12345):
    """ELBO of the mixture (log_weights, means, chols) as the runner defines it (gmmvi_runner.py:131-133: mean target
    log-density + temperature * entropy estimate, x ~ q) on a FIXED set of Philox draws, scored in fp64 by the oracle's
    model class -> (elbo, sigma_mc = standard error of that mean)."""
    from oracle import gmm as ogmm
    k, d = np.asarray(means).shape
    m = ogmm.FullCovGMM(np.ones(k) / k, np.asarray(means, np.float64), np.broadcast_to(np.eye(d), (k, d, d)))
    m.log_weights = np.asarray(log_weights, np.float64).copy()
    m.chol_cov = np.asarray(chols, np.float64).copy()
    x, _ = m.sample(num_samples, seed, 0)
    per = target.log_density(x) - temperature * m.log_density(x)
    return float(np.mean(per)), float(np.std(per) / np.sqrt(num_samples))
