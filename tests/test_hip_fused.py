"""The single-call iteration (gmmvi_train_iter_samtron, optimization/fused.py) must equal the module-by-module path:
same kernels, same order, same state arrays -> identical trajectories; and it must step aside whenever a module or
setting outside its scope is configured."""
import numpy as np
import pytest

from helpers import samtron_config, make_oracle, make_device

pytestmark = pytest.mark.gpu


def _pair(kind, d, k, s, seed, cfg):
    o = make_oracle(kind, d, k, s, seed, cfg)
    fast = make_device(kind, d, k, s, seed, cfg, o)
    slow = make_device(kind, d, k, s, seed, cfg, o)
    slow._fast_path.enabled = False
    return o, fast, slow


@pytest.mark.parametrize("kind,d,k,s,cfg", [
    ("stm", 4, 3, 32, samtron_config(32)),
    ("planar", 10, 4, 50, samtron_config(50)),
    ("gmm", 20, 8, 64, samtron_config(64, weight_updater="direct", wstep=0.05)),
    ("gmm", 3, 1, 40, samtron_config(40)),
    ("stm", 6, 5, 30, samtron_config(30, snis=False, initial_stepsize=0.01)),
    ("gmm", 32, 4, 80, samtron_config(80)),            # tiled Stein kernel + wide mixture_eval (what bench.py c3 composes)
    ("gmm", 50, 5, 100, samtron_config(100)),
    ("gmm", 32, 4, 80, samtron_config(80, snis=False, initial_stepsize=0.01)),   # plain weights: finalize launch + update
    # sample reuse (the reference's default selector, component-based.yml:3-4: ratio 2): data-dependent numbers of new samples
    ("stm", 4, 3, 32, samtron_config(32, reuse_ratio=2.0)),
    ("gmm", 20, 8, 64, samtron_config(64, reuse_ratio=2.0)),
    ("planar", 10, 4, 50, samtron_config(50, reuse_ratio=1.0)),
    ("gmm", 32, 4, 80, samtron_config(80, reuse_ratio=2.0)),
])
def test_fast_path_equals_modular_path(kind, d, k, s, cfg):
    o, fast, slow = _pair(kind, d, k, s, 23, cfg)
    assert fast._fast_path.eligible() and not slow._fast_path.eligible()
    fast._fast_path.explicit_estimate = True           # the estimate materialised as the modules do it: bit-equal
    for it in range(8):
        fast.train_iter()
        slow.train_iter()
        for name in ("means", "chol_cov", "log_weights", "stepsizes", "last_log_etas", "l2_regularizers",
                     "num_received_updates"):
            a, b = getattr(fast.model, name).numpy(), getattr(slow.model, name).numpy()
            np.testing.assert_array_equal(a, b, err_msg=f"iteration {it}: {name}")
        np.testing.assert_array_equal(fast.model.reward_slot(0).numpy(), slow.model.reward_slot(0).numpy())
        np.testing.assert_array_equal(fast.weight_stepsize_adapter._state.numpy(),
                                      slow.weight_stepsize_adapter._state.numpy())
    np.testing.assert_array_equal(fast.sample_db.samples.numpy(), slow.sample_db.samples.numpy())
    np.testing.assert_array_equal(fast.sample_db.mapping.numpy(), slow.sample_db.mapping.numpy())
    np.testing.assert_array_equal(fast.sample_db.means.numpy(), slow.sample_db.means.numpy())
    np.testing.assert_array_equal(fast.sample_db.target_grads.numpy(), slow.sample_db.target_grads.numpy())
    assert int(fast.sample_db.num_samples_written) == int(slow.sample_db.num_samples_written)
    assert int(fast.num_updates) == int(slow.num_updates) == 8
    np.testing.assert_array_equal(fast.model.weight_history[:, -3:], slow.model.weight_history[:, -3:])
    np.testing.assert_array_equal(fast.ng_based_updater.last_success.numpy(), slow.ng_based_updater.last_success.numpy())


@pytest.mark.parametrize("kind,d,k,s,cfg", [
    ("stm", 4, 3, 32, samtron_config(32)),
    ("planar", 10, 4, 50, samtron_config(50)),
    ("gmm", 20, 8, 64, samtron_config(64)),
    ("gmm", 32, 4, 80, samtron_config(80)),            # wide instances: dense L^-1 from the fragments, register-blocked products
    ("gmm", 50, 5, 100, samtron_config(100)),
])
def test_direct_whitening_of_the_moment_sums(kind, d, k, s, cfg):
    """Default single-call route at D = 4 / 10 / 20 / 32 / 40 / 50 with self-normalised weights: the update kernel forms
    M = -sym(L^T C L^-T) / sum e from the Stein moment sums instead of materialising H (csrc/update_kl.hip).  Same mathematics,
    fewer roundings: the first iteration (identical inputs: the only difference is how the whitened matrix M and the vector
    w are formed) agrees with the module-by-module path to 2e-5 of the parameter scale, every later iteration to a FIXED
    2e-4 (no growth with the iteration count: 260-iteration runs of this route stay within 3e-5 of the fp64 oracle,
    tests/test_hip_long_horizon.py), the accept / reject decisions are the same."""
    o, fast, slow = _pair(kind, d, k, s, 23, cfg)
    assert not fast._fast_path.explicit_estimate
    for it in range(6):
        fast.train_iter()
        slow.train_iter()
        tol = 2e-5 if it == 0 else 2e-4
        for name in ("means", "chol_cov", "log_weights"):
            a, b = getattr(fast.model, name).numpy(), getattr(slow.model, name).numpy()
            scale = max(1.0, float(np.abs(b).max()))
            np.testing.assert_allclose(a, b, rtol=tol, atol=tol * scale, err_msg=f"iteration {it}: {name}")
        np.testing.assert_array_equal(fast.ng_based_updater.last_success.numpy(), slow.ng_based_updater.last_success.numpy())
        np.testing.assert_array_equal(fast.model.num_received_updates.numpy(), slow.model.num_received_updates.numpy())


def test_fast_path_with_adaptive_components_and_oracle():
    ad = {"del_iters": 6, "add_iters": 3, "max_components": 6, "thresholds_for_add_heuristic": [50., 20., 10.],
          "min_weight_for_del_heuristic": 1e-6, "num_database_samples": 200, "num_prior_samples": 0}
    cfg = samtron_config(40, adaptive=ad)
    o, fast, slow = _pair("gmm", 3, 2, 40, 9, cfg)
    for it in range(12):
        o.train_iter(); fast.train_iter(); slow.train_iter()
    assert fast.model.num_components == slow.model.num_components == o.model.num_components > 2
    np.testing.assert_array_equal(fast.model.means.numpy(), slow.model.means.numpy())
    np.testing.assert_allclose(fast.model.means.numpy(), o.model.means, rtol=0.05, atol=0.05)


def test_default_samtron_config_takes_the_single_call_path():
    """get_default_algorithm_config("SAMTRON") as shipped (sample reuse ratio 2, adaptive number of components) through the
    runner: every iteration is eligible for the single-call path; the database thinning iteration alone falls back."""
    from gmmvi.gmmvi_runner import GmmviRunner
    from gmmvi.configs import update_config, get_default_experiment_config, get_default_algorithm_config
    config = update_config(update_config(get_default_experiment_config("stm20"), {"start_seed": 0}),
                           update_config(get_default_algorithm_config("SAMTRON"),
                                         {"model_initialization": {"num_initial_components": 6},
                                          "gmmvi_runner_config": {"log_metrics_interval": 100}}))
    assert config["sample_selector_config"]["ratio_reused_samples_to_desired"] == 2.0
    runner = GmmviRunner.build_from_config(config=config)
    fp = runner.gmmvi._fast_path
    for n in range(8):
        assert fp.eligible()
        runner.iterate_and_log(n)
    db = runner.gmmvi.sample_db
    assert int(db.num_samples_written) == db.samples.shape[0] > 6 * 100


def test_fast_path_with_reuse_and_database_thinning():
    """A small database limit: the iteration that has to thin the database out runs module by module, the others in one
    call; the trajectory equals the all-modular one bit for bit."""
    cfg = samtron_config(40, reuse_ratio=2.0, max_database_size=700)
    o, fast, slow = _pair("gmm", 4, 3, 40, 31, cfg)
    fast._fast_path.explicit_estimate = True
    took_fast = 0
    for it in range(14):
        took_fast += bool(fast._fast_path.eligible())
        fast.train_iter()
        slow.train_iter()
        np.testing.assert_array_equal(fast.model.means.numpy(), slow.model.means.numpy(), err_msg=f"iteration {it}")
        np.testing.assert_array_equal(fast.sample_db.mapping.numpy(), slow.sample_db.mapping.numpy())
    assert 0 < took_fast < 14
    np.testing.assert_array_equal(fast.sample_db.samples.numpy(), slow.sample_db.samples.numpy())


def test_fast_path_steps_aside():
    cfg = samtron_config(30, updater="direct", initial_stepsize=0.01)
    o, fast, _ = _pair("gmm", 4, 3, 30, 5, cfg)
    assert not fast._fast_path.eligible()
    cfg = samtron_config(30, own=True)
    o, fast, _ = _pair("gmm", 4, 3, 30, 5, cfg)
    assert not fast._fast_path.eligible()
    cfg = samtron_config(30)
    o, fast, _ = _pair("gmm", 4, 3, 30, 5, cfg)
    fast.ng_based_updater.want_info = True
    assert not fast._fast_path.eligible()
    fast.train_iter()                                            # and the modular path still runs



def test_early_draw_is_dropped_when_something_changes_in_between():
    """The single-call iteration draws the NEXT iteration's samples behind its component update (csrc/riders.h).  That draw
    must only be used if nothing touched the components, the sample counts or the database since: here the components are
    replaced, an iteration runs on the module-by-module path, and the desired samples per component change between
    single-call iterations -- the trajectories must stay bit-equal to a twin that never draws early."""
    cfg = samtron_config(48)
    o = make_oracle("stm", 6, 5, 48, 29, cfg)
    early = make_device("stm", 6, 5, 48, 29, cfg, o)
    plain = make_device("stm", 6, 5, 48, 29, cfg, o)
    plain._fast_path.presample = False
    for g in (early, plain):
        g._fast_path.explicit_estimate = True

    def both(fn):
        fn(early); fn(plain)

    def check(tag):
        for name in ("means", "chol_cov", "log_weights", "stepsizes"):
            np.testing.assert_array_equal(getattr(early.model, name).numpy(), getattr(plain.model, name).numpy(), err_msg=f"{tag}: {name}")
        np.testing.assert_array_equal(early.sample_db.samples.numpy(), plain.sample_db.samples.numpy(), err_msg=tag)

    for _ in range(3):
        both(lambda g: g.train_iter())
    assert early._fast_path._presample_token is not None and plain._fast_path._presample_token is None
    check("three single-call iterations")
    # (1) the components are replaced: the samples drawn early came from the old ones
    both(lambda g: g.model.replace_components(g.model.means.numpy() + 0.25, g.model.chol_cov.numpy()))
    both(lambda g: g.train_iter())
    check("after replace_components")
    # (2) an iteration on the module-by-module path in between
    def modular_iteration(g):
        g._fast_path.enabled = False
        g.train_iter()
        g._fast_path.enabled = True
    both(modular_iteration)
    both(lambda g: g.train_iter())
    check("after a module-by-module iteration")
    # (3) another number of samples per component
    def fewer(g):
        g.sample_selector.desired_samples_per_component = 40
    both(fewer)
    both(lambda g: g.train_iter())
    both(lambda g: g.train_iter())
    check("after changing the samples per component")


@pytest.mark.parametrize("reuse", [0.0, 2.0])
def test_database_move_into_a_mapped_range_does_not_change_the_run(reuse, monkeypatch):
    """The sample database's arrays move from a doubling allocation into a grow-in-place address range when they pass
    _Growable.MAPPED_FROM (1 GiB; here 256 KiB, so that it happens within a few iterations): base pointers change under the
    single-call iteration, the early draw and -- with reuse -- the prefetched window caches.  Same trajectory as without the move."""
    from gmmvi_amd.optimization import sample_db
    cfg = samtron_config(64, reuse_ratio=reuse)
    o = make_oracle("gmm", 20, 8, 64, 31, cfg)
    plain = make_device("gmm", 20, 8, 64, 31, cfg, o)
    for _ in range(14):
        plain.train_iter()
    assert plain.sample_db._samples._range is None
    monkeypatch.setattr(sample_db._Growable, "MAPPED_FROM", 256 << 10)
    monkeypatch.setattr(sample_db._Growable, "FIRST_APPENDS", 2)
    moved = make_device("gmm", 20, 8, 64, 31, cfg, o)
    for _ in range(14):
        moved.train_iter()
    assert moved.sample_db._samples._range is not None and moved.sample_db._target_grads._range is not None
    for name in ("means", "chol_cov", "log_weights", "stepsizes"):
        np.testing.assert_array_equal(getattr(moved.model, name).numpy(), getattr(plain.model, name).numpy(), err_msg=name)
    np.testing.assert_array_equal(moved.sample_db.samples.numpy(), plain.sample_db.samples.numpy())
    np.testing.assert_array_equal(moved.sample_db.target_lnpdfs.numpy(), plain.sample_db.target_lnpdfs.numpy())
