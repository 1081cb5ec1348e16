"""CPU suite (-m "not gpu"): oracle against the committed golden trajectories, host logic (config merge, alias
package), and the C-ABI library: loads and exports every symbol include/gmmvi_hip.h declares (no compute calls)."""
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from helpers import samtron_config, make_oracle  # noqa: E402

GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "samtron_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    kind, d, k, s, seed = str(g["kind"]), int(g["d"]), int(g["k"]), int(g["s"]), int(g["seed"])
    iters = min(int(g["iters"]), 8 if d > 10 else 20)
    o = make_oracle(kind, d, k, s, seed, samtron_config(s))
    np.testing.assert_allclose(o.model.means, g["init_means"], rtol=1e-12)
    for it in range(iters):
        info = o.train_iter()
        np.testing.assert_allclose(o.model.means, g["means"][it], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(o.model.chol_cov, g["chols"][it], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(o.model.log_weights, g["log_weights"][it], rtol=1e-7, atol=1e-9)
        np.testing.assert_array_equal(info["success"], g["success"][it])
        np.testing.assert_array_equal(info["n_probes"], g["n_probes"][it])
        np.testing.assert_allclose(o.model.stepsizes, g["stepsizes"][it], rtol=1e-12)


def test_golden_files_hold_data_only():
    for p in GOLDEN:
        g = np.load(p)
        assert {"init_means", "init_covs", "means", "chols", "log_weights", "success", "elbo"} <= set(g.files)
        assert os.path.getsize(p) < 1 << 20


# ---------------------------------------------------------------------------------------------- C ABI
@pytest.fixture(scope="session")
def built_lib():
    from gmmvi_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "gmmvi_amd", "csrc"), "-j8"], check=True)
    return _lib.load()


def test_library_exports_every_declared_symbol(built_lib):
    from gmmvi_amd import _lib
    header = open(os.path.join(ROOT, "include", "gmmvi_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(gmmvi_[a-z0-9_]+)\s*\(", header))
    declared -= {"gmmvi_ctx"}
    assert len(declared) > 40
    missing_binding = declared - set(_lib.EXPORTED_SYMBOLS)
    assert not missing_binding, f"declared in the header but not bound in _lib.py: {sorted(missing_binding)}"
    for name in declared:
        assert hasattr(built_lib, name), f"{name} is declared but not exported by libgmmvi_hip.so"
    extra = set(_lib.EXPORTED_SYMBOLS) - declared
    assert not extra, f"bound but not declared in the header: {sorted(extra)}"
    # argument-free query works without a GPU
    # padded D <= 24: [mu | 1/diag | triangle by rows | by columns | log-normaliser] + the sweep stream [mu, log-normaliser |
    # columns with their reciprocal diagonal entry | rows with theirs], every part a whole number of 16-byte words
    assert built_lib.gmmvi_packed_stride(20) == ((2 * 20 + 20 * 19 + 1 + 3) // 4) * 4 + 24 + 2 * ((210 + 3) // 4 * 4)
    assert built_lib.gmmvi_packed_stride(40) == 1664 + 64 * (4 + 8 + 10) + 64 * (10 + 6 + 2) + 44 + 2 * 820   # + L^-1 fragments
    assert built_lib.gmmvi_packed_stride(65) == 68 + 65 * 65          # blocked path: [mu, const, pad to 4 | L^-1]
    assert built_lib.gmmvi_packed_stride(300) == 304 + 300 * 300
    assert built_lib.gmmvi_packed_stride(513) == 0
    assert built_lib.gmmvi_device_count() >= 0


def test_context_creation_fails_loudly_without_gpu(built_lib):
    from gmmvi_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    from gmmvi_amd.device import Context
    with pytest.raises(_lib.GmmviError):
        Context(0)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under gmmvi_amd/ or gmmvi/ may import it."""
    for pkg in ("gmmvi_amd", "gmmvi"):
        for path in glob.glob(os.path.join(ROOT, pkg, "**", "*.py"), recursive=True):
            src = open(path).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path


# ---------------------------------------------------------------------------------------------- host logic
def test_config_merge_semantics():
    from gmmvi_amd.configs import get_default_algorithm_config, get_default_experiment_config, update_config, \
        get_default_config
    c = get_default_algorithm_config("SAMTRON")
    assert c["ng_estimator_type"] == "Stein" and c["num_component_adapter_type"] == "adaptive"
    assert c["sample_selector_type"] == "component-based" and c["ng_based_updater_type"] == "trust-region"
    assert c["component_stepsize_adapter_type"] == "improvement-based"          # hyphen
    assert c["weight_stepsize_adapter_type"] == "improvement_based"             # underscore
    assert c["weight_updater_type"] == "trust-region"
    assert c["sample_selector_config"] == {"desired_samples_per_component": 100, "ratio_reused_samples_to_desired": 2.}
    assert c["num_component_adapter_config"]["thresholds_for_add_heuristic"] == [5000., 1000., 500., 200., 100., 50.]
    e = get_default_experiment_config("stm20")
    assert e["model_initialization"]["num_initial_components"] == 20 and e["environment_config"]["num_dimensions"] == 20
    u = update_config(e, {"model_initialization": {"num_initial_components": 45}, "start_seed": 3})
    assert u["model_initialization"] == dict(e["model_initialization"], num_initial_components=45) and u["start_seed"] == 3
    assert e["model_initialization"]["num_initial_components"] == 20                # inputs untouched
    u2 = update_config(c, {"num_component_adapter_config": {"thresholds_for_add_heuristic": [1.0]}})
    assert u2["num_component_adapter_config"]["thresholds_for_add_heuristic"] == [1.0]     # lists are REPLACED
    full = get_default_config("SEPYFUX", "planar_robot_4")
    assert full["ng_based_updater_type"] == "iBLR" and full["environment_name"] == "PlanarRobot4"
    assert len(full["model_initialization"]["prior_scale"]) == 10
    with pytest.raises(KeyError):
        get_default_algorithm_config("SAMTRO?")


def test_alias_package_maps_reference_module_paths():
    import gmmvi  # noqa: F401
    import gmmvi.gmmvi_runner as a
    import gmmvi_amd.gmmvi_runner as b
    assert a is b
    from gmmvi.optimization.gmmvi_modules.ng_based_component_updater import NgBasedComponentUpdater
    from gmmvi.optimization.gmmvi_modules.weight_updater import WeightUpdater
    from gmmvi.optimization.gmmvi_modules.ng_estimator import NgEstimator
    from gmmvi.optimization.gmmvi_modules.sample_selector import SampleSelector
    from gmmvi.optimization.gmmvi_modules.component_adaptation import ComponentAdaptation
    from gmmvi.optimization.gmmvi_modules.component_stepsize_adaptation import ComponentStepsizeAdaptation
    from gmmvi.optimization.gmmvi_modules.weight_stepsize_adaptation import WeightStepsizeAdaptation
    from gmmvi.experiments.target_distributions.lnpdf import LNPDF
    for cls, key in ((NgBasedComponentUpdater, "ng_based_updater_type"), (WeightUpdater, "weight_updater_type"),
                     (NgEstimator, "ng_estimator_type"), (SampleSelector, "sample_selector_type")):
        with pytest.raises(ValueError):
            args = ({key: "bogus", "temperature": 1.0}, None) if cls is not NgEstimator else \
                ({key: "bogus"}, 1.0, None)
            if cls is SampleSelector:
                args = ({key: "bogus"}, None, None, None)
            cls.build_from_config(*args)
    assert LNPDF(use_log_density_and_grad=True, safe_for_tf_graph=False).use_log_density_and_grad


def test_blocked_threshold_knob(monkeypatch):
    """gmmvi_amd._lib.blocked_above() mirrors csrc/blocked.h gmmvi_blocked_above(): default 50, atoi parsing, clamped to 16..64,
    read once per process (as the library's static is)."""
    from gmmvi_amd import _lib
    monkeypatch.setattr(_lib, "_blocked_above", None)
    monkeypatch.delenv("GMMVI_BLOCKED_ABOVE", raising=False)
    assert _lib.blocked_above() == _lib.BLOCKED_ABOVE_DEFAULT == 50
    monkeypatch.setenv("GMMVI_BLOCKED_ABOVE", "32")
    assert _lib.blocked_above() == 50                  # cached: the two sides cannot disagree within one process
    for raw, want in (("32", 32), ("5", 16), ("200", 64), ("junk", 16), ("40x", 40)):
        monkeypatch.setattr(_lib, "_blocked_above", None)
        monkeypatch.setenv("GMMVI_BLOCKED_ABOVE", raw)
        assert _lib.blocked_above() == want
    monkeypatch.setattr(_lib, "_blocked_above", None)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gmmvi_amd", "csrc", "blocked.h")).read()
    assert "atoi(s) : 50" in src                       # the C side carries the same default


def test_alias_maps_diagonal_and_mmd_modules():
    """The reference's module paths for the diagonal model, its target and the MMD metric resolve through the alias."""
    import importlib
    for name in ("gmmvi.models.diagonal_gmm", "gmmvi.experiments.evaluation.mmd",
                 "gmmvi.experiments.target_distributions.diag_gmm"):
        mod = importlib.import_module(name)
        assert mod.__name__.startswith("gmmvi")
    from gmmvi.models.diagonal_gmm import DiagonalGMM
    from gmmvi.experiments.evaluation.mmd import MMD
    assert {"compute_MMD", "set_alpha", "compute_ustat", "kernel_mix", "compute_sigma"} <= set(dir(MMD))
    assert {"component_log_densities", "add_component", "gaussian_entropy", "covs"} <= set(dir(DiagonalGMM))


# ---------------------------------------------------------------------------------------------- bench.py self-launch
def _bench_module():
    import importlib
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_bench_spawns_its_own_ranks(tmp_path, capsys):
    """`python bench.py --gpus N` without a launcher starts N fresh rank processes with the launcher's environment
    variables, relays rank 0's JSON line and fails when any rank fails.  Stub child: no GPU involved."""
    bench = _bench_module()
    child = tmp_path / "child.py"
    child.write_text(
        "import json, os, sys\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert int(os.environ['MASTER_PORT']) > 0 and os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'\n"
        "open(os.path.join(sys.argv[1], f'rank{r}'), 'w').write(' '.join(sys.argv[2:]))\n"
        "if '--fail' in sys.argv and r == w - 1:\n"
        "    sys.exit(7)\n"
        "print(json.dumps({'rank': r, 'world': w}))\n")
    rc = bench.spawn_ranks(3, [str(tmp_path), "--gpus", "3", "--steps", "5"], child=str(child), timeout=60)
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0 and out == ['{"rank": 0, "world": 3}']                  # only rank 0's line reaches stdout
    for r in range(3):
        assert (tmp_path / f"rank{r}").read_text() == "--gpus 3 --steps 5"
    rc = bench.spawn_ranks(2, [str(tmp_path), "--fail"], child=str(child), timeout=60)
    assert rc == 7 and capsys.readouterr().out == ""


def test_bench_launcher_ends_the_job_when_one_rank_dies(tmp_path, capsys):
    """Rank 0 parked forever (as inside an RCCL call whose peer is gone), rank 1 exits 7 with a message on stderr: the
    launcher must return 7 within seconds, terminate rank 0 and relay the failing rank's stderr."""
    import time
    bench = _bench_module()
    child = tmp_path / "hang.py"
    child.write_text(
        "import os, sys, time\n"
        "if os.environ['RANK'] == '0':\n"
        "    open(os.path.join(sys.argv[1], 'pid0'), 'w').write(str(os.getpid()))\n"
        "    time.sleep(3600)\n"
        "time.sleep(0.5)\n"
        "sys.stderr.write('rank 1: ncclCommInitRank failed (stub)\\n')\n"
        "sys.exit(7)\n")
    t0 = time.time()
    rc = bench.spawn_ranks(2, [str(tmp_path)], child=str(child), timeout=120, log_dir=str(tmp_path))
    took = time.time() - t0
    cap = capsys.readouterr()
    assert rc == 7 and took < 30 and cap.out == ""
    assert "rank 1 exited with code 7" in cap.err and "ncclCommInitRank failed (stub)" in cap.err
    pid0 = int((tmp_path / "pid0").read_text())
    with pytest.raises(ProcessLookupError):
        os.kill(pid0, 0)                                                   # rank 0 is gone (terminated and reaped)
    # a job nobody finishes ends at the timeout with 124
    child.write_text("import time\ntime.sleep(3600)\n")
    assert bench.spawn_ranks(2, [], child=str(child), timeout=1.0, log_dir=str(tmp_path)) == 124


def test_bench_main_delegates_before_touching_the_gpu(monkeypatch):
    """With --gpus N > 1 and no WORLD_SIZE, main() must hand over to spawn_ranks before any device call."""
    bench = _bench_module()
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.setattr(bench, "spawn_ranks", lambda n, argv, **kw: seen.update(n=n, argv=list(argv)) or 0)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and seen == {"n": 4, "argv": ["--gpus", "4", "--steps", "3"]}


def test_bench_workload_spec_is_host_only():
    bench = _bench_module()
    w = bench.spec("ns", 1)
    assert (w["k_total"], w["d"], w["n_total"]) == (100, 20, 10000)
    w8 = bench.spec("ns", 8)
    assert (w8["k_total"], w8["s"]) == (800, 13)
    o = bench.make_oracle(bench.spec("tiny", 1), dtype=np.float32)
    o.train_iter()
    assert o.model.means.dtype == np.float32


def test_random_subset_is_a_uniform_subset_and_the_same_in_oracle_and_product():
    """The candidate draw of the add heuristic (sample_db.py:137-152: shuffle + slice): `random_subset` must return a strictly
    ascending set of distinct indices of the requested size, consume the generator identically in the oracle and in the product
    (the two are compared sample for sample in the adaptive tests), and pick every index with the same probability."""
    import numpy as np
    from oracle.sample_db import random_subset as oracle_subset
    from gmmvi_amd.optimization.sample_db import random_subset as product_subset
    for total, size in [(3_000_000, 100_000), (500, 300), (1000, 200), (10 ** 6, 2 * 10 ** 5), (50, 100), (10 ** 5, 10), (7, 7)]:
        r1, r2 = np.random.default_rng(5), np.random.default_rng(5)
        a, b = oracle_subset(r1, total, size), product_subset(r2, total, size)
        assert a.dtype == np.int32 and np.array_equal(a, b)
        assert a.shape[0] == min(size, total) and a.min() >= 0 and a.max() < total
        assert a.shape[0] < 2 or np.all(np.diff(a) > 0)
        assert r1.random() == r2.random()                      # same generator state behind the draw
    rng, hits, trials = np.random.default_rng(1), np.zeros(2000), 3000
    for _ in range(trials):
        hits[product_subset(rng, 2000, 100)] += 1              # sparse branch (4 * size <= total)
    p = 100 / 2000
    assert abs(hits.mean() - trials * p) < 1e-9
    assert abs(hits.std() - np.sqrt(trials * p * (1 - p))) < 0.15 * np.sqrt(trials * p * (1 - p))
    assert hits.min() > trials * p - 6 * np.sqrt(trials * p) and hits.max() < trials * p + 6 * np.sqrt(trials * p)
