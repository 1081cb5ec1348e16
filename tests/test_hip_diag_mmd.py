"""GPU parity of the diagonal-covariance GMM path (models/diagonal_gmm.py and the `diagonal_covs` branches of the
estimator, updaters, SampleDB) and of the MMD pair sums (experiments/evaluation/mmd.py) against the fp64 oracle.
Tolerances as in test_hip_kernels.py (kernel outputs rtol 1e-4 / atol 1e-5 unless stated)."""
import numpy as np
import pytest

from oracle import gmm as ogmm, stein as ostein, updaters as oupd, mmd as ommd
from helpers import samtron_config, make_oracle, make_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from gmmvi_amd.device import get_context
    return get_context()


def random_diag_gmm(rng, k, d, dtype=np.float64):
    means = rng.normal(size=(k, d)) * 3.0
    var = rng.uniform(0.3, 3.0, size=(k, d))
    w = rng.random(k) + 0.1
    return ogmm.DiagonalGMM(w / w.sum(), means, var, dtype=dtype)


def device_diag(ctx, m):
    from gmmvi_amd.models.diagonal_gmm import DiagonalGMM
    return DiagonalGMM(m.weights, m.means.astype(np.float32), m.covs.astype(np.float32), ctx=ctx)


@pytest.mark.parametrize("k,d,n", [(3, 4, 64), (8, 20, 512), (1, 2, 5), (5, 33, 200), (4, 64, 100), (3, 100, 90)])
def test_diag_densities_grad_sampling(ctx, rng, k, d, n):
    m = random_diag_gmm(rng, k, d)
    g = device_diag(ctx, m)
    assert g.diagonal_covs and g.chol_cov.shape == (k, d)
    x = (m.means[rng.integers(0, k, n)] + rng.normal(size=(n, d)) * 1.5).astype(np.float32)
    lp, grad, ld = g.log_density_and_grad(ctx.asarray(x))
    olp, ograd, old = m.log_density_and_grad(x.astype(np.float64))
    np.testing.assert_allclose(ld.numpy(), old, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(lp.numpy(), olp, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(grad.numpy(), ograd, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(g.component_log_densities(ctx.asarray(x)).numpy(), old, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(g.covs, m.covs, rtol=1e-6)
    np.testing.assert_allclose(g.get_average_entropy(), m.get_average_entropy(), rtol=1e-5)
    # sampling with supplied normals: x = mu + sigma * eps
    n_k = rng.integers(0, 7, k)
    eps = rng.normal(size=(int(n_k.sum()), d)).astype(np.float32)
    xs, mp = g.sample_from_components_no_shuffle(n_k, eps=eps)
    oxs, omp = m.sample_from_components_no_shuffle(n_k, eps.astype(np.float64))
    np.testing.assert_array_equal(mp.numpy(), omp)
    np.testing.assert_allclose(xs.numpy(), oxs, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("k,d,n", [(40, 20, 300), (3, 300, 130), (2, 512, 70), (70, 33, 1000), (64, 5, 2000), (9, 96, 257)])
def test_diag_sweep_kernels(ctx, rng, k, d, n):
    """The dedicated O(D)-per-pair kernels (csrc/diag_sweep.hip) through their own entry points: component chunks with a
    partial merge (K = 40 / 64 / 70 on few tiles), dimensions of several 32-float pieces with a ragged last piece (33, 96, 300,
    512), the dual sweep (second set of weights over the same components), sampling from the device Philox stream."""
    from gmmvi_amd import hip_ops
    from oracle import philox
    m = random_diag_gmm(rng, k, d)
    x = (m.means[rng.integers(0, k, n)] + rng.normal(size=(n, d)) * 1.5).astype(np.float32)
    means, sigma = ctx.asarray(m.means), ctx.asarray(m.chol_cov)
    logw = ctx.asarray(m.log_weights)
    packed = hip_ops.diag_pack(ctx, means, sigma)
    assert packed.shape == (k, hip_ops.diag_packed_stride(d))
    xd = ctx.asarray(x)
    ld, lp, grad = hip_ops.diag_mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_lp=True, want_grad=True)
    olp, ograd, old = m.log_density_and_grad(x.astype(np.float64))
    np.testing.assert_allclose(ld.numpy(), old, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(lp.numpy(), olp, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(grad.numpy(), ograd, rtol=1e-3, atol=1e-3)
    # gradient without asking for the log densities (internal scratch), log values alone
    _, lp_b, grad_b = hip_ops.diag_mixture_eval(ctx, packed, logw, xd, d, want_ld=False, want_lp=False, want_grad=True)
    assert lp_b is None
    np.testing.assert_array_equal(grad_b.numpy(), grad.numpy())
    np.testing.assert_array_equal(hip_ops.diag_mixture_eval(ctx, packed, logw, xd, d)[1].numpy(), lp.numpy())
    # dual sweep
    w2 = rng.dirichlet(np.ones(k))
    logw2 = ctx.asarray(np.log(w2).astype(np.float32))
    ld2, lp2a, grad2, lp2b = hip_ops.diag_mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_grad=True, logw2=logw2)
    np.testing.assert_array_equal(lp2a.numpy(), lp.numpy())
    np.testing.assert_array_equal(grad2.numpy(), grad.numpy())
    from scipy.special import logsumexp
    np.testing.assert_allclose(lp2b.numpy(), logsumexp(old + np.log(w2)[:, None], axis=0), rtol=1e-4, atol=1e-3)
    # sampling: device Philox stream == oracle Philox stream, x = mu + sigma * eps
    n_k = rng.integers(0, 9, k)
    ns = int(n_k.sum())
    offs = ctx.asarray(np.concatenate([[0], np.cumsum(n_k)]).astype(np.int32), np.int32)
    xs, mp = hip_ops.diag_sample(ctx, means, sigma, offs, ns, seed=3, first_index=100)
    eps = philox.normals(3, 100, ns, d)
    oxs, omp = m.sample_from_components_no_shuffle(n_k, eps)
    np.testing.assert_array_equal(mp.numpy(), omp)
    np.testing.assert_allclose(xs.numpy(), oxs, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k,d,n", [(5, 20, 600), (3, 70, 500)])
@pytest.mark.parametrize("snis", [True, False])
def test_diag_stein_own_samples(ctx, rng, k, d, n, snis):
    """only_use_own_samples on a diagonal model (ng_estimator.py:110-118 with the diagonal branches :159-162, :178-181)."""
    from gmmvi_amd.models.gmm_wrapper import GmmWrapper
    from gmmvi_amd.optimization.gmmvi_modules.ng_estimator import SteinNgEstimator
    m = random_diag_gmm(rng, k, d)
    g = GmmWrapper(device_diag(ctx, m), 0.1, 1e-12, 4)
    mp = np.sort(rng.integers(0, k, n)).astype(np.int32)
    mp[-1] = k - 1                                                     # the newest sample belongs to the newest component
    x = (m.means[mp] + rng.normal(size=(n, d)) * 1.2).astype(np.float32)
    x64 = x.astype(np.float64)
    bg = m.log_density(x64).astype(np.float32)
    tlp = rng.normal(size=n).astype(np.float32)
    tg = rng.normal(size=(n, d)).astype(np.float32)
    est = SteinNgEstimator(1.0, g, only_use_own_samples=True, use_self_normalized_importance_weights=snis)
    h, gr = est.get_expected_hessian_and_grad(ctx.asarray(x), ctx.asarray(mp, np.int32), ctx.asarray(bg), ctx.asarray(tlp),
                                              ctx.asarray(tg))
    oh, og = ostein.get_expected_hessian_and_grad(m, x64, mp, bg.astype(np.float64), tlp.astype(np.float64),
                                                  tg.astype(np.float64), True, snis)
    np.testing.assert_allclose(h.numpy(), oh, rtol=2e-3, atol=2e-4 * max(1.0, np.abs(oh).max()))
    np.testing.assert_allclose(gr.numpy(), og, rtol=2e-3, atol=2e-4 * max(1.0, np.abs(og).max()))


def test_diag_embed_extract_roundtrip(ctx, rng):
    from gmmvi_amd import hip_ops
    a = rng.normal(size=(5, 37)).astype(np.float32)
    dense = hip_ops.diag_embed(ctx, ctx.asarray(a))
    host = dense.numpy()
    np.testing.assert_array_equal(host, np.stack([np.diag(r) for r in a]))
    np.testing.assert_array_equal(hip_ops.diag_extract(ctx, dense).numpy(), a)
    np.testing.assert_allclose(hip_ops.reciprocal(ctx, ctx.asarray(a)).numpy(), 1.0 / a, rtol=1e-6)


@pytest.mark.parametrize("k,d,n", [(3, 4, 200), (8, 20, 1000), (4, 40, 300), (3, 70, 400)])
@pytest.mark.parametrize("snis", [True, False])
def test_diag_stein(ctx, rng, k, d, n, snis):
    from gmmvi_amd.models.gmm_wrapper import GmmWrapper
    from gmmvi_amd.optimization.gmmvi_modules.ng_estimator import SteinNgEstimator
    m = random_diag_gmm(rng, k, d)
    g = GmmWrapper(device_diag(ctx, m), 0.1, 1e-12, 4)
    x = (m.means[rng.integers(0, k, n)] + rng.normal(size=(n, d)) * 1.2).astype(np.float32)
    x64 = x.astype(np.float64)
    bg = (m.log_density(x64) + 0.1 * rng.normal(size=n)).astype(np.float32)
    tlp = rng.normal(size=n).astype(np.float32)
    tg = rng.normal(size=(n, d)).astype(np.float32)
    mp = np.sort(rng.integers(0, k, n)).astype(np.int32)
    est = SteinNgEstimator(1.0, g, only_use_own_samples=False, use_self_normalized_importance_weights=snis)
    h, gr = est.get_expected_hessian_and_grad(ctx.asarray(x), ctx.asarray(mp, np.int32), ctx.asarray(bg), ctx.asarray(tlp),
                                              ctx.asarray(tg))
    oh, og = ostein.get_expected_hessian_and_grad(m, x64, mp, bg.astype(np.float64), tlp.astype(np.float64),
                                                  tg.astype(np.float64), False, snis)
    assert h.shape == (k, d) and oh.shape == (k, d)
    scale = max(1.0, np.abs(oh).max())
    np.testing.assert_allclose(h.numpy(), oh, rtol=2e-3, atol=2e-4 * scale)
    np.testing.assert_allclose(gr.numpy(), og, rtol=2e-3, atol=2e-4 * max(1.0, np.abs(og).max()))


def _diag_update_inputs(rng, k, d):
    m = random_diag_gmm(rng, k, d)
    hs = rng.normal(size=(k, d)) * 0.5 + 0.3          # mixed signs: some new precisions go negative at small eta
    gs = rng.normal(size=(k, d))
    return m, hs, gs


@pytest.mark.parametrize("k,d", [(3, 4), (8, 20), (1, 2), (4, 64), (3, 65), (2, 300), (2, 512)])
def test_diag_update_kl(ctx, rng, k, d):
    from gmmvi_amd import hip_ops
    m, hs, gs = _diag_update_inputs(rng, k, d)
    m32 = ogmm.DiagonalGMM(m.weights, m.means.astype(np.float32), m.covs.astype(np.float32))
    w = ogmm.GmmWrapper(m32, 0.1, 1e-12, 4)
    w.stepsizes = np.linspace(0.05, 0.5, k)
    means, chols = ctx.asarray(m32.means), ctx.asarray(m32.chol_cov)
    last_eta = ctx.asarray(w.last_log_etas); l2 = ctx.asarray(w.l2_regularizers)
    nupd = ctx.asarray(w.num_received_updates); steps = ctx.asarray(w.stepsizes)
    for round_ in range(2):                          # cold bracket, then warm start
        succ, kl, probes = hip_ops.update_components_diag(ctx, "kl", means, chols, ctx.asarray(hs), ctx.asarray(gs), steps,
                                                          1.0, 1e-12, last_eta, l2, nupd, want_info=True)
        rs, retas, rkls, rprobes = oupd.apply_ng_update_kl(w, hs, gs, w.stepsizes, 1.0, traces=[])
        np.testing.assert_array_equal(succ.numpy().astype(bool), rs)
        np.testing.assert_array_equal(probes.numpy(), rprobes)          # same bisection path
        np.testing.assert_allclose(last_eta.numpy(), retas, rtol=1e-5)
        np.testing.assert_allclose(kl.numpy(), rkls, rtol=5e-3, atol=1e-5 * d)
        np.testing.assert_allclose(means.numpy(), m32.means, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(chols.numpy(), m32.chol_cov, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(l2.numpy(), w.l2_regularizers, rtol=1e-6)
        np.testing.assert_allclose(nupd.numpy(), w.num_received_updates)


def test_diag_update_kl_failure(ctx, rng):
    from gmmvi_amd import hip_ops
    k, d = 3, 6
    m, hs, gs = _diag_update_inputs(rng, k, d)
    hs[0] = np.nan
    hs[1] = -np.inf                                   # negative precision at every eta of the bracket (NaN KL)
    w = ogmm.GmmWrapper(m, 0.1, 1e-12, 4)
    means, chols = ctx.asarray(m.means), ctx.asarray(m.chol_cov)
    old_means, old_chols = means.numpy(), chols.numpy()
    last_eta = ctx.asarray(w.last_log_etas); l2 = ctx.asarray(w.l2_regularizers); nupd = ctx.asarray(w.num_received_updates)
    succ, _, _ = hip_ops.update_components_diag(ctx, "kl", means, chols, ctx.asarray(hs), ctx.asarray(gs),
                                                ctx.asarray(w.stepsizes), 1.0, 1e-12, last_eta, l2, nupd, want_info=True)
    rs, retas, _, _ = oupd.apply_ng_update_kl(w, hs, gs, w.stepsizes, 1.0, traces=[])
    np.testing.assert_array_equal(succ.numpy().astype(bool), rs)
    assert not rs[0] and not rs[1] and rs[2]
    np.testing.assert_array_equal(means.numpy()[:2], old_means[:2])
    np.testing.assert_array_equal(chols.numpy()[:2], old_chols[:2])
    np.testing.assert_allclose(last_eta.numpy(), retas, rtol=1e-4)
    np.testing.assert_allclose(l2.numpy(), w.l2_regularizers, rtol=1e-6)


def test_diag_update_iblr(ctx, rng):
    from gmmvi_amd import hip_ops
    k, d = 5, 70
    m, hs, gs = _diag_update_inputs(rng, k, d)
    hs[2] = np.nan                                    # NaN chol -> rejected (:202); finite inputs never fail here
    w = ogmm.GmmWrapper(m, 0.1, 1e-12, 4)
    steps = np.full(k, 0.3)
    means, chols = ctx.asarray(m.means), ctx.asarray(m.chol_cov)
    l2 = ctx.asarray(w.l2_regularizers); nupd = ctx.asarray(w.num_received_updates)
    for round_ in range(2):                           # the first update leaves the means alone (:184-186)
        succ, _, _ = hip_ops.update_components_diag(ctx, "iblr", means, chols, ctx.asarray(hs), ctx.asarray(gs),
                                                    ctx.asarray(steps), 0.0, 1e-12, None, l2, nupd)
        rs = oupd.apply_ng_update_iblr(w, hs, gs, steps)
        np.testing.assert_array_equal(succ.numpy().astype(bool), rs)
        assert not rs[2] and rs[0]
        np.testing.assert_allclose(means.numpy(), m.means, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(chols.numpy(), m.chol_cov, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(l2.numpy(), w.l2_regularizers, rtol=1e-6)
        np.testing.assert_allclose(nupd.numpy(), w.num_received_updates)


def _run_pair(kind, d, k, s, iters, cfg, seed=11):
    o = make_oracle(kind, d, k, s, seed, cfg)
    g = make_device(kind, d, k, s, seed, cfg, o)
    assert g.model.diagonal_covs and g.sample_db.diagonal_covariances and not g._fast_path.eligible()
    for it in range(iters):
        info = o.train_iter()
        g.train_iter()
        om, gm = o.model, g.model
        assert gm.num_components == om.num_components
        tol = 2e-3 * (1 + it)
        assert gm.chol_cov.shape == om.chol_cov.shape
        assert np.abs(gm.means.numpy() - om.means).max() / max(1.0, np.abs(om.means).max()) <= tol, it
        assert np.abs(gm.chol_cov.numpy() - om.chol_cov).max() / np.abs(om.chol_cov).max() <= tol, it
        assert np.abs(np.exp(gm.log_weights.numpy()) - om.weights).max() <= tol, it
        if "success" in info and g.ng_based_updater.last_success is not None:
            np.testing.assert_array_equal(g.ng_based_updater.last_success.numpy().astype(bool), info["success"])
        np.testing.assert_allclose(gm.num_received_updates.numpy(), om.num_received_updates)
    return o, g


@pytest.mark.parametrize("kind,d,k,s,reuse", [("diaggmm", 6, 4, 40, 0.0), ("stm", 20, 8, 64, 0.0), ("planar", 10, 4, 50, 0.0),
                                              ("diaggmm", 5, 3, 30, 2.0)])
def test_diag_trajectory_matches_oracle(kind, d, k, s, reuse):
    """GMMVI.train_iter() with a DiagonalGMM (Stein + KL trust region) against the fp64 oracle on identical draws;
    reuse ratio 2 exercises the diagonal SampleDB (background density over [Kb, D] snapshots, ESS)."""
    cfg = samtron_config(s, reuse_ratio=reuse, diag=True)
    o, g = _run_pair(kind, d, k, s, 10, cfg)
    np.testing.assert_allclose(g.model.last_log_etas.numpy(), o.model.last_log_etas, rtol=5e-2, atol=1e-6)
    elbo_o = o.elbo(4000, seed=5)[0]
    o.model.model.means = g.model.means.numpy().astype(np.float64)
    o.model.model.chol_cov = g.model.chol_cov.numpy().astype(np.float64)
    o.model.model.log_weights = g.model.log_weights.numpy().astype(np.float64)
    elbo_g = o.elbo(4000, seed=5)[0]
    assert abs(elbo_g - elbo_o) < 1e-2 + 1e-3 * abs(elbo_o), (elbo_g, elbo_o)
    db, odb = g.sample_db, o.sample_db
    assert db.chols.shape == odb.chols.shape
    np.testing.assert_allclose(db.inv_chols.numpy(), odb.inv_chols, rtol=2e-2)


def test_diag_trajectory_iblr_and_adaptive():
    """iBLR updater on a diagonal model, with components added and deleted (component_adaptation.py:220-223)."""
    adaptive = dict(del_iters=6, add_iters=3, max_components=6, thresholds_for_add_heuristic=[50.0, 20.0, 10.0],
                    min_weight_for_del_heuristic=1e-6, num_database_samples=200, num_prior_samples=0)
    cfg = samtron_config(40, updater="iBLR", initial_stepsize=0.05, adaptive=adaptive, diag=True)
    o, g = _run_pair("diaggmm", 5, 3, 40, 10, cfg)
    assert g.model.num_components > 3


def test_diag_rejects_unsupported_modules(ctx, rng):
    from gmmvi_amd.models.gmm_wrapper import GmmWrapper
    from gmmvi_amd.optimization.gmmvi_modules.ng_estimator import MoreNgEstimator
    from gmmvi_amd.optimization.gmmvi_modules.ng_based_component_updater import DirectNgBasedComponentUpdater
    from gmmvi_amd.optimization.sample_db import SampleDB
    g = GmmWrapper(device_diag(ctx, random_diag_gmm(rng, 2, 3)), 0.1, 1e-12, 4)
    with pytest.raises(ValueError):
        MoreNgEstimator(1.0, g, False, 1e-12, True)
    with pytest.raises(NotImplementedError):
        DirectNgBasedComponentUpdater(g, 1.0).apply_NG_update(ctx.zeros((2, 3)), ctx.zeros((2, 3)), ctx.full((2,), 0.1))
    with pytest.raises(ValueError):
        SampleDB(3, False, True, ctx=ctx).evaluate_background(np.ones(2) / 2, g.means, g.chol_cov, None, ctx.zeros((4, 3)))


# ---- MMD ------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("na,nb,d", [(1, 1, 1), (70, 33, 3), (300, 257, 20), (129, 600, 45), (64, 256, 32)])
def test_mmd_pair_sum(ctx, rng, na, nb, d):
    from gmmvi_amd import hip_ops
    a = rng.normal(size=(na, d)).astype(np.float32)
    b = (rng.normal(size=(nb, d)) * 1.3 + 0.2).astype(np.float32)
    bw = rng.uniform(0.02, 0.3, d).astype(np.float32)
    got = hip_ops.mmd_pair_sum(ctx, ctx.asarray(a), ctx.asarray(b), ctx.asarray(bw))
    want = ommd.pair_sum(a, b, np.diag(bw.astype(np.float64)))
    np.testing.assert_allclose(got, want, rtol=2e-5)
    # symmetric in (a, b); the self sum counts the na unit diagonal terms
    np.testing.assert_allclose(hip_ops.mmd_pair_sum(ctx, ctx.asarray(b), ctx.asarray(a), ctx.asarray(bw)), want, rtol=2e-5)
    assert hip_ops.mmd_pair_sum(ctx, ctx.asarray(a), ctx.asarray(a), ctx.asarray(bw)) >= na - 1e-3


def test_mmd_class_matches_oracle(ctx, rng):
    from gmmvi_amd.experiments.evaluation.mmd import MMD
    d = 6
    gt = (rng.normal(size=(400, d)) * np.linspace(0.5, 3, d)).astype(np.float32)
    sample = (rng.normal(size=(350, d)) * np.linspace(0.6, 2.5, d) + 0.3).astype(np.float32)
    mmd = MMD(gt, 20.0, ctx=ctx)
    sigma = ommd.compute_sigma(gt)
    np.testing.assert_allclose(np.diag(mmd.sigma), np.diag(sigma), rtol=1e-6)
    got = mmd.compute_MMD(ctx.asarray(sample))
    want = ommd.compute_mmd(gt, sample, 20.0, sigma)
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-6)
    assert abs(mmd.compute_MMD(ctx.asarray(gt))) < 1e-6          # MMD(X, X) = 0
    mmd.set_alpha(5.0)
    np.testing.assert_allclose(mmd.compute_MMD(sample), ommd.compute_mmd(gt, sample, 5.0, sigma), rtol=1e-3, atol=1e-6)


def test_runner_diagonal_model_with_mmd(tmp_path):
    """GmmviRunner flow (configs -> runner -> iterate_and_log -> npz dumps) with a diagonal model
    (model_initialization.use_diagonal_covs), the DIAGGMM target of setup_experiment.py:71-73 and the MMD metric of
    gmmvi_runner.py:45-54,140-142; the default SAMTRON modules incl. sample reuse and component adaptation."""
    from gmmvi.gmmvi_runner import GmmviRunner
    from gmmvi.configs import update_config, get_default_experiment_config, get_default_algorithm_config
    from gmmvi.experiments.target_distributions.diag_gmm import make_target
    np.random.seed(3)
    groundtruth = make_target(6).sample(500)[0].numpy()          # the runner reseeds: same target below (seed 3)
    np.save(tmp_path / "gt.npy", groundtruth)
    algorithm_config = get_default_algorithm_config("SAMTRON")
    environment_config = update_config(get_default_experiment_config("gmm20"), {"start_seed": 3})
    used = {"environment_name": "DIAGGMM6", "environment_config": {"num_dimensions": 6},
            "model_initialization": {"use_diagonal_covs": True, "num_initial_components": 4, "prior_scale": 30.,
                                     "initial_cov": 100.},
            "num_component_adapter_config": {"del_iters": 100, "add_iters": 5},
            "sample_selector_config": {"desired_samples_per_component": 60},
            "gmmvi_runner_config": {"log_metrics_interval": 10},
            "mmd_evaluation_config": {"sample_dir": str(tmp_path / "gt.npy"), "alpha": 20.},
            "dump_gmm_path": str(tmp_path)}
    config = update_config(environment_config, update_config(algorithm_config, used))
    runner = GmmviRunner.build_from_config(config=config)
    assert runner.gmmvi.model.diagonal_covs and runner.gmmvi.sample_db.diagonal_covariances
    elbos, mmds = [], []
    for n in range(41):
        metrics = runner.iterate_and_log(n)
        runner.log_to_disk(n)
        if "-elbo" in metrics:
            elbos.append(-metrics["-elbo"])
            mmds.append(metrics["MMD:"])
    runner.finalize()
    assert runner.gmmvi.model.num_components > 4
    assert elbos[-1] > elbos[0] and all(e == e for e in elbos)
    assert mmds[-1] < mmds[0] and mmds[-1] >= -1e-6
    dump = np.load(str(next(tmp_path.glob("*/final_gmm_dump.npz"))))
    assert dump["covs"].shape == (runner.gmmvi.model.num_components, 6)
