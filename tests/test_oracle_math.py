"""Pins the CPU oracle (oracle/) against SciPy, finite differences and closed-form identities.
The reference ships no tests or golden vectors for this path, so these checks are what
anchors the restatement; everything GPU-side is then compared with the oracle."""
import numpy as np
import pytest
from scipy import stats
from scipy.special import logsumexp

from oracle import philox, gmm as ogmm, targets, sample_db as odb, stein, more, updaters, weights as oweights, \
    stepsizes, train


def fd_grad(f, x, h=1e-5):
    g = np.zeros_like(x)
    for j in range(x.shape[1]):
        e = np.zeros(x.shape[1]); e[j] = h
        g[:, j] = (f(x + e) - f(x - e)) / (2 * h)
    return g


def random_gmm(rng, k, d, dtype=np.float64, spread=3.0):
    means = rng.normal(size=(k, d)) * spread
    covs = []
    for _ in range(k):
        a = rng.normal(size=(d, d))
        covs.append(a @ a.T / d + 0.3 * np.eye(d))
    w = rng.random(k) + 0.1
    return ogmm.FullCovGMM(w / w.sum(), means, np.stack(covs), dtype=dtype)


# ---------------------------------------------------------------------------------------------- philox
def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    f = philox.philox4x32_10
    assert [hex(v) for v in f(np.zeros(4, np.uint32), np.zeros(2, np.uint32))] == \
        ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(v) for v in f(np.full(4, 0xffffffff, np.uint32), np.full(2, 0xffffffff, np.uint32))] == \
        ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(v) for v in f(np.array([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], np.uint32),
                              np.array([0xa4093822, 0x299f31d0], np.uint32))] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_philox_normals_moments_and_indexing():
    e = philox.normals(5, 0, 100000, 6)
    assert abs(e.mean()) < 0.01 and abs(e.std() - 1) < 0.01
    assert np.abs(np.corrcoef(e.T) - np.eye(6)).max() < 0.02
    # counter-based: a window equals the same rows of a longer draw
    np.testing.assert_array_equal(philox.normals(5, 1000, 10, 6), e[1000:1010])
    assert stats.kstest(e[:, 3], 'norm').pvalue > 1e-3


# ---------------------------------------------------------------------------------------------- model
def test_component_log_densities_vs_scipy(rng):
    m = random_gmm(rng, 4, 5)
    x = rng.normal(size=(50, 5)) * 3
    cld = m.component_log_densities(x)
    for i in range(4):
        ref = stats.multivariate_normal(m.means[i], m.covs[i]).logpdf(x)
        np.testing.assert_allclose(cld[i], ref, rtol=1e-10, atol=1e-10)
    ref = logsumexp(cld + np.log(m.weights)[:, None], axis=0)
    np.testing.assert_allclose(m.log_density(x), ref, rtol=1e-12)


def test_log_density_grad_vs_finite_differences(rng):
    m = random_gmm(rng, 3, 4)
    x = rng.normal(size=(20, 4)) * 2
    lq, g, cld = m.log_density_and_grad(x)
    np.testing.assert_allclose(g, fd_grad(m.log_density, x), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(lq, m.log_density(x))


def test_marginals_and_entropy(rng):
    m = random_gmm(rng, 3, 4)
    x = rng.normal(size=(10, 4))
    ref = logsumexp(np.stack([stats.norm(m.means[i, 2], np.sqrt(m.covs[i, 2, 2])).logpdf(x[:, 2])
                              for i in range(3)]) + m.log_weights[:, None], axis=0)
    np.testing.assert_allclose(m.marginal_log_density(x, 2), ref, rtol=1e-10)
    for i in range(3):
        np.testing.assert_allclose(m.component_entropies()[i],
                                   stats.multivariate_normal(m.means[i], m.covs[i]).entropy(), rtol=1e-10)


def test_sampling_law(rng):
    m = random_gmm(rng, 2, 3)
    eps = philox.normals(3, 0, 40000, 3)
    x, mapping = m.sample_from_components_no_shuffle([20000, 20000], eps)
    assert mapping[0] == 0 and mapping[-1] == 1 and mapping.shape == (40000,)
    for i in range(2):
        xs = x[mapping == i]
        assert np.abs(xs.mean(0) - m.means[i]).max() < 0.05
        assert np.abs(np.cov(xs.T) - m.covs[i]).max() < 0.08
    u = philox.uniform01(3, 0, 50000)
    comp = m.sample_categorical_from_uniform(u)
    assert abs((comp == 0).mean() - m.weights[0]) < 0.01


def test_wrapper_bookkeeping(rng):
    m = random_gmm(rng, 3, 2)
    w = ogmm.GmmWrapper(m, 0.5, 1e-12, 6)
    assert w.reward_history.shape == (3, 6) and w.reward_history[0, 0] == np.finfo(np.float32).min
    w.store_rewards(np.array([1., 2., 3.]))
    assert w.reward_history[1, -1] == 2. and w.reward_history[1, -2] == np.finfo(np.float32).min
    w.add_component(1e-29, np.zeros(2), np.eye(2), [5.0], [1.0])
    assert w.num_components == 4 and w.stepsizes[-1] == 0.5 and w.unique_component_ids[-1] == 3
    assert np.all(w.weight_history[-1] == 1e-29)
    np.testing.assert_allclose(np.exp(w.log_weights).sum(), 1.0)
    w.remove_component(1)
    assert w.num_components == 3 and list(w.unique_component_ids) == [0, 2, 3] and w.reward_history.shape == (3, 6)
    w.replace_weights(np.log([0.2, 0.3, 0.5]) + 7.0)
    np.testing.assert_allclose(w.weights, [0.2, 0.3, 0.5])
    np.testing.assert_allclose(w.weight_history[:, -1], [0.2, 0.3, 0.5])


# ---------------------------------------------------------------------------------------------- targets
def test_gmm_target_vs_scipy_and_fd(rng):
    t = targets.make_gmm_target(5, rng, num_components=3)
    x = t.means[rng.integers(0, 3, 30)] + rng.normal(size=(30, 5)) * 4
    ref = logsumexp(np.stack([stats.multivariate_normal(t.means[i], t.covs[i]).logpdf(x) for i in range(3)])
                    + np.log(1 / 3), axis=0)
    lp, g = t.log_density_and_grad(x)
    np.testing.assert_allclose(lp, ref, rtol=1e-10)
    np.testing.assert_allclose(g, fd_grad(t.log_density, x), rtol=1e-5, atol=1e-6)


def test_student_t_target_vs_scipy_and_fd(rng):
    t = targets.make_stm_target(6, rng)
    x = t.means[rng.integers(0, 10, 40)] + rng.normal(size=(40, 6)) * 2
    ref = logsumexp(np.stack([stats.multivariate_t(t.means[i], t.covs[i], df=2).logpdf(x) for i in range(10)])
                    + np.log(0.1), axis=0)
    lp, g = t.log_density_and_grad(x)
    np.testing.assert_allclose(lp, ref, rtol=1e-10)
    np.testing.assert_allclose(g, fd_grad(t.log_density, x, 1e-6), rtol=2e-5, atol=1e-5)


def test_planar_robot_target(rng):
    t = targets.PlanarRobotTarget(10, 4)
    th = rng.normal(size=(25, 10)) * t.prior_stds
    fk = t.forward_kinematics(th)
    # explicit loop of planar_robot.py:58-64
    for n in range(3):
        xs = sum(np.cos(th[n, :i + 1].sum()) for i in range(10))
        ys = sum(np.sin(th[n, :i + 1].sum()) for i in range(10))
        np.testing.assert_allclose(fk[n], [xs, ys], rtol=1e-12)
    prior = stats.multivariate_normal(np.zeros(10), np.diag(t.prior_stds ** 2)).logpdf(th)
    lik = np.max(np.stack([stats.multivariate_normal(g, 1e-4 * np.eye(2)).logpdf(fk) for g in t.goals]), axis=0)
    lp, g = t.log_density_and_grad(th)
    np.testing.assert_allclose(lp, prior + lik, rtol=1e-10)
    np.testing.assert_allclose(g, fd_grad(t.log_density, th, 1e-7), rtol=1e-4, atol=1e-2)
    # all zeros reaches (10, 0): nearest goal is (7, 0)
    lp0 = t.log_density(np.zeros((1, 10)))
    assert np.isclose(lp0[0], stats.multivariate_normal(np.zeros(10), np.diag(t.prior_stds ** 2)).logpdf(np.zeros(10))
                      + stats.multivariate_normal([7, 0], 1e-4 * np.eye(2)).logpdf([10, 0]))


# ---------------------------------------------------------------------------------------------- sample db
def test_sample_db_background_is_count_weighted_mixture(rng):
    m = random_gmm(rng, 3, 4)
    db = odb.SampleDB(4, False, True, None)
    n_k = [5, 9, 2]
    x, mapping = m.sample_from_components_no_shuffle(n_k, philox.normals(0, 0, 16, 4))
    db.add_samples(x, m.means, m.chol_cov, np.arange(16.), np.ones((16, 4)), mapping)
    bg, xs, mp, lp, gr = db.get_newest_samples(16)
    ref = logsumexp(m.component_log_densities(x) + np.log(np.array(n_k) / 16.)[:, None], axis=0)
    np.testing.assert_allclose(bg, ref, rtol=1e-10)
    np.testing.assert_array_equal(mp, mapping)
    # newest 4 samples: components 1 (2 samples) and 2 (2 samples)
    bg4, xs4, mp4, _, _ = db.get_newest_samples(4)
    ref4 = logsumexp(m.component_log_densities(xs4)[1:] + np.log([0.5, 0.5])[:, None], axis=0)
    np.testing.assert_allclose(bg4, ref4, rtol=1e-10)
    assert db.get_newest_samples(0)[1].shape == (0, 4)
    # second batch gets its own component snapshots (mapping offset, sample_db.py:115)
    db.add_samples(x, m.means + 1, m.chol_cov, np.arange(16.), np.ones((16, 4)), mapping)
    assert db.means.shape[0] == 6 and db.mapping.max() == 5 and db.num_samples_written == 32


def test_sample_db_halving(rng):
    m = random_gmm(rng, 2, 3)
    db = odb.SampleDB(3, False, True, max_samples=30)
    for it in range(3):
        x, mapping = m.sample_from_components_no_shuffle([6, 6], philox.normals(0, 12 * it, 12, 3))
        db.add_samples(x, m.means + it, m.chol_cov, np.full(12, float(it)), np.zeros((12, 3)), mapping)
    assert db.samples.shape[0] == 24 + 0 or db.samples.shape[0] == 24  # 12+12 then halve(24)->12, +12
    assert db.num_samples_written == 36
    assert db.mapping.max() + 1 == db.means.shape[0]
    # every stored sample's snapshot mean still belongs to the iteration that drew it
    its = db.target_lnpdfs.astype(int)
    np.testing.assert_allclose(db.means[db.mapping][:, 0] - m.means[db.mapping % 2][:, 0] * 0,
                               db.means[db.mapping][:, 0])
    assert set(np.unique(its)) <= {0, 1, 2}


def test_effective_sample_size(rng):
    sel = odb.VipsSampleSelector(None, None, None, 10, 0.)
    ld = np.zeros((2, 50)); bg = np.zeros(50)
    np.testing.assert_allclose(sel.get_effective_samples(ld, bg), [50., 50.])
    ld[1, 0] = 100.
    assert sel.get_effective_samples(ld, bg)[1] < 1.0001


# ---------------------------------------------------------------------------------------------- stein / more
def test_stein_identities_on_gaussian_target(rng):
    """Target N(m_t, S_t), single-component model N(mu, S): E_q[grad log p/q] = -S_t^-1 (mu - m_t)  (exact in the
    self-normalised estimator up to MC error) and E_q[Hess log p/q] = S^-1 - S_t^-1."""
    d = 3
    a = rng.normal(size=(d, d)); s_t = a @ a.T + np.eye(d); m_t = rng.normal(size=d)
    tgt = targets.GmmTarget([1.0], [m_t], [s_t])
    b = rng.normal(size=(d, d)); s = b @ b.T / d + 0.5 * np.eye(d); mu = rng.normal(size=d)
    m = ogmm.FullCovGMM([1.0], [mu], [s])
    n = 200000
    x, mapping = m.sample_from_components_no_shuffle([n], philox.normals(1, 0, n, d))
    tlp, tg = tgt.log_density_and_grad(x)
    bg = m.log_density(x)
    h_neg, g_neg = stein.get_expected_hessian_and_grad(m, x, mapping, bg, tlp, tg)
    np.testing.assert_allclose(-g_neg[0], -np.linalg.solve(s_t, mu - m_t), atol=0.03)
    np.testing.assert_allclose(-h_neg[0], np.linalg.inv(s) - np.linalg.inv(s_t), atol=0.05)
    # plain importance weights with bg == component density: all weights are exactly 1
    h2, g2 = stein.get_expected_hessian_and_grad(m, x, mapping, bg, tlp, tg, use_self_normalized_importance_weights=False)
    np.testing.assert_allclose(g2, g_neg, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(0.5 * (h2[0] + h2[0].T), h_neg[0], rtol=1e-8, atol=1e-10)


def test_stein_orientation_and_own_samples(rng):
    m = random_gmm(rng, 3, 4)
    n_k = [30, 30, 30]
    x, mapping = m.sample_from_components_no_shuffle(n_k, philox.normals(2, 0, 90, 4))
    tgt = targets.make_gmm_target(4, rng, 2)
    tlp, tg = tgt.log_density_and_grad(x)
    bg = logsumexp(m.component_log_densities(x) + np.log(1 / 3), axis=0)
    h_neg, g_neg = stein.get_expected_hessian_and_grad(m, x, mapping, bg, tlp, tg)
    # brute force for component 1
    lq, gq, cld = m.log_density_and_grad(x)
    w = np.exp(cld[1] - bg); w /= w.sum()
    g = tg - gq
    y = np.linalg.solve(m.covs[1], (x - m.means[1]).T).T
    a = sum(w[n] * np.outer(g[n], y[n]) for n in range(90))
    np.testing.assert_allclose(-h_neg[1], 0.5 * (a + a.T), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(-g_neg[1], (w[:, None] * g).sum(0), rtol=1e-9, atol=1e-12)
    h_own, g_own = stein.get_expected_hessian_and_grad(m, x, mapping, bg, tlp, tg, only_use_own_samples=True)
    own = mapping == 2
    np.testing.assert_allclose(-g_own[2], g[own].mean(0), rtol=1e-9)


def test_more_recovers_quadratic_reward(rng):
    """If log p~ - log q is exactly quadratic around a single Gaussian q, MORE's fit is exact:
    R = S_t^-1 - S^-1 (quad term of -(log p - log q) ... sign convention: returns G = -(d^2/dx^2)(log p/q))."""
    d = 3
    a = rng.normal(size=(d, d)); s_t = a @ a.T + np.eye(d); m_t = rng.normal(size=d)
    tgt = targets.GmmTarget([1.0], [m_t], [s_t])
    mu = rng.normal(size=d); s = 0.7 * np.eye(d)
    m = ogmm.FullCovGMM([1.0], [mu], [s])
    n = 400
    x, mapping = m.sample_from_components_no_shuffle([n], philox.normals(4, 0, n, d))
    bg = m.log_density(x)
    h_neg, g_neg = more.get_expected_hessian_and_grad(m, np.array([1e-12]), x, mapping, bg, tgt.log_density(x))
    np.testing.assert_allclose(h_neg[0], np.linalg.inv(s_t) - np.linalg.inv(s), rtol=1e-5, atol=1e-6)
    # expected_gradient_neg = -E[grad log p/q] evaluated through the quadratic model at the mean:
    grad_at_mean = -np.linalg.solve(s_t, mu - m_t)
    np.testing.assert_allclose(-g_neg[0], grad_at_mean, rtol=1e-5, atol=1e-6)
    f = more.quad_features(np.array([[1., 2., 3.]]))
    np.testing.assert_allclose(f[0], [1, 2, 3, 4, 6, 9, 1, 2, 3, 1])


# ---------------------------------------------------------------------------------------------- updaters
def gaussian_kl(m1, s1, m0, s0):
    d = len(m0)
    s0i = np.linalg.inv(s0)
    return 0.5 * (np.trace(s0i @ s1) + (m0 - m1) @ s0i @ (m0 - m1) - d
                  + np.linalg.slogdet(s0)[1] - np.linalg.slogdet(s1)[1])


def test_kl_formula_matches_closed_form(rng):
    d = 5
    m = random_gmm(rng, 1, d)
    l, mu = m.chol_cov[0], m.means[0]
    linv = np.linalg.inv(l); q = linv.T @ linv
    b = rng.normal(size=(d, d)); r = b @ b.T * 0.3
    g = rng.normal(size=d)
    rl = r @ mu - g
    for eta in [0.3, 2.0, 50.0]:
        val, new_mean, new_prec, cinv = updaters.kl(eta, q @ mu, q, linv, rl, r, 2 * np.sum(np.log(np.diag(l))) - d,
                                                    mu, False)
        np.testing.assert_allclose(val, gaussian_kl(new_mean, np.linalg.inv(new_prec), mu, m.covs[0]), rtol=1e-9)
        np.testing.assert_allclose(cinv.T @ cinv, np.linalg.inv(new_prec), rtol=1e-9)
        np.testing.assert_allclose(new_prec, q + r / eta, rtol=1e-12)
    # indefinite update -> failure value
    val = updaters.kl(1e-3, q @ mu, q, linv, rl, -r * 100, 0.0, mu, False)[0]
    assert val == np.finfo(np.float32).max


def test_kl_constrained_update_hits_bound(rng):
    m = random_gmm(rng, 4, 5)
    w = ogmm.GmmWrapper(m, 0.1, 1e-12, 4)
    old_means, old_covs = m.means.copy(), m.covs.copy()
    hs = np.stack([(lambda b: b @ b.T)(rng.normal(size=(5, 5))) for _ in range(4)])
    hs[3] = -hs[3] * 0.01                        # mildly indefinite: still solvable with a large eta
    gs = rng.normal(size=(4, 5))
    traces = []
    succ, etas, kls, nprobes = updaters.apply_ng_update_kl(w, hs, gs, w.stepsizes, 1.0, traces=traces)
    assert succ.all() and (nprobes > 0).all()
    for i in range(4):
        val = gaussian_kl(m.means[i], m.covs[i], old_means[i], old_covs[i])
        assert val < 0.1 * 1.1 + 1e-9
        np.testing.assert_allclose(val, kls[i], rtol=1e-6)
        assert etas[i] >= 1.0
    np.testing.assert_array_equal(w.last_log_etas, etas)           # stores eta, not log(eta)
    np.testing.assert_allclose(w.num_received_updates, 1)
    # warm start brackets around the previous eta
    succ2, etas2, _, nprobes2 = updaters.apply_ng_update_kl(w, hs, gs, w.stepsizes, 1.0, traces=[])
    assert succ2.all()


def test_kl_update_failure_path(rng):
    m = random_gmm(rng, 2, 3)
    w = ogmm.GmmWrapper(m, 0.1, 1e-12, 4)
    old = (m.means.copy(), m.chol_cov.copy())
    hs = np.stack([np.full((3, 3), np.nan), np.eye(3)])
    gs = np.zeros((2, 3))
    succ, etas, kls, _ = updaters.apply_ng_update_kl(w, hs, gs, w.stepsizes, 1.0, traces=[])
    assert list(succ) == [False, True]
    np.testing.assert_array_equal(m.means[0], old[0][0]); np.testing.assert_array_equal(m.chol_cov[0], old[1][0])
    assert etas[0] == -1 and w.last_log_etas[0] == -1
    np.testing.assert_allclose(w.l2_regularizers, [min(1e-6, 10 * 1e-12), max(0.5e-12, 1e-12)])


def test_direct_and_iblr_updates(rng):
    m = random_gmm(rng, 2, 3)
    w = ogmm.GmmWrapper(m, 0.01, 1e-12, 4)
    old_means, old_covs = m.means.copy(), m.covs.copy()
    hs = np.stack([np.eye(3) * 0.5, np.eye(3) * 2.0]); gs = rng.normal(size=(2, 3))
    succ = updaters.apply_ng_update_direct(w, hs, gs, np.array([0.5, 0.5]))
    assert succ.all()
    for i in range(2):
        p = np.linalg.inv(old_covs[i]) + 0.5 * hs[i]
        np.testing.assert_allclose(m.covs[i], np.linalg.inv(p), rtol=1e-9)
        lin = np.linalg.solve(old_covs[i], old_means[i]) + 0.5 * (hs[i] @ old_means[i] - gs[i])
        np.testing.assert_allclose(m.means[i], np.linalg.solve(p, lin), rtol=1e-9)
    m2 = random_gmm(rng, 1, 3)
    w2 = ogmm.GmmWrapper(m2, 0.01, 1e-12, 4)
    mu0, s0 = m2.means[0].copy(), m2.covs[0].copy()
    succ = updaters.apply_ng_update_iblr(w2, hs[:1], gs[:1], np.array([0.2]))
    np.testing.assert_array_equal(m2.means[0], mu0)                       # first update: mean untouched
    p = np.linalg.inv(s0) + 0.2 * (hs[0] + 0.1 * hs[0] @ s0 @ hs[0])
    np.testing.assert_allclose(m2.covs[0], np.linalg.inv(p), rtol=1e-9)
    s1 = m2.covs[0].copy()
    updaters.apply_ng_update_iblr(w2, hs[:1], gs[:1], np.array([0.2]))
    np.testing.assert_allclose(m2.means[0], mu0 - 0.2 * s1 @ gs[0], rtol=1e-9)


# ---------------------------------------------------------------------------------------------- weights / stepsizes
def test_weight_trust_region(rng):
    k = 6
    lw = np.log(rng.dirichlet(np.ones(k)))
    elr = rng.normal(size=k) * 3
    for eps in [0.01, 0.1, 1.0]:
        val, eta, nl = oweights.weights_bracketing_search(lw, elr, eps, 1.0)
        assert eta > 0
        np.testing.assert_allclose(np.exp(nl).sum(), 1.0)
        true_kl = np.sum(np.exp(nl) * (nl - lw))
        assert true_kl < 1.1 * eps + 1e-12
    # huge bound: upper bracket never violated -> eta = exp(-45..): essentially greedy weights
    val, eta, nl = oweights.weights_bracketing_search(lw, elr, 1e6, 1.0)
    greedy = (lw * 0 + elr) - logsumexp(elr)
    assert np.argmax(nl) == np.argmax(greedy)
    # floor at exp(-69.07)
    _, nl = oweights.weights_kl(1e-9, lw, np.array([0, -1000., 0, 0, 0, 0]), 1.0)
    assert nl[1] > -69.08


def test_expected_log_ratios_and_rewards(rng):
    m = random_gmm(rng, 3, 3)
    w = ogmm.GmmWrapper(m, 0.1, 1e-12, 5)
    x, mapping = m.sample_from_components_no_shuffle([20, 20, 20], philox.normals(6, 0, 60, 3))
    bg = logsumexp(m.component_log_densities(x) + np.log(1 / 3), axis=0)
    tlp = rng.normal(size=60)
    elr = oweights.get_expected_log_ratios(w, x, bg, tlp, 1.0)
    cld = m.component_log_densities(x)
    for i in range(3):
        iw = np.exp(cld[i] - bg); iw /= iw.sum()
        np.testing.assert_allclose(elr[i], iw @ (tlp - m.log_density(x)), rtol=1e-10)
    np.testing.assert_allclose(w.reward_history[:, -1], m.log_weights + elr)


def test_stepsize_rules():
    rh = np.array([[1., 2.], [2., 1.], [3., 3.]])
    out = stepsizes.component_stepsize_improvement(np.array([0.5, 0.5, 0.002]), rh, 0.001, 1.0, 1.15, 0.85)
    np.testing.assert_allclose(out, [0.575, 0.425, 0.0017])
    np.testing.assert_allclose(stepsizes.component_stepsize_decaying(np.array([0., 1., 4.]), 1.0, 0.5), [1, .5, 1 / 3])

    class W:
        weights = np.array([0.5, 0.5]); log_weights = np.log(weights)
        reward_history = np.full((2, 3), np.finfo(np.float32).min)
    a = stepsizes.WeightStepsizeImprovement(1.0, 1e-4, 1.0, 1.15, 0.85)
    assert a.update_stepsize(W) == 0.85                    # sentinel absorbs the entropy: "not greater"
    W.reward_history = np.array([[0., -3.], [0., -2.]])
    np.testing.assert_allclose(a.update_stepsize(W), 0.85 * 1.15)
    W.reward_history = np.array([[0., -30.], [0., -20.]])
    np.testing.assert_allclose(a.update_stepsize(W), 0.85 * 1.15 * 0.85)


# ---------------------------------------------------------------------------------------------- end to end
def test_single_gaussian_known_answer():
    """SAMTRON on a single Gaussian target converges to its mean/covariance and ELBO -> log Z = 0."""
    rng = np.random.default_rng(0)
    d = 4
    a = rng.normal(size=(d, d)); cov = a @ a.T + np.eye(d); mean = rng.normal(size=d) * 3
    tgt = targets.GmmTarget([1.0], [mean], [cov])
    model = train.construct_initial_mixture(d, 1, 0., 5., 10., np.random.default_rng(1))
    algo = train.OracleGMMVI(tgt, model, seed=2, desired_samples_per_component=300,
                             component_stepsize_config=dict(initial_stepsize=0.1))
    for _ in range(80):
        algo.train_iter()
    np.testing.assert_allclose(algo.model.means[0], mean, atol=2e-3)
    np.testing.assert_allclose(algo.model.covs[0], cov, rtol=2e-3, atol=2e-3)
    assert abs(algo.elbo(4000, 99)[0]) < 1e-3


def test_elbo_nondecreasing_on_stm():
    rng = np.random.default_rng(0)
    tgt = targets.make_stm_target(4, rng)
    model = train.construct_initial_mixture(4, 4, 0., 10., 30., np.random.default_rng(1))
    algo = train.OracleGMMVI(tgt, model, seed=3, desired_samples_per_component=80,
                             component_stepsize_config=dict(initial_stepsize=0.1))
    e0 = algo.elbo(4000, 7)[0]
    for _ in range(40):
        algo.train_iter()
    e1 = algo.elbo(4000, 7)[0]
    assert e1 > e0 + 1.0 and e1 <= 0.2        # normalised target: ELBO <= log Z = 0 (up to MC noise)


# ---- diagonal-covariance branches and MMD (models/diagonal_gmm.py, experiments/evaluation/mmd.py) --------------------

def _diag_and_full(rng, k, d):
    from oracle import gmm as og
    means = rng.normal(size=(k, d)) * 2
    var = rng.uniform(0.4, 2.5, size=(k, d))
    w = rng.random(k) + 0.1
    w /= w.sum()
    return og.DiagonalGMM(w, means, var), og.FullCovGMM(w, means, np.stack([np.diag(v) for v in var]))


def test_diagonal_gmm_matches_scipy_and_full_cov_oracle(rng):
    from scipy.stats import multivariate_normal
    dm, fm = _diag_and_full(rng, 4, 6)
    x = rng.normal(size=(50, 6)) * 2
    ld = dm.component_log_densities(x)
    for i in range(4):
        np.testing.assert_allclose(ld[i], multivariate_normal(dm.means[i], np.diag(dm.covs[i])).logpdf(x), rtol=1e-10)
    for a, b in zip(dm.log_density_and_grad(x), fm.log_density_and_grad(x)):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)
    eps = 1e-5                                              # central finite differences of the analytic gradient
    g = dm.log_density_and_grad(x)[1]
    for c in range(6):
        e = np.zeros(6); e[c] = eps
        np.testing.assert_allclose(g[:, c], (dm.log_density(x + e) - dm.log_density(x - e)) / (2 * eps), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dm.get_average_entropy(), fm.get_average_entropy(), rtol=1e-12)


@pytest.mark.parametrize("snis", [True, False])
def test_diagonal_stein_is_the_diagonal_of_the_full_estimate(rng, snis):
    """ng_estimator.py:159-162 / :178-181 against the full-covariance branch on L = diag(sigma)."""
    from oracle import stein
    dm, fm = _diag_and_full(rng, 3, 5)
    n = 300
    x = rng.normal(size=(n, 5)) * 2
    bg = dm.log_density(x) + 0.1 * rng.normal(size=n)
    tlp, tg = rng.normal(size=n), rng.normal(size=(n, 5))
    mp = np.sort(rng.integers(0, 3, n)).astype(np.int32)
    hd, gd = stein.get_expected_hessian_and_grad(dm, x, mp, bg, tlp, tg, False, snis)
    hf, gf = stein.get_expected_hessian_and_grad(fm, x, mp, bg, tlp, tg, False, snis)
    np.testing.assert_allclose(hd, np.stack([np.diag(h) for h in hf]), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gd, gf, rtol=1e-12)


def test_diagonal_kl_update_matches_closed_form_and_full_cov_updater(rng):
    from oracle import gmm as og, updaters
    dm, fm = _diag_and_full(rng, 4, 5)
    hs = rng.uniform(0.1, 1.0, size=(4, 5))
    gs = rng.normal(size=(4, 5))
    old_means, old_var = dm.means.copy(), dm.covs.copy()
    wd, wf = og.GmmWrapper(dm, 1.0, 1e-12, 4), og.GmmWrapper(fm, 1.0, 1e-12, 4)
    eps = np.array([0.01, 0.05, 0.2, 0.5])
    sd, etas_d, kls_d, _ = updaters.apply_ng_update_kl(wd, hs, gs, eps, 1.0, traces=[])
    sf, etas_f, kls_f, _ = updaters.apply_ng_update_kl(wf, np.stack([np.diag(h) for h in hs]), gs, eps, 1.0, traces=[])
    assert sd.all() and sf.all()
    np.testing.assert_allclose(etas_d, etas_f, rtol=1e-9)
    np.testing.assert_allclose(dm.means, fm.means, rtol=1e-9)
    np.testing.assert_allclose(dm.chol_cov, np.stack([np.diag(c) for c in fm.chol_cov]), rtol=1e-9)
    # the reported KL is the closed-form Gaussian KL(new || old) and respects the bound
    new_var = dm.covs
    kl = 0.5 * np.sum(np.log(old_var / new_var) + new_var / old_var - 1 + np.square(dm.means - old_means) / old_var, axis=1)
    np.testing.assert_allclose(kls_d, kl, rtol=1e-8, atol=1e-12)
    assert np.all(kls_d < 1.1 * eps)


def test_diagonal_iblr_matches_full_cov_updater(rng):
    from oracle import gmm as og, updaters
    dm, fm = _diag_and_full(rng, 3, 4)
    hs = rng.uniform(-0.2, 1.0, size=(3, 4))
    gs = rng.normal(size=(3, 4))
    wd, wf = og.GmmWrapper(dm, 1.0, 1e-12, 4), og.GmmWrapper(fm, 1.0, 1e-12, 4)
    for _ in range(2):
        sd = updaters.apply_ng_update_iblr(wd, hs, gs, np.full(3, 0.2))
        sf = updaters.apply_ng_update_iblr(wf, np.stack([np.diag(h) for h in hs]), gs, np.full(3, 0.2))
        np.testing.assert_array_equal(sd, sf)
        np.testing.assert_allclose(dm.means, fm.means, rtol=1e-9)
        np.testing.assert_allclose(dm.chol_cov, np.stack([np.diag(c) for c in fm.chol_cov]), rtol=1e-9)


def test_diagonal_sample_db_background_matches_full(rng):
    from oracle import sample_db as odb
    dm, fm = _diag_and_full(rng, 3, 4)
    x = rng.normal(size=(30, 4))
    mp = np.repeat(np.arange(3), 10).astype(np.int32)
    dbd, dbf = odb.SampleDB(4, True, True), odb.SampleDB(4, False, True)
    for db, m in ((dbd, dm), (dbf, fm)):
        db.add_samples(x, m.means, m.chol_cov, np.zeros(30), np.zeros((30, 4)), mp)
    assert dbd.chols.shape == (3, 4) and dbd.inv_chols.shape == (3, 4)
    np.testing.assert_allclose(dbd.get_newest_samples(30)[0], dbf.get_newest_samples(30)[0], rtol=1e-10)


def test_mmd_oracle_properties(rng):
    from oracle import mmd
    gt = rng.normal(size=(60, 3)) * np.array([0.5, 1.0, 2.0])
    sigma = mmd.compute_sigma(gt)
    # "nearest" median of the i <= j squared differences, against a direct sort
    iu, ju = np.triu_indices(60)
    g32 = gt.astype(np.float32)
    for c in range(3):
        col = np.sort(np.square(g32[iu, c] - g32[ju, c]))
        assert sigma[c, c] == col[int(np.round((len(col) - 1) * 0.5))]
    assert abs(mmd.compute_mmd(gt, gt, 10.0, sigma)) < 1e-12
    shifted = gt + 3.0
    assert mmd.compute_mmd(gt, shifted, 10.0, sigma) > mmd.compute_mmd(gt, gt + 0.1, 10.0, sigma) > 0
    # pair_sum against the dense formula
    a, b = rng.normal(size=(7, 3)), rng.normal(size=(5, 3))
    k = np.linalg.inv(10.0 * sigma)
    dense = sum(np.exp(-(x - y) @ k @ (x - y)) for x in a for y in b)
    np.testing.assert_allclose(mmd.pair_sum(a, b, k), dense, rtol=1e-12)
