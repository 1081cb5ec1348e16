"""GPU parity of every HIP kernel against the fp64 oracle (BASELINE.md section 4 tolerances: kernel outputs rtol 1e-4 /
atol 1e-5 unless a looser bound is stated with its reason).  All calls go through the C ABI (gmmvi_amd.hip_ops)."""
import numpy as np
import pytest
from scipy.special import logsumexp

from oracle import philox, gmm as ogmm, targets as otargets, stein as ostein, updaters as oupd, weights as oweights, \
    stepsizes as osteps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from gmmvi_amd.device import get_context
    return get_context()


def ops():
    from gmmvi_amd import hip_ops
    return hip_ops


def random_gmm(rng, k, d, spread=3.0, scale=1.0):
    means = rng.normal(size=(k, d)) * spread
    covs = []
    for _ in range(k):
        a = rng.normal(size=(d, d))
        covs.append(scale * (a @ a.T / d + 0.3 * np.eye(d)))
    w = rng.random(k) + 0.1
    return ogmm.FullCovGMM(w / w.sum(), means, np.stack(covs))


def upload_model(ctx, m):
    return (ctx.asarray(m.log_weights), ctx.asarray(m.means), ctx.asarray(m.chol_cov))


SHAPES = [(3, 4, 64), (1, 2, 5), (8, 20, 512), (5, 10, 200), (7, 3, 130), (20, 32, 300), (4, 50, 100), (3, 64, 70),
          (40, 20, 1000)]


def test_philox_bits_and_normals(ctx):
    u = ops().philox_uniforms(ctx, 12345, 7, 1000, stream_id=1).numpy()
    np.testing.assert_array_equal(u, philox.uniform01(12345, 7, 1000, 1, dtype=np.float32))   # bit-exact integers
    e = ops().philox_normals(ctx, 99, 1 << 33, 257, 7, stream_id=0).numpy()
    np.testing.assert_allclose(e, philox.normals(99, 1 << 33, 257, 7, 0), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("k,d,n", SHAPES)
def test_pack_and_inverse(ctx, rng, k, d, n):
    m = random_gmm(rng, k, d)
    _, means, chols = upload_model(ctx, m)
    packed, inv = ops().pack_components(ctx, means, chols, want_inverse=True)
    np.testing.assert_allclose(inv.numpy(), np.linalg.inv(m.chol_cov), rtol=2e-4, atol=2e-5)
    covs = ctx.asarray(m.covs)
    ch, ok = ops().cholesky(ctx, covs)
    assert ok.numpy().all()
    np.testing.assert_allclose(ch.numpy(), m.chol_cov, rtol=2e-4, atol=2e-5)
    bad = m.covs.copy(); bad[0] = -np.eye(d)
    ch, ok = ops().cholesky(ctx, ctx.asarray(bad))
    assert ok.numpy()[0] == 0 and np.isnan(ch.numpy()[0]).all() and ok.numpy()[1:].all()


def assert_parity(actual, desired, rtol, atol, what):
    """BASELINE.md section 4 form: |actual - desired| <= atol * scale + rtol * |desired| with scale = max(1, max |desired|):
    the absolute part covers elements that are small because larger terms cancelled (gradient components near a mode)."""
    desired = np.asarray(desired)
    scale = max(1.0, float(np.max(np.abs(desired)))) if desired.size else 1.0
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=atol * scale, err_msg=what)


@pytest.mark.parametrize("k,d,n", SHAPES)
def test_mixture_eval_gauss(ctx, rng, k, d, n):
    m = random_gmm(rng, k, d)
    x = m.means[rng.integers(0, k, n)] + rng.normal(size=(n, d)) * 1.5
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    ld, lp, grad = ops().mixture_eval(ctx, packed, logw, ctx.asarray(x), d, want_ld=True, want_lp=True, want_grad=True)
    lq, g, cld = m.log_density_and_grad(x.astype(np.float32).astype(np.float64))
    # rtol 1e-4 / atol 1e-5 -- the kernels hold a ten times tighter relative bound on the log densities
    assert_parity(ld.numpy(), cld, 1e-5, 1e-6, "component log densities")
    assert_parity(lp.numpy(), lq, 1e-5, 1e-6, "mixture log density")
    assert_parity(grad.numpy(), g, 1e-4, 1e-5, "gradient")
    # no-grad variant gives the same densities
    _, lp2, _ = ops().mixture_eval(ctx, packed, logw, ctx.asarray(x), d)
    np.testing.assert_allclose(lp2.numpy(), lp.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("k,d,n", [(6, 1, 70), (5, 7, 300), (9, 8, 257), (11, 11, 400), (6, 12, 129), (7, 13, 333),
                                   (12, 16, 500), (5, 17, 260), (10, 23, 300), (9, 24, 300), (64, 20, 4000), (100, 10, 3000)])
@pytest.mark.parametrize("family", ["gauss", "student_t"])
def test_mixture_eval_every_padded_dimension(ctx, rng, k, d, n, family):
    """The scalar-fed sweep with software-pipelined block loads (csrc/subst_phased.h) at every padded dimension it is
    instantiated for (2, 4, 8, 10, 12, 16, 20, 24: one to nine 32-float pieces per substitution, pieces that start in the
    middle of a 16-byte word), dimensions below their padding (7 in 8, 11 in 12, 13 in 16, 17 in 20, 23 in 24), ragged last
    tiles, and component counts that split the sweep into chunks (K = 64 / 100: partial merge)."""
    m = random_gmm(rng, k, d)
    x = m.means[rng.integers(0, k, n)] + rng.normal(size=(n, d)) * 1.5
    logw, means, chols = upload_model(ctx, m)
    if family == "gauss":
        packed, _ = ops().pack_components(ctx, means, chols)
        ld, lp, grad = ops().mixture_eval(ctx, packed, logw, ctx.asarray(x), d, want_ld=True, want_lp=True, want_grad=True)
        lq, g, cld = m.log_density_and_grad(x.astype(np.float32).astype(np.float64))
    else:
        from gmmvi_amd import _lib
        t = otargets.StudentTMixtureTarget(m.weights, m.means, m.covs, 2.0)
        packed, _ = ops().pack_components(ctx, means, chols, family=_lib.STUDENT_T, nu=2.0)
        ld, lp, grad = ops().mixture_eval(ctx, packed, logw, ctx.asarray(x), d, family=_lib.STUDENT_T, nu=2.0, want_ld=True,
                                          want_lp=True, want_grad=True)
        xs = x.astype(np.float32).astype(np.float64)
        lq, g = t.log_density_and_grad(xs)
        cld = None
    if cld is not None:
        assert_parity(ld.numpy(), cld, 1e-5, 1e-6, "component log densities")
    assert_parity(lp.numpy(), lq, 1e-5, 1e-6, "mixture log density")
    assert_parity(grad.numpy(), g, 1e-4, 1e-5, "gradient")
    # the dual sweep (second set of weights over the same components) and the sweep without the gradient agree with it
    if family == "gauss":
        logw2 = ctx.asarray(np.log(rng.dirichlet(np.ones(k))).astype(np.float32))
        ld_d, lp_d, grad_d, lp2_d = ops().mixture_eval_dual(ctx, packed, logw, logw2, ctx.asarray(x), d)
        np.testing.assert_array_equal(lp_d.numpy(), lp.numpy())
        np.testing.assert_array_equal(grad_d.numpy(), grad.numpy())
        _, lp2_ref, _ = ops().mixture_eval(ctx, packed, logw2, ctx.asarray(x), d)
        np.testing.assert_allclose(lp2_d.numpy(), lp2_ref.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("k,d,n", [(20, 50, 2304), (17, 40, 2100), (16, 32, 2049), (33, 45, 2600)])
def test_mixture_eval_workgroup_shared_blocks(ctx, rng, k, d, n):
    """K >= 16 and N >= 2048 with the gradient on the matrix-core dimensions: the eight waves of a workgroup share a component
    block through LDS (density.hip mixture_eval_mfma_ws): ragged last tile, chunk merge, padded dimension (45 in 50)."""
    m = random_gmm(rng, k, d)
    x = m.means[rng.integers(0, k, n)] + rng.normal(size=(n, d)) * 1.5
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    ld, lp, grad = ops().mixture_eval(ctx, packed, logw, ctx.asarray(x), d, want_ld=True, want_lp=True, want_grad=True)
    lq, g, cld = m.log_density_and_grad(x.astype(np.float32).astype(np.float64))
    assert_parity(ld.numpy(), cld, 1e-5, 1e-6, "component log densities")
    assert_parity(lp.numpy(), lq, 1e-5, 1e-6, "mixture log density")
    assert_parity(grad.numpy(), g, 1e-4, 1e-5, "gradient")
    _, lp2, _ = ops().mixture_eval(ctx, packed, logw, ctx.asarray(x), d)          # the sweep without the gradient: other kernel
    np.testing.assert_allclose(lp2.numpy(), lp.numpy(), rtol=1e-5, atol=1e-4)


def test_mixture_eval_far_samples_and_empty(ctx, rng):
    m = random_gmm(rng, 6, 5)
    x = rng.normal(size=(100, 5)) * 200                     # far tails: LSE must not underflow to -inf/NaN
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    _, lp, grad = ops().mixture_eval(ctx, packed, logw, ctx.asarray(x), 5, want_grad=True)
    lq, g, _ = m.log_density_and_grad(x.astype(np.float32).astype(np.float64))
    assert np.isfinite(lp.numpy()).all()
    np.testing.assert_allclose(lp.numpy(), lq, rtol=2e-4)
    np.testing.assert_allclose(grad.numpy(), g, rtol=2e-3, atol=1e-2)
    _, lp0, _ = ops().mixture_eval(ctx, packed, logw, ctx.empty((0, 5)), 5)
    assert lp0.shape == (0,)


@pytest.mark.parametrize("d,c,n", [(6, 10, 300), (20, 10, 1000), (3, 2, 65)])
def test_student_t_target(ctx, rng, d, c, n):
    t = otargets.make_stm_target(d, rng)
    x = t.means[rng.integers(0, t.means.shape[0], n)] + rng.normal(size=(n, d)) * 2
    from gmmvi_amd import _lib
    packed, _ = ops().pack_components(ctx, ctx.asarray(t.means), ctx.asarray(t.chols), family=_lib.STUDENT_T, nu=2.0)
    _, lp, grad = ops().mixture_eval(ctx, packed, ctx.asarray(t.log_weights), ctx.asarray(x), d, family=_lib.STUDENT_T,
                                     nu=2.0, want_grad=True)
    rlp, rg = t.log_density_and_grad(x.astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(lp.numpy(), rlp, rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(grad.numpy(), rg, rtol=2e-3, atol=2e-3)


def test_student_t_target_workgroup_shared_blocks(ctx, rng):
    """The harder Student-t mixture (20 components) at D = 32 with 2 100 samples: family switch of the workgroup-shared
    matrix-core sweep (K >= 16, N >= 2048, gradient)."""
    d, n = 32, 2100
    t = otargets.make_stm_target(d, rng, harder_setting=True)
    x = t.means[rng.integers(0, t.means.shape[0], n)] + rng.normal(size=(n, d)) * 2
    from gmmvi_amd import _lib
    packed, _ = ops().pack_components(ctx, ctx.asarray(t.means), ctx.asarray(t.chols), family=_lib.STUDENT_T, nu=2.0)
    _, lp, grad = ops().mixture_eval(ctx, packed, ctx.asarray(t.log_weights), ctx.asarray(x), d, family=_lib.STUDENT_T,
                                     nu=2.0, want_grad=True)
    rlp, rg = t.log_density_and_grad(x.astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(lp.numpy(), rlp, rtol=1e-4, atol=5e-4)
    np.testing.assert_allclose(grad.numpy(), rg, rtol=2e-3, atol=4e-3)


def test_planar_target(ctx, rng):
    t = otargets.PlanarRobotTarget(10, 4)
    th = rng.normal(size=(500, 10)) * t.prior_stds
    lp, grad = ops().target_planar(ctx, ctx.asarray(t.prior_stds), ctx.asarray(t.goals), t.likelihood_std,
                                   ctx.asarray(th))
    rlp, rg = t.log_density_and_grad(th.astype(np.float32).astype(np.float64))
    # log-likelihood values are O(1e5) (0.01 m goal std): relative tolerance only
    np.testing.assert_allclose(lp.numpy(), rlp, rtol=2e-5)
    np.testing.assert_allclose(grad.numpy(), rg, rtol=2e-3, atol=1.0)


@pytest.mark.parametrize("k,d,n", SHAPES[:6])
def test_sample_components(ctx, rng, k, d, n):
    m = random_gmm(rng, k, d)
    n_k = rng.multinomial(n, np.ones(k) / k)
    offs = np.concatenate([[0], np.cumsum(n_k)]).astype(np.int32)
    _, means, chols = upload_model(ctx, m)
    eps = philox.normals(3, 100, n, d)
    x, mp = ops().sample_components(ctx, means, chols, ctx.asarray(offs, np.int32), n, eps=ctx.asarray(eps))
    rx, rmp = m.sample_from_components_no_shuffle(n_k, eps.astype(np.float32).astype(np.float64))
    np.testing.assert_array_equal(mp.numpy(), rmp)
    np.testing.assert_allclose(x.numpy(), rx, rtol=1e-5, atol=1e-5)
    # device Philox stream == oracle Philox stream
    x2, _ = ops().sample_components(ctx, means, chols, ctx.asarray(offs, np.int32), n, seed=3, first_index=100)
    np.testing.assert_allclose(x2.numpy(), rx, rtol=1e-4, atol=1e-4)


def _stein_inputs(rng, k, d, n):
    m = random_gmm(rng, k, d)
    n_k = rng.multinomial(n, np.ones(k) / k)
    x, mapping = m.sample_from_components_no_shuffle(n_k, philox.normals(5, 0, n, d))
    x = x.astype(np.float32).astype(np.float64)
    tgt = otargets.make_gmm_target(d, rng, 3)
    tlp, tg = tgt.log_density_and_grad(x)
    tg = tg.astype(np.float32).astype(np.float64)
    cnt = np.maximum(n_k, 1e-9)
    bg = logsumexp(m.component_log_densities(x) + np.log(cnt / cnt.sum())[:, None], axis=0)
    return m, x, mapping, tlp, tg, bg


@pytest.mark.parametrize("k,d,n", [(3, 4, 64), (8, 20, 512), (5, 10, 700), (2, 2, 33), (6, 31, 300), (3, 40, 260),
                                   (40, 20, 3000), (4, 50, 300), (3, 45, 520), (4, 12, 300), (3, 15, 500), (3, 16, 260), (2, 23, 300)])   # d > 40: blocked contractions on rebuilt L^-1 blocks
@pytest.mark.parametrize("snis", [True, False])
def test_stein(ctx, rng, k, d, n, snis):
    m, x, mapping, tlp, tg, bg = _stein_inputs(rng, k, d, n)
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    xd = ctx.asarray(x)
    ld, lp, qg = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_grad=True)
    h, g = ops().stein(ctx, packed, xd, ld, qg, ctx.asarray(bg), ctx.asarray(tg), d, self_normalized=snis)
    rh, rg = ostein.get_expected_hessian_and_grad(m, x, mapping, bg, tlp, tg, False, snis)
    scale_h = np.abs(rh).max(axis=(1, 2), keepdims=True)
    scale_g = np.abs(rg).max(axis=1, keepdims=True)
    # importance weights exp(ld - bg) amplify the fp32 error of ld (abs ~1e-4): 1e-3 of the per-component magnitude
    assert np.abs(h.numpy() - rh).max() <= 2e-3 * scale_h.max() or np.all(np.abs(h.numpy() - rh) <= 2e-3 * scale_h)
    assert np.all(np.abs(h.numpy() - rh) <= 3e-3 * scale_h + 1e-6)
    assert np.all(np.abs(g.numpy() - rg) <= 3e-3 * scale_g + 1e-6)


@pytest.mark.parametrize("k,d,n", [(4, 6, 400), (3, 48, 600)])
@pytest.mark.parametrize("snis", [True, False])
def test_stein_own_samples(ctx, rng, k, d, n, snis):
    m, x, mapping, tlp, tg, bg = _stein_inputs(rng, k, d, n)
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    xd = ctx.asarray(x)
    ld, lp, qg = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_grad=True)
    mp = mapping + 7                                  # DB indices: offset so that max(mapping) -> K-1
    h, g = ops().stein(ctx, packed, xd, ld, qg, ctx.asarray(bg), ctx.asarray(tg), d,
                       mapping=ctx.asarray(mp, np.int32), map_offset=k - 1 - int(mp.max()), own_samples_only=True,
                       self_normalized=snis)
    # plain importance weights over the own samples: weights exp(0), divisor = number of own samples
    # (ng_estimator.py:110-118,146-152), Hessian not symmetrised
    rh, rg = ostein.get_expected_hessian_and_grad(m, x, mp, bg, tlp, tg, True, snis)
    np.testing.assert_allclose(h.numpy(), rh, rtol=2e-3, atol=2e-3 * np.abs(rh).max())
    np.testing.assert_allclose(g.numpy(), rg, rtol=2e-3, atol=2e-3 * np.abs(rg).max())


@pytest.mark.parametrize("k,d,n", [(3, 4, 400), (8, 20, 4000), (5, 10, 1500), (2, 2, 200), (4, 16, 2500), (3, 21, 3000),
                                   (1, 7, 640), (20, 20, 6000)])
@pytest.mark.parametrize("snis", [True, False])
def test_more(ctx, rng, k, d, n, snis):
    """gmmvi_more against the oracle's restatement of ng_estimator.py:296-376 / least_squares.py:126-191."""
    from oracle import more as omore
    m, x, mapping, tlp, tg, bg = _stein_inputs(rng, k, d, n)
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    xd = ctx.asarray(x)
    ld, lp, _ = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_lp=True)
    l2 = np.full(k, 1e-6)
    h, g = ops().more(ctx, packed, chols, xd, ld, lp, ctx.asarray(bg), ctx.asarray(tlp), ctx.asarray(l2), d,
                      self_normalized=snis)
    rh, rg = omore.get_expected_hessian_and_grad(m, l2, x, mapping, bg, tlp, False, snis)
    h, g = h.numpy(), g.numpy()
    assert np.all(np.isfinite(h)) and np.all(np.isfinite(g))
    scale_h = np.abs(rh).max(axis=(1, 2), keepdims=True)
    scale_g = np.abs(rg).max(axis=1, keepdims=True)
    # fp64 Gram + solve on fp32 inputs (z, weights, rewards carry ~1e-6 relative error from the fp32 density pass)
    assert np.all(np.abs(h - rh) <= 2e-3 * scale_h + 1e-5), np.abs(h - rh).max() / scale_h.max()
    assert np.all(np.abs(g - rg) <= 2e-3 * scale_g + 1e-5), np.abs(g - rg).max() / scale_g.max()


@pytest.mark.parametrize("k,d,n", [(2, 24, 2600), (2, 32, 4200), (2, 50, 9000), (1, 37, 5500)])
@pytest.mark.parametrize("snis", [True, False])
def test_more_beyond_the_register_resident_system(ctx, rng, k, d, n, snis):
    """D > 21: F + 1 = 326 ... 1 327 features -- the tiled fp64 Gram launch and the blocked fp64 Cholesky in global memory
    (csrc/more.hip, more_gram_big / more_solve_big) against the oracle, well-posed regime (N several times F), ridge 1e-6."""
    from oracle import more as omore
    m, x, mapping, tlp, tg, bg = _stein_inputs(rng, k, d, n)
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    xd = ctx.asarray(x)
    ld, lp, _ = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_lp=True)
    l2 = np.full(k, 1e-6)
    h, g = ops().more(ctx, packed, chols, xd, ld, lp, ctx.asarray(bg), ctx.asarray(tlp), ctx.asarray(l2), d,
                      self_normalized=snis)
    rh, rg = omore.get_expected_hessian_and_grad(m, l2, x, mapping, bg, tlp, False, snis)
    h, g = h.numpy(), g.numpy()
    assert np.all(np.isfinite(h)) and np.all(np.isfinite(g))
    scale_h = np.abs(rh).max(axis=(1, 2), keepdims=True)
    scale_g = np.abs(rg).max(axis=1, keepdims=True)
    # the fp32 inputs of the regression (z, weights, rewards) carry ~1e-6 relative error that the ridge solution of a larger
    # system amplifies more than at D = 20: 1e-2 of the per-component magnitude
    assert np.all(np.abs(h - rh) <= 1e-2 * scale_h + 1e-5), np.abs(h - rh).max() / scale_h.max()
    assert np.all(np.abs(g - rg) <= 1e-2 * scale_g + 1e-5), np.abs(g - rg).max() / scale_g.max()


@pytest.mark.parametrize("snis", [True, False])
def test_more_on_a_blocked_path_dimension(ctx, rng, snis):
    """50 < D <= 63 under the default GMMVI_BLOCKED_ABOVE = 50: the model's component blocks come from the blocked pack
    ([mu | log-normaliser | dense L^-1], another stride) which the MORE kernels cannot read -- gmmvi_more re-packs the
    components in the register-path layout for its call (D = 56: F + 1 = 1 654 features) and matches the oracle as the other
    tiled sizes do; beyond D = 63 it refuses cleanly."""
    from oracle import more as omore
    from gmmvi_amd import _lib
    if _lib.blocked_above() >= 56:
        pytest.skip("GMMVI_BLOCKED_ABOVE moved: D = 56 is a register-path dimension in this process")
    k, d, n = 1, 56, 9000
    m, x, mapping, tlp, tg, bg = _stein_inputs(rng, k, d, n)
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    assert packed.shape[1] != 9344                                       # the blocked block, not Pack<64>
    xd = ctx.asarray(x)
    ld, lp, _ = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_lp=True)
    l2 = np.full(k, 1e-6)
    h, g = ops().more(ctx, packed, chols, xd, ld, lp, ctx.asarray(bg), ctx.asarray(tlp), ctx.asarray(l2), d, self_normalized=snis)
    rh, rg = omore.get_expected_hessian_and_grad(m, l2, x, mapping, bg, tlp, False, snis)
    h, g = h.numpy(), g.numpy()
    assert np.all(np.isfinite(h)) and np.all(np.isfinite(g))
    scale_h = np.abs(rh).max(axis=(1, 2), keepdims=True)
    scale_g = np.abs(rg).max(axis=1, keepdims=True)
    assert np.all(np.abs(h - rh) <= 1e-2 * scale_h + 1e-5), np.abs(h - rh).max() / scale_h.max()
    assert np.all(np.abs(g - rg) <= 1e-2 * scale_g + 1e-5), np.abs(g - rg).max() / scale_g.max()
    if snis:
        # no route beyond D = 63: a clean error from the Python mirror and GMMVI_ERR_ARG through the C ABI
        d2, n2 = 70, 256
        m2, x2, mp2, tlp2, tg2, bg2 = _stein_inputs(rng, 1, d2, n2)
        lw2, mu2, ch2 = upload_model(ctx, m2)
        pk2, _ = ops().pack_components(ctx, mu2, ch2)
        x2d = ctx.asarray(x2)
        ld2, lp2, _ = ops().mixture_eval(ctx, pk2, lw2, x2d, d2, want_ld=True, want_lp=True)
        with pytest.raises(ValueError, match="MORE"):
            ops().more(ctx, pk2, ch2, x2d, ld2, lp2, ctx.asarray(bg2), ctx.asarray(tlp2), ctx.asarray(np.full(1, 1e-6)), d2)
        hh, gg = ctx.empty((1, d2, d2)), ctx.empty((1, d2))
        rc = ctx.lib.gmmvi_more(ctx.handle, 1, d2, pk2.ptr, ch2.ptr, x2d.ptr, n2, ld2.ptr, lp2.ptr, ctx.asarray(bg2).ptr,
                                ctx.asarray(tlp2).ptr, None, 0, 1, ctx.asarray(np.full(1, 1e-6)).ptr, hh.ptr, gg.ptr)
        assert rc == -2                                                  # GMMVI_ERR_ARG (include/gmmvi_hip.h)


@pytest.mark.parametrize("k,d,n", [(2, 10, 40), (3, 20, 150), (4, 20, 240)])
def test_more_fewer_samples_than_features(ctx, rng, k, d, n):
    """Rank-deficient ridge systems (N < F = D(D+1)/2 + D + 1), the state early in a run.  With a ridge that fp64 resolves
    but fp32 does not (1e-6: cond ~ 1e8) the fp64 path must reproduce the oracle; with the reference's 1e-12 the
    regression itself is ill-posed (the fp64 oracle moves by tens of percent under a 1e-6 perturbation of its inputs),
    so only finiteness is asserted there."""
    from oracle import more as omore
    m, x, mapping, tlp, tg, bg = _stein_inputs(rng, k, d, n)
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    xd = ctx.asarray(x)
    ld, lp, _ = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_lp=True)
    mp = mapping + 2
    args = dict(mapping=ctx.asarray(mp, np.int32), map_offset=k - 1 - int(mp.max()), own_samples_only=True)
    for ridge, tol in ((1e-6, 1e-2), (1e-12, None)):
        l2 = np.full(k, ridge)
        h, g = ops().more(ctx, packed, chols, xd, ld, lp, ctx.asarray(bg), ctx.asarray(tlp), ctx.asarray(l2), d, **args)
        h, g = h.numpy(), g.numpy()
        assert np.all(np.isfinite(h)) and np.all(np.isfinite(g))
        if tol is None:
            continue
        rh, rg = omore.get_expected_hessian_and_grad(m, l2, x, mp, bg, tlp, True, True)
        scale_h = np.abs(rh).max(axis=(1, 2), keepdims=True)
        scale_g = np.abs(rg).max(axis=1, keepdims=True)
        assert np.all(np.abs(h - rh) <= tol * scale_h), (np.abs(h - rh) / scale_h).max()
        assert np.all(np.abs(g - rg) <= tol * scale_g), (np.abs(g - rg) / scale_g).max()


def test_more_own_samples(ctx, rng):
    from oracle import more as omore
    k, d, n = 3, 5, 900
    m, x, mapping, tlp, tg, bg = _stein_inputs(rng, k, d, n)
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    xd = ctx.asarray(x)
    ld, lp, _ = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_lp=True)
    mp = mapping + 4
    l2 = np.full(k, 1e-6)
    h, g = ops().more(ctx, packed, chols, xd, ld, lp, ctx.asarray(bg), ctx.asarray(tlp), ctx.asarray(l2), d,
                      mapping=ctx.asarray(mp, np.int32), map_offset=k - 1 - int(mp.max()), own_samples_only=True)
    rh, rg = omore.get_expected_hessian_and_grad(m, l2, x, mp, bg, tlp, True, True)
    np.testing.assert_allclose(h.numpy(), rh, rtol=2e-2, atol=2e-2 * np.abs(rh).max())
    np.testing.assert_allclose(g.numpy(), rg, rtol=2e-2, atol=2e-2 * np.abs(rg).max())


def test_more_quadratic_reward_is_exact(ctx, rng):
    """A reward that IS quadratic is recovered whatever the weights: H = -2A' form, g from the linear term."""
    from oracle import gmm as ogmm2
    k, d, n = 2, 6, 1200
    m, x, mapping, _, _, bg = _stein_inputs(rng, k, d, n)
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    xd = ctx.asarray(x)
    ld, lp, _ = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_lp=True)
    b = rng.normal(size=(d, d)); q = b @ b.T / d + np.eye(d)
    lin = rng.normal(size=d)
    x32 = xd.numpy().astype(np.float64)
    rew = -0.5 * np.einsum("ni,ij,nj->n", x32, q, x32) + x32 @ lin + 0.3
    tlp = rew + lp.numpy().astype(np.float64)              # reward = tlp - logq
    l2 = np.full(k, 1e-10)
    h, g = ops().more(ctx, packed, chols, xd, ld, lp, ctx.asarray(bg), ctx.asarray(tlp), ctx.asarray(l2), d)
    for i in range(k):
        np.testing.assert_allclose(h.numpy()[i], q, rtol=0, atol=3e-2 * np.abs(q).max())
        np.testing.assert_allclose(g.numpy()[i], q @ m.means[i] - lin, rtol=0, atol=3e-2 * np.abs(q @ m.means[i] - lin).max())


def _update_inputs(rng, k, d):
    m = random_gmm(rng, k, d)
    hs = np.stack([(lambda b: b @ b.T / d)(rng.normal(size=(d, d))) for _ in range(k)])
    if k > 2:
        hs[2] = -0.01 * hs[2]
    gs = rng.normal(size=(k, d))
    return m, hs.astype(np.float32).astype(np.float64), gs.astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("k,d", [(3, 4), (8, 20), (5, 10), (1, 2), (4, 50), (2, 64)])
def test_update_components_kl(ctx, rng, k, d):
    m, hs, gs = _update_inputs(rng, k, d)
    m32 = ogmm.FullCovGMM(m.weights, m.means.astype(np.float32), m.covs.astype(np.float32))
    w = ogmm.GmmWrapper(m32, 0.1, 1e-12, 4)
    w.stepsizes = np.linspace(0.05, 0.5, k)
    logw, means, chols = upload_model(ctx, m32)
    last_eta = ctx.asarray(w.last_log_etas); l2 = ctx.asarray(w.l2_regularizers)
    nupd = ctx.asarray(w.num_received_updates); steps = ctx.asarray(w.stepsizes)
    for round_ in range(2):                          # cold bracket, then warm start
        succ, kl, probes = ops().update_components_kl(ctx, means, chols, ctx.asarray(hs), ctx.asarray(gs), steps, 1.0,
                                                      1e-12, last_eta, l2, nupd, want_info=True)
        rs, retas, rkls, rprobes = oupd.apply_ng_update_kl(w, hs, gs, w.stepsizes, 1.0, traces=[])
        np.testing.assert_array_equal(succ.numpy().astype(bool), rs)
        np.testing.assert_array_equal(probes.numpy(), rprobes)          # same bisection path
        np.testing.assert_allclose(last_eta.numpy(), retas, rtol=1e-5)
        np.testing.assert_allclose(kl.numpy(), rkls, rtol=5e-3, atol=1e-5)
        np.testing.assert_allclose(means.numpy(), m32.means, rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(chols.numpy(), m32.chol_cov, rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(l2.numpy(), w.l2_regularizers, rtol=1e-6)
        np.testing.assert_allclose(nupd.numpy(), w.num_received_updates)


def test_update_components_kl_failure(ctx, rng):
    m, hs, gs = _update_inputs(rng, 3, 5)
    hs[0] = np.nan
    hs[1] = -1e6 * np.eye(5)                         # hopelessly indefinite: no eta in the bracket is feasible ... or is
    w = ogmm.GmmWrapper(m, 0.1, 1e-12, 4)
    logw, means, chols = upload_model(ctx, m)
    old_means, old_chols = means.numpy(), chols.numpy()
    last_eta = ctx.asarray(w.last_log_etas); l2 = ctx.asarray(w.l2_regularizers); nupd = ctx.asarray(w.num_received_updates)
    succ, kl, probes = ops().update_components_kl(ctx, means, chols, ctx.asarray(hs), ctx.asarray(gs),
                                                  ctx.asarray(w.stepsizes), 1.0, 1e-12, last_eta, l2, nupd, want_info=True)
    rs, retas, _, _ = oupd.apply_ng_update_kl(w, hs, gs, w.stepsizes, 1.0, traces=[])
    np.testing.assert_array_equal(succ.numpy().astype(bool), rs)
    assert not rs[0]
    np.testing.assert_array_equal(means.numpy()[0], old_means[0])
    np.testing.assert_array_equal(chols.numpy()[0], old_chols[0])
    np.testing.assert_allclose(last_eta.numpy(), retas, rtol=1e-4)
    np.testing.assert_allclose(l2.numpy(), w.l2_regularizers, rtol=1e-6)


@pytest.mark.parametrize("mode", ["direct", "iblr"])
def test_update_components_plain(ctx, rng, mode):
    k, d = 4, 6
    m, hs, gs = _update_inputs(rng, k, d)
    w = ogmm.GmmWrapper(m, 0.1, 1e-12, 4)
    steps = np.full(k, 0.3)
    logw, means, chols = upload_model(ctx, m)
    l2 = ctx.asarray(w.l2_regularizers); nupd = ctx.asarray(w.num_received_updates)
    for round_ in range(2):
        succ = ops().update_components_plain(ctx, mode, means, chols, ctx.asarray(hs), ctx.asarray(gs),
                                             ctx.asarray(steps), 1e-12, l2, nupd)
        rs = (oupd.apply_ng_update_direct if mode == "direct" else oupd.apply_ng_update_iblr)(w, hs, gs, steps)
        np.testing.assert_array_equal(succ.numpy().astype(bool), rs)
        np.testing.assert_allclose(means.numpy(), m.means, rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(chols.numpy(), m.chol_cov, rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(l2.numpy(), w.l2_regularizers, rtol=1e-6)


@pytest.mark.parametrize("k,n", [(3, 64), (8, 512), (100, 3000), (1, 10)])
def test_expected_log_ratios(ctx, rng, k, n):
    ld = rng.normal(size=(k, n)) * 3 - 10
    bg = logsumexp(ld, axis=0) - np.log(k) + rng.normal(size=n) * 0.1
    tlp = rng.normal(size=n) * 5 - 20
    logq = logsumexp(ld - np.log(k), axis=0)
    logw = np.log(rng.dirichlet(np.ones(k)))
    for snis in (True, False):
        e, ess = ops().expected_log_ratios(ctx, ctx.asarray(ld), ctx.asarray(bg), ctx.asarray(tlp), ctx.asarray(logq),
                                           1.0, ctx.asarray(logw), snis, want_ess=True)
        lw = ld - bg[None]
        if snis:
            iw = np.exp(lw - logsumexp(lw, axis=1, keepdims=True))
            ref = iw @ (tlp - logq)
            np.testing.assert_allclose(ess.numpy(), 1 / np.sum(iw ** 2, axis=1), rtol=1e-3)
        else:
            ref = np.exp(lw) @ (tlp - logq) / n
        np.testing.assert_allclose(e.numpy(), ref, rtol=2e-4, atol=1e-4)


@pytest.mark.parametrize("k", [2, 6, 100, 257, 512, 700, 1024, 1300])     # register forms up to 1024, the LDS loop behind
def test_update_weights(ctx, rng, k):
    lw = np.log(rng.dirichlet(np.ones(k)))
    lw = (lw - logsumexp(lw)).astype(np.float32).astype(np.float64)
    elr = (rng.normal(size=k) * 3).astype(np.float32).astype(np.float64)
    for eps in [0.01, 0.3, 1.0, 1e4]:
        logw = ctx.asarray(lw)
        info = ops().update_weights(ctx, "trust-region", logw, ctx.asarray(elr), ctx.asarray([eps]), 1.0, True)
        kl, eta, nl = oweights.weights_bracketing_search(lw, elr, eps, 1.0)
        nl = nl - logsumexp(nl)
        np.testing.assert_allclose(info.numpy()[1], eta, rtol=1e-4)
        np.testing.assert_allclose(logw.numpy(), nl, rtol=1e-4, atol=2e-4)
    logw = ctx.asarray(lw)
    ops().update_weights(ctx, "direct", logw, ctx.asarray(elr), ctx.asarray([0.5]), 1.0)
    u = lw + 0.5 * elr
    nl = np.maximum(u - logsumexp(u), -69.07); nl -= logsumexp(nl)
    np.testing.assert_allclose(logw.numpy(), nl, rtol=1e-4, atol=2e-4)


def test_stepsize_kernels(ctx, rng):
    k = 37
    steps = rng.random(k) * 0.9 + 0.002
    prev, last = rng.normal(size=k), rng.normal(size=k)
    prev[:5] = last[:5] = np.finfo(np.float32).min
    s = ctx.asarray(steps)
    ops().component_stepsize_improvement(ctx, s, ctx.asarray(prev), ctx.asarray(last), 0.001, 1.0, 1.15, 0.85)
    ref = osteps.component_stepsize_improvement(steps.astype(np.float32), np.stack([prev, last], 1).astype(np.float32),
                                                0.001, 1.0, 1.15, 0.85)
    np.testing.assert_allclose(s.numpy(), ref, rtol=1e-6)

    class W:
        pass
    w = W(); w.log_weights = np.log(np.ones(k) / k); w.weights = np.exp(w.log_weights)
    w.reward_history = np.full((k, 2), np.finfo(np.float32).min)
    a = osteps.WeightStepsizeImprovement(1.0, 1e-4, 1.0, 1.15, 0.85)
    state = ctx.asarray([1.0, np.finfo(np.float32).min])
    for rewards in [w.reward_history[:, -1], rng.normal(size=k) - 5, rng.normal(size=k) - 50, rng.normal(size=k)]:
        w.reward_history = np.stack([rewards, rewards], 1)
        ops().weight_stepsize_improvement(ctx, ctx.asarray(w.log_weights), ctx.asarray(rewards), state, 1e-4, 1.0, 1.15,
                                          0.85)
        ref = a.update_stepsize(w)
        np.testing.assert_allclose(state.numpy()[0], ref, rtol=1e-6)


def test_combine_partials(ctx, rng):
    r, n, d = 4, 300, 5
    lp = rng.normal(size=(r, n)) * 5
    g = rng.normal(size=(r, n, d))
    out_lp, out_g = ops().combine_partials(ctx, ctx.asarray(lp), ctx.asarray(g), d)
    ref_lp = logsumexp(lp, axis=0)
    np.testing.assert_allclose(out_lp.numpy(), ref_lp, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out_g.numpy(), np.einsum('rn,rnd->nd', np.exp(lp - ref_lp[None]), g), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("k,d", [(6, 4), (16, 20), (5, 33), (3, 64), (7, 1), (4, 2)])
def test_update_kl_fast_kernel_matches_reference_formulation(ctx, rng, k, d):
    """Production kernel (whitened + tridiagonal + speculative bisection) vs the kernel that follows the reference's
    own arithmetic, over cold and warm starts and a non-symmetric reward Hessian (plain-IW Stein branch)."""
    m, hs, gs = _update_inputs(rng, k, d)
    if k > 1:
        hs[1] = hs[1] + 0.05 * rng.normal(size=(d, d))                 # not symmetric: only the lower part counts
    state = {}
    for name in ("fast", "ref"):
        logw, means, chols = upload_model(ctx, m)
        state[name] = dict(means=means, chols=chols, last_eta=ctx.full((k,), -1.0), l2=ctx.full((k,), 1e-12),
                           nupd=ctx.zeros((k,)))
    steps = ctx.asarray(np.linspace(0.02, 0.6, k))
    for round_ in range(3):
        out = {}
        for name in ("fast", "ref"):
            st = state[name]
            out[name] = ops().update_components_kl(ctx, st["means"], st["chols"], ctx.asarray(hs), ctx.asarray(gs), steps,
                                                   1.0, 1e-12, st["last_eta"], st["l2"], st["nupd"], want_info=True,
                                                   reference=(name == "ref"))
        np.testing.assert_array_equal(out["fast"][0].numpy(), out["ref"][0].numpy())
        np.testing.assert_array_equal(out["fast"][2].numpy(), out["ref"][2].numpy())           # identical probe counts
        np.testing.assert_allclose(out["fast"][1].numpy(), out["ref"][1].numpy(), rtol=5e-3, atol=1e-5)
        np.testing.assert_allclose(state["fast"]["last_eta"].numpy(), state["ref"]["last_eta"].numpy(), rtol=1e-5)
        np.testing.assert_allclose(state["fast"]["means"].numpy(), state["ref"]["means"].numpy(), rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(state["fast"]["chols"].numpy(), state["ref"]["chols"].numpy(), rtol=2e-3, atol=2e-4)


def test_update_kl_emits_packed_blocks(ctx, rng):
    """The parameter blocks written by the update kernel equal gmmvi_pack_components on the updated model."""
    for k, d in [(5, 4), (6, 20), (3, 33)]:
        m, hs, gs = _update_inputs(rng, k, d)
        hs[0] = np.nan                                  # one rejected component keeps (and packs) its old parameters
        logw, means, chols = upload_model(ctx, m)
        succ, _, _, packed = ops().update_components_kl(
            ctx, means, chols, ctx.asarray(hs), ctx.asarray(gs), ctx.full((k,), 0.1), 1.0, 1e-12, ctx.full((k,), -1.0),
            ctx.full((k,), 1e-12), ctx.zeros((k,)), want_packed=True)
        assert succ.numpy()[0] == 0 and succ.numpy()[1:].all()
        ref, _ = ops().pack_components(ctx, means, chols)
        np.testing.assert_allclose(packed.numpy(), ref.numpy(), rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("k,d,n", [(3, 4, 64), (8, 20, 512), (40, 20, 1000), (5, 50, 130)])
def test_mixture_eval_dual(ctx, rng, k, d, n):
    """Fused background + model sweep equals the two separate sweeps."""
    m = random_gmm(rng, k, d)
    x = m.means[rng.integers(0, k, n)] + rng.normal(size=(n, d)) * 1.5
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    counts = rng.integers(1, 50, k).astype(np.float64)
    logc = ctx.asarray(np.log(counts / counts.sum()))
    xd = ctx.asarray(x)
    ld, lp, grad, bg = ops().mixture_eval_dual(ctx, packed, logw, logc, xd, d)
    ld1, lp1, grad1 = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_grad=True)
    _, bg1, _ = ops().mixture_eval(ctx, packed, logc, xd, d)
    np.testing.assert_allclose(ld.numpy(), ld1.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(lp.numpy(), lp1.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(grad.numpy(), grad1.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(bg.numpy(), bg1.numpy(), rtol=1e-6, atol=1e-6)
    ref = logsumexp(m.component_log_densities(x.astype(np.float32).astype(np.float64)) + np.log(counts / counts.sum())[:, None], axis=0)
    np.testing.assert_allclose(bg.numpy(), ref, rtol=1e-4, atol=2e-4)


def test_small_uploads_through_the_staging_ring_keep_their_contents(ctx, rng):
    """Host -> device copies of up to 64 KB go through a ring of 64 pinned staging slots without waiting for the stream
    (csrc/api.hip: gmmvi_upload): several laps around the ring, sizes up to the slot size and just above it (the synchronous
    route), interleaved with kernels on the same stream -- every array must arrive intact."""
    host, dev = [], []
    sizes = [1, 7, 300, 4096, 16384, 16385, 20000, 3]            # floats: 16384 = one full slot, 16385 takes the other route
    for i in range(300):
        a = rng.normal(size=sizes[i % len(sizes)]).astype(np.float32)
        host.append(a)
        dev.append(ctx.asarray(a))
        if i % 5 == 0:                                           # keep the stream busy between the copies
            ops().exp_into(ctx, ctx.empty((1 << 16,)), ctx.zeros((1 << 16,)))
    for a, d in zip(host, dev):
        np.testing.assert_array_equal(d.numpy(), a)


def test_database_buffers_grow_in_place(ctx, rng, monkeypatch):
    """optimization/sample_db.py _Growable: doubling while small, then one move into a reserved address range that grows by
    mapping chunks (gmmvi_vmm_*): appends across the move and across a chunk boundary keep every row; assign() re-fills it."""
    from gmmvi_amd.optimization.sample_db import _Growable
    monkeypatch.setattr(_Growable, "MAPPED_FROM", 8 << 20)
    monkeypatch.setattr(_Growable, "FIRST_APPENDS", 2)
    width = 1 << 16                                            # 256 KiB per row
    g = _Growable(ctx, (width,))
    rows = []
    for i in range(9):                                          # 4 rows per append: 1 MiB; the buffer moves into a mapped range on the way
        blk = rng.normal(size=(4, 8)).astype(np.float32)
        rows.append(blk)
        full = np.zeros((4, width), np.float32); full[:, :8] = blk; full[:, -1] = i
        g.append(ctx.asarray(full))
    assert g._range is not None and g.n == 36
    chunk = g._range.chunk
    big = ctx.zeros((chunk // (width * 4) + 8, width))          # one append that crosses the first chunk boundary
    g.append(big)
    assert g._range.mapped >= 2 * chunk and g.n == 36 + big.shape[0]
    head = g.view(0, 36).numpy()
    np.testing.assert_array_equal(head[:, :8], np.concatenate(rows))
    np.testing.assert_array_equal(head[:, -1], np.repeat(np.arange(9, dtype=np.float32), 4))
    tail = ctx.asarray(np.full((3, width), 7.0, np.float32))
    g.append(tail)
    np.testing.assert_array_equal(g.view(g.n - 3).numpy()[:, ::4096], 7.0)
    g.assign(ctx.asarray(np.full((5, width), 2.0, np.float32)))
    assert g.n == 5 and g._range is not None
    np.testing.assert_array_equal(g.view().numpy()[:, ::4096], 2.0)


def test_database_buffers_keep_doubling_without_virtual_memory_management(ctx, monkeypatch):
    """A stack that cannot reserve / map address ranges: the buffer warns once and goes on doubling by copy."""
    from gmmvi_amd.optimization import sample_db
    monkeypatch.setattr(sample_db._Growable, "MAPPED_FROM", 8 << 20)
    monkeypatch.setattr(sample_db._Growable, "FIRST_APPENDS", 2)

    def refuse(self, ctx, reserve_bytes):
        raise sample_db.hip_ops._lib.GmmviError("virtual memory management is not supported")
    monkeypatch.setattr(sample_db._MappedRange, "__init__", refuse)
    g = sample_db._Growable(ctx, (1 << 16,))
    with pytest.warns(UserWarning, match="grow-in-place"):
        for i in range(12):
            g.append(ctx.asarray(np.full((4, 1 << 16), float(i), np.float32)))
    assert g._range is None and g.n == 48
    np.testing.assert_array_equal(g.view().numpy()[:, 0], np.repeat(np.arange(12, dtype=np.float32), 4))
    assert sample_db._Growable.MAPPED_FROM == float("inf")
    monkeypatch.setattr(sample_db._Growable, "MAPPED_FROM", 1 << 30)       # (the class attribute the fallback set: restored)
