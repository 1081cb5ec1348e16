"""GPU parity of the blocked path (64 < D <= 512: config C5, D = 300) against the fp64 oracle: same entry points, the
per-pair work on the matrix cores (csrc/blocked.hip).  Tolerances as in test_hip_kernels.py; the quadratic forms sum D
terms in fp32, so absolute errors of log-densities scale with D (stated per assertion)."""
import os

import numpy as np
import pytest
from scipy.special import logsumexp

from oracle import philox, gmm as ogmm, targets as otargets, stein as ostein, updaters as oupd
from test_hip_kernels import random_gmm, upload_model, _stein_inputs, _update_inputs, ops
from test_hip_train_iter import run_pair
from helpers import samtron_config

pytestmark = pytest.mark.gpu

# (3, 201, 150) and (2, 161, 600): wide enough for the split-operand route of the contractions (>= 160 columns) with rows that are
# NOT multiples of 16 bytes -- its element-wise staging route; (4, 300, 200) / (2, 512, 140): its aligned route; the others run
# the f32 matrix-core route for the triangular launches (csrc/blocked.hip: bgemm_use_split)
# (2, 324, 200) / (2, 332, 130): a second column tile whose k range is SHORTER than one 16-wide staging step (324 - 320 = 4):
# pieces of the fast staging route that lie behind the end of the range must read inside the operand
DIMS = [(3, 72, 300), (2, 130, 257), (4, 300, 200), (1, 65, 10), (2, 512, 140), (3, 201, 150), (2, 161, 600), (2, 324, 200),
        (2, 332, 130)]


@pytest.fixture(scope="module")
def ctx():
    from gmmvi_amd.device import get_context
    return get_context()


@pytest.mark.parametrize("k,d,n", DIMS)
def test_blocked_pack_cholesky(ctx, rng, k, d, n):
    m = random_gmm(rng, k, d)
    _, means, chols = upload_model(ctx, m)
    packed, inv = ops().pack_components(ctx, means, chols, want_inverse=True)
    ref = np.linalg.inv(m.chol_cov)
    np.testing.assert_allclose(inv.numpy(), ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max())
    p = packed.numpy()
    np.testing.assert_allclose(p[:, :d], m.means, rtol=1e-6)
    const = -np.log(np.diagonal(m.chol_cov, axis1=1, axis2=2)).sum(axis=1) - 0.5 * d * np.log(2 * np.pi)
    np.testing.assert_allclose(p[:, d], const, rtol=1e-5)
    ch, ok = ops().cholesky(ctx, ctx.asarray(m.covs))
    assert ok.numpy().all()
    np.testing.assert_allclose(ch.numpy(), m.chol_cov, rtol=2e-4, atol=2e-5)
    bad = m.covs.copy(); bad[0] = -np.eye(d)
    ch, ok = ops().cholesky(ctx, ctx.asarray(bad))
    assert ok.numpy()[0] == 0 and np.isnan(ch.numpy()[0]).all() and ok.numpy()[1:].all()


@pytest.mark.parametrize("k,d,n", DIMS)
def test_blocked_mixture_eval(ctx, rng, k, d, n):
    m = random_gmm(rng, k, d)
    x = m.means[rng.integers(0, k, n)] + rng.normal(size=(n, d)) * 1.5
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    xd = ctx.asarray(x)
    ld, lp, grad = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_lp=True, want_grad=True)
    lq, g, cld = m.log_density_and_grad(x.astype(np.float32).astype(np.float64))
    # |z|^2 is O(D..100 D) here and carries fp32 relative error ~1e-6 per term: atol scales with the magnitude
    atol = 2e-6 * np.abs(cld).max() + 2e-4
    np.testing.assert_allclose(ld.numpy(), cld, rtol=1e-4, atol=atol)
    np.testing.assert_allclose(lp.numpy(), lq, rtol=1e-4, atol=atol)
    np.testing.assert_allclose(grad.numpy(), g, rtol=1e-3, atol=1e-3 * np.abs(g).max())
    _, lp2, _ = ops().mixture_eval(ctx, packed, logw, xd, d)
    np.testing.assert_allclose(lp2.numpy(), lp.numpy(), rtol=1e-6, atol=1e-6)
    # dual sweep
    counts = rng.integers(1, 50, k).astype(np.float64)
    logc = ctx.asarray(np.log(counts / counts.sum()))
    ld3, lp3, grad3, bg = ops().mixture_eval_dual(ctx, packed, logw, logc, xd, d)
    np.testing.assert_allclose(ld3.numpy(), ld.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(grad3.numpy(), grad.numpy(), rtol=1e-5, atol=1e-5)
    ref = logsumexp(cld + np.log(counts / counts.sum())[:, None], axis=0)
    np.testing.assert_allclose(bg.numpy(), ref, rtol=1e-4, atol=atol)


def test_blocked_mixture_eval_component_chunks(ctx, rng):
    """The component-chunked route (Z of all components does not fit the scratch budget) gives the same results."""
    k, d, n = 5, 96, 150
    m = random_gmm(rng, k, d)
    x = m.means[rng.integers(0, k, n)] + rng.normal(size=(n, d)) * 1.5
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    xd = ctx.asarray(x)
    ld, lp, grad = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_lp=True, want_grad=True)
    os.environ["GMMVI_BLOCKED_ZBYTES"] = str(2 * n * d * 4)          # two components per chunk
    try:
        ld2, lp2, grad2 = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_lp=True, want_grad=True)
        _, lp3, grad3 = ops().mixture_eval(ctx, packed, logw, xd, d, want_grad=True)
    finally:
        del os.environ["GMMVI_BLOCKED_ZBYTES"]
    np.testing.assert_array_equal(ld2.numpy(), ld.numpy())
    np.testing.assert_array_equal(lp2.numpy(), lp.numpy())
    np.testing.assert_allclose(grad2.numpy(), grad.numpy(), rtol=1e-5, atol=1e-5 * np.abs(grad.numpy()).max())
    np.testing.assert_array_equal(lp3.numpy(), lp.numpy())
    np.testing.assert_allclose(grad3.numpy(), grad.numpy(), rtol=1e-5, atol=1e-5 * np.abs(grad.numpy()).max())


def test_blocked_student_t_target(ctx, rng):
    d, n = 80, 300
    t = otargets.make_stm_target(d, rng)
    x = t.means[rng.integers(0, t.means.shape[0], n)] + rng.normal(size=(n, d)) * 2
    from gmmvi_amd import _lib
    packed, _ = ops().pack_components(ctx, ctx.asarray(t.means), ctx.asarray(t.chols), family=_lib.STUDENT_T, nu=2.0)
    _, lp, grad = ops().mixture_eval(ctx, packed, ctx.asarray(t.log_weights), ctx.asarray(x), d, family=_lib.STUDENT_T,
                                     nu=2.0, want_grad=True)
    rlp, rg = t.log_density_and_grad(x.astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(lp.numpy(), rlp, rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(grad.numpy(), rg, rtol=2e-3, atol=2e-3 * np.abs(rg).max())


@pytest.mark.parametrize("k,d,n", DIMS[:4] + DIMS[5:6])
def test_blocked_sample_components(ctx, rng, k, d, n):
    m = random_gmm(rng, k, d)
    n_k = rng.multinomial(n, np.ones(k) / k)
    offs = np.concatenate([[0], np.cumsum(n_k)]).astype(np.int32)
    _, means, chols = upload_model(ctx, m)
    eps = philox.normals(3, 100, n, d)
    x, mp = ops().sample_components(ctx, means, chols, ctx.asarray(offs, np.int32), n, eps=ctx.asarray(eps))
    rx, rmp = m.sample_from_components_no_shuffle(n_k, eps.astype(np.float32).astype(np.float64))
    np.testing.assert_array_equal(mp.numpy(), rmp)
    np.testing.assert_allclose(x.numpy(), rx, rtol=1e-5, atol=2e-5)
    x2, _ = ops().sample_components(ctx, means, chols, ctx.asarray(offs, np.int32), n, seed=3, first_index=100)
    np.testing.assert_allclose(x2.numpy(), rx, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k,d,n", [(3, 72, 600), (2, 300, 900), (4, 130, 257), (2, 201, 700)])
@pytest.mark.parametrize("snis", [True, False])
def test_blocked_stein(ctx, rng, k, d, n, snis):
    m, x, mapping, tlp, tg, bg = _stein_inputs(rng, k, d, n)
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    xd = ctx.asarray(x)
    ld, lp, qg = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_grad=True)
    h, g = ops().stein(ctx, packed, xd, ld, qg, ctx.asarray(bg), ctx.asarray(tg), d, self_normalized=snis)
    rh, rg = ostein.get_expected_hessian_and_grad(m, x, mapping, bg, tlp, tg, False, snis)
    scale_h = np.abs(rh).max(axis=(1, 2), keepdims=True)
    scale_g = np.abs(rg).max(axis=1, keepdims=True)
    # the weights exp(ld - bg) amplify the fp32 error of ld, which grows with D (see test_blocked_mixture_eval)
    assert np.all(np.abs(h.numpy() - rh) <= 1e-2 * scale_h + 1e-6)
    assert np.all(np.abs(g.numpy() - rg) <= 1e-2 * scale_g + 1e-6)


@pytest.mark.parametrize("snis", [True, False])
def test_blocked_stein_own_samples(ctx, rng, snis):
    k, d, n = 3, 70, 500
    m, x, mapping, tlp, tg, bg = _stein_inputs(rng, k, d, n)
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    xd = ctx.asarray(x)
    ld, lp, qg = ops().mixture_eval(ctx, packed, logw, xd, d, want_ld=True, want_grad=True)
    mp = mapping + 7
    h, g = ops().stein(ctx, packed, xd, ld, qg, ctx.asarray(bg), ctx.asarray(tg), d,
                       mapping=ctx.asarray(mp, np.int32), map_offset=k - 1 - int(mp.max()), own_samples_only=True,
                       self_normalized=snis)
    rh, rg = ostein.get_expected_hessian_and_grad(m, x, mp, bg, tlp, tg, True, snis)
    np.testing.assert_allclose(h.numpy(), rh, rtol=2e-3, atol=2e-3 * np.abs(rh).max())
    np.testing.assert_allclose(g.numpy(), rg, rtol=2e-3, atol=2e-3 * np.abs(rg).max())


def test_blocked_stein_after_workspace_grow(rng):
    """A fresh context (empty workspace): at D = 300, N = 600 the Stein scratch is about twice the density sweep's, beyond
    the 1.5x slack of gmmvi_ws_reserve, so the workspace is re-allocated between the sweep that left Z behind and the
    Stein call.  hipMalloc may return the old address: the hand-over of Z must be dropped with the old block
    (api.hip gmmvi_ws_reserve), otherwise the contraction would read uninitialised memory."""
    from gmmvi_amd.device import Context
    fresh = Context()
    k, d, n = 2, 300, 600
    m, x, mapping, tlp, tg, bg = _stein_inputs(rng, k, d, n)
    logw, means, chols = upload_model(fresh, m)
    packed, _ = ops().pack_components(fresh, means, chols)
    xd = fresh.asarray(x)
    for _ in range(2):          # second round: the workspace is large enough, the hand-over is taken
        ld, lp, qg = ops().mixture_eval(fresh, packed, logw, xd, d, want_ld=True, want_grad=True)
        h, g = ops().stein(fresh, packed, xd, ld, qg, fresh.asarray(bg), fresh.asarray(tg), d)
        rh, rg = ostein.get_expected_hessian_and_grad(m, x, mapping, bg, tlp, tg, False, True)
        scale_h = np.abs(rh).max(axis=(1, 2), keepdims=True)
        scale_g = np.abs(rg).max(axis=1, keepdims=True)
        assert np.all(np.abs(h.numpy() - rh) <= 1e-2 * scale_h + 1e-6)
        assert np.all(np.abs(g.numpy() - rg) <= 1e-2 * scale_g + 1e-6)


@pytest.mark.parametrize("k,d", [(3, 72), (2, 300), (3, 129), (2, 201)])
def test_blocked_update_components_kl(ctx, rng, k, d):
    m, hs, gs = _update_inputs(rng, k, d)
    m32 = ogmm.FullCovGMM(m.weights, m.means.astype(np.float32), m.covs.astype(np.float32))
    w = ogmm.GmmWrapper(m32, 0.1, 1e-12, 4)
    w.stepsizes = np.linspace(0.05, 0.5, k)
    logw, means, chols = upload_model(ctx, m32)
    last_eta = ctx.asarray(w.last_log_etas); l2 = ctx.asarray(w.l2_regularizers)
    nupd = ctx.asarray(w.num_received_updates); steps = ctx.asarray(w.stepsizes)
    for round_ in range(2):
        succ, kl, probes, packed = ops().update_components_kl(ctx, means, chols, ctx.asarray(hs), ctx.asarray(gs), steps,
                                                              1.0, 1e-12, last_eta, l2, nupd, want_info=True,
                                                              want_packed=True)
        rs, retas, rkls, rprobes = oupd.apply_ng_update_kl(w, hs, gs, w.stepsizes, 1.0, traces=[])
        np.testing.assert_array_equal(succ.numpy().astype(bool), rs)
        np.testing.assert_array_equal(probes.numpy(), rprobes)
        np.testing.assert_allclose(last_eta.numpy(), retas, rtol=1e-5)
        np.testing.assert_allclose(kl.numpy(), rkls, rtol=1e-2, atol=1e-5)
        np.testing.assert_allclose(means.numpy(), m32.means, rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(chols.numpy(), m32.chol_cov, rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(l2.numpy(), w.l2_regularizers, rtol=1e-6)
        np.testing.assert_allclose(nupd.numpy(), w.num_received_updates)
        ref, _ = ops().pack_components(ctx, means, chols)
        np.testing.assert_allclose(packed.numpy(), ref.numpy(), rtol=2e-6, atol=1e-6)


def test_blocked_update_components_kl_failure(ctx, rng):
    k, d = 3, 80
    m, hs, gs = _update_inputs(rng, k, d)
    hs[0] = np.nan
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


@pytest.mark.parametrize("kind,d,k,s,iters", [("gmm", 64, 3, 60, 8), ("gmm", 72, 3, 60, 10), ("gauss", 300, 2, 64, 8)])
def test_blocked_trajectory_matches_oracle(kind, d, k, s, iters):
    """GMMVI.train_iter() at D >= 64 (modular plug-in path over the blocked kernels) against the oracle on the same draws:
    parameters, accept / reject decisions, multipliers per iteration, and the matched ELBO at the end."""
    cfg = samtron_config(s)
    o, g, worst = run_pair(kind, d, k, s, seed=11, iters=iters, cfg=cfg, tol_scale=2.0)
    elbo_o = o.elbo(4000, seed=5)[0]
    o.model.model.means = g.model.means.numpy().astype(np.float64)
    o.model.model.chol_cov = g.model.chol_cov.numpy().astype(np.float64)
    o.model.model.log_weights = g.model.log_weights.numpy().astype(np.float64)
    elbo_g = o.elbo(4000, seed=5)[0]
    assert abs(elbo_g - elbo_o) < 1e-2 + 1e-3 * abs(elbo_o), (elbo_g, elbo_o, worst)


def test_c5_shard_shape_matches_oracle():
    """The composition bench.py --workload c5 times (BASELINE configs[4] per GPU: single-Gaussian target D = 300 with the
    gmm.py:148-162 law, stm300.yml initial mixture, 312 samples per component, single-call path not eligible -> modular
    path on the blocked kernels), cut from 64 to 8 components so that the fp64 oracle finishes in seconds: two iterations,
    state compared after each (parameters 1e-3 of the parameter scale, identical accept / reject decisions)."""
    import importlib, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    w = bench.build("c5_k8", 1, 0)
    assert (w["d"], w["k_total"], w["s"]) == (300, 8, 312)
    o = bench.make_oracle(w)
    g = bench.make_gmmvi(w, 1, 0)
    g.ng_based_updater.want_info = True
    for it in range(2):
        info = o.train_iter()
        g.train_iter()
        om, gm = o.model, g.model
        for key, got, ref in (("means", gm.means.numpy(), om.means), ("chols", gm.chol_cov.numpy(), om.chol_cov)):
            dev = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
            assert dev <= 1e-3, f"iteration {it}: {key} deviate by {dev:.2e}"
        np.testing.assert_allclose(np.exp(gm.log_weights.numpy()), om.weights, atol=1e-3)
        np.testing.assert_array_equal(g.ng_based_updater.last_success.numpy().astype(bool), info["success"])
        # the bisection returns the first probe whose KL is within 10 % of the bound (ng_based_component_updater.py:240-243): where
        # a probe sits at that edge, f32 rounding decides between it and its neighbour -- all but at most one component agree to
        # 1 %, that one to the width of the band
        etas, ref_etas = gm.last_log_etas.numpy(), om.last_log_etas
        rel = np.abs(etas - ref_etas) / np.maximum(np.abs(ref_etas), 1e-6)
        assert (rel > 1e-2).sum() <= 1 and rel.max() <= 0.15, rel
        np.testing.assert_allclose(gm.stepsizes.numpy(), om.stepsizes, rtol=1e-6)


def test_blocked_runner_flow_stm300():
    """The shipped 300-dimensional Student-t experiment (stm300.yml) through GmmviRunner with the SAMTRON defaults:
    sample reuse, adaptive number of components, Student-t target -- every module on the blocked kernels."""
    from gmmvi.gmmvi_runner import GmmviRunner
    from gmmvi.configs import update_config, get_default_experiment_config, get_default_algorithm_config
    config = update_config(update_config(get_default_experiment_config("stm300"), {"start_seed": 0}),
                           update_config(get_default_algorithm_config("SAMTRON"),
                                         {"gmmvi_runner_config": {"log_metrics_interval": 5}}))
    runner = GmmviRunner.build_from_config(config=config)
    elbos = []
    for n in range(16):
        m = runner.iterate_and_log(n)
        if "-elbo" in m:
            elbos.append(-m["-elbo"])
    assert len(elbos) >= 3 and all(np.isfinite(elbos)) and elbos[-1] > elbos[0]
    assert runner.gmmvi.model.num_dimensions == 300


def test_blocked_stein_reuses_whitened_samples_only_when_inputs_match(ctx, rng):
    """The Stein estimate reuses the whitened samples left in the scratch by the density sweep that produced ld / qgrad,
    guarded by device-side content hashes: stale scratch (same pointers, different contents) is recomputed."""
    k, d, n = 3, 72, 400
    m, x2, mapping, tlp, tg, bg = _stein_inputs(rng, k, d, n)
    x1 = x2 + rng.normal(size=x2.shape)
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    bgd, tgd = ctx.asarray(bg), ctx.asarray(tg)
    x2d = ctx.asarray(x2)
    ld2, _, qg2 = ops().mixture_eval(ctx, packed, logw, x2d, d, want_ld=True, want_grad=True)
    h_ref, g_ref = ops().stein(ctx, packed, x2d, ld2, qg2, bgd, tgd, d)          # reuse: same contents
    # defeat the hand-over (another scratch user in between): same result bit for bit
    ld2b, _, qg2b = ops().mixture_eval(ctx, packed, logw, x2d, d, want_ld=True, want_grad=True)
    ops().mixture_eval(ctx, packed, logw, ctx.asarray(x1), d)
    h_b, g_b = ops().stein(ctx, packed, x2d, ld2b, qg2b, bgd, tgd, d)
    np.testing.assert_array_equal(h_b.numpy(), h_ref.numpy())
    np.testing.assert_array_equal(g_b.numpy(), g_ref.numpy())
    # stale scratch: the sweep ran on x1, then the SAME buffer is overwritten with x2
    xbuf = ctx.asarray(x1)
    ops().mixture_eval(ctx, packed, logw, xbuf, d, want_ld=True, want_grad=True)
    xbuf.copy_from(x2d)
    h_c, g_c = ops().stein(ctx, packed, xbuf, ld2, qg2, bgd, tgd, d)
    np.testing.assert_array_equal(h_c.numpy(), h_ref.numpy())
    np.testing.assert_array_equal(g_c.numpy(), g_ref.numpy())
    rh, rg = ostein.get_expected_hessian_and_grad(m, x2, mapping, bg, tlp, tg, False, True)
    assert np.all(np.abs(h_ref.numpy() - rh) <= 1e-2 * np.abs(rh).max(axis=(1, 2), keepdims=True) + 1e-6)


@pytest.mark.parametrize("mode", ["direct", "iblr"])
@pytest.mark.parametrize("k,d", [(3, 72), (2, 130)])
def test_blocked_update_components_plain(ctx, rng, mode, k, d):
    m, hs, gs = _update_inputs(rng, k, d)
    hs[0] = -50.0 * np.eye(d)     # direct: new precision not positive definite -> rejected, parameters kept (iBLR stays PD)
    w = ogmm.GmmWrapper(m, 0.1, 1e-12, 4)
    steps = np.full(k, 0.3)
    logw, means, chols = upload_model(ctx, m)
    l2 = ctx.asarray(w.l2_regularizers); nupd = ctx.asarray(w.num_received_updates)
    for round_ in range(2):
        succ = ops().update_components_plain(ctx, mode, means, chols, ctx.asarray(hs), ctx.asarray(gs),
                                             ctx.asarray(steps), 1e-12, l2, nupd)
        rs = (oupd.apply_ng_update_direct if mode == "direct" else oupd.apply_ng_update_iblr)(w, hs, gs, steps)
        np.testing.assert_array_equal(succ.numpy().astype(bool), rs)
        assert rs[1:].all() and (mode == "iblr" or not rs[0])
        np.testing.assert_allclose(means.numpy(), m.means, rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(chols.numpy(), m.chol_cov, rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(l2.numpy(), w.l2_regularizers, rtol=1e-6)
        np.testing.assert_allclose(nupd.numpy(), w.num_received_updates)


_ROUTE_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from gmmvi_amd.device import get_context
from gmmvi_amd import hip_ops
d = np.load(sys.argv[2])
ctx = get_context()
packed, _ = hip_ops.pack_components(ctx, ctx.asarray(d["means"]), ctx.asarray(d["chols"]))
ld, lp, grad = hip_ops.mixture_eval(ctx, packed, ctx.asarray(d["logw"]), ctx.asarray(d["x"]), int(d["x"].shape[1]),
                                    want_ld=True, want_lp=True, want_grad=True)
np.savez(sys.argv[3], ld=ld.numpy(), lp=lp.numpy(), grad=grad.numpy())
"""


def test_split_operand_route_is_as_accurate_as_the_f32_route(ctx, rng, tmp_path):
    """The default route of the blocked contractions (three bf16 planes per f32 operand, six partial products on the bf16 matrix
    cores, f32 accumulation: csrc/blocked.hip) against the f32 matrix-core route (GMMVI_BLOCKED_F32=1, run in a child process:
    the switch is read once per process), both measured against the fp64 oracle on the same inputs: the split route's errors
    must be of the size of the f32 route's (it drops terms below 2^-25 of a product, less than the rounding of one f32
    multiply-add).  D = 300, means far from the origin (|mu| ~ 30 sigma): x - mu is formed in f32 BEFORE the split."""
    import subprocess, sys
    k, d, n = 4, 300, 1024
    m = random_gmm(rng, k, d)
    m.means[:] = m.means + 30.0
    x = m.means[rng.integers(0, k, n)] + rng.normal(size=(n, d)) * 1.5
    x = x.astype(np.float32).astype(np.float64)
    inp = tmp_path / "in.npz"
    np.savez(inp, means=m.means.astype(np.float32), chols=m.chol_cov.astype(np.float32), logw=m.log_weights.astype(np.float32),
             x=x.astype(np.float32))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, flag in (("split", "0"), ("f32", "1")):
        out = tmp_path / f"{name}.npz"
        env = dict(os.environ, GMMVI_BLOCKED_F32=flag)
        subprocess.run([sys.executable, "-c", _ROUTE_SCRIPT, root, str(inp), str(out)], check=True, env=env, timeout=300)
        outs[name] = np.load(out)
    # fp64 reference on the f32-rounded model (what both routes were given)
    m.means = m.means.astype(np.float32).astype(np.float64)
    m.chol_cov = m.chol_cov.astype(np.float32).astype(np.float64)
    m.log_weights = m.log_weights.astype(np.float32).astype(np.float64)
    lq, g, cld = m.log_density_and_grad(x)
    err = {name: (np.abs(o["ld"] - cld).max(), np.abs(o["grad"] - g).max() / np.abs(g).max()) for name, o in outs.items()}
    # same tolerance as test_blocked_mixture_eval for both, and the split route within 2x of the f32 route (+ a floor)
    atol = 2e-6 * np.abs(cld).max() + 2e-4
    for name in err:
        assert err[name][0] <= atol, (name, err)
        assert err[name][1] <= 1e-3, (name, err)
    assert err["split"][0] <= 2.0 * err["f32"][0] + 1e-4, err
    assert err["split"][1] <= 2.0 * err["f32"][1] + 1e-6, err
    print("max |ld - fp64| / relative gradient error:", err)


def test_split_operand_route_keeps_non_finite_samples_to_themselves(ctx, rng):
    """A NaN / an infinity in one sample must surface in that sample's results only (the split of x - mu into three bf16 planes
    turns an infinity into a NaN -- inf - bf16(inf) -- which is what the f32 route's 0 x inf products give as well)."""
    k, d, n = 2, 300, 256
    m = random_gmm(rng, k, d)
    x = m.means[rng.integers(0, k, n)] + rng.normal(size=(n, d))
    logw, means, chols = upload_model(ctx, m)
    packed, _ = ops().pack_components(ctx, means, chols)
    _, lp0, g0 = ops().mixture_eval(ctx, packed, logw, ctx.asarray(x), d, want_lp=True, want_grad=True)
    xb = x.copy()
    xb[5, 17] = np.nan
    xb[130, 299] = np.inf
    _, lp1, g1 = ops().mixture_eval(ctx, packed, logw, ctx.asarray(xb), d, want_lp=True, want_grad=True)
    lp0, lp1, g0, g1 = lp0.numpy(), lp1.numpy(), g0.numpy(), g1.numpy()
    bad = np.zeros(n, bool); bad[[5, 130]] = True
    assert not np.isfinite(lp1[bad]).any()
    np.testing.assert_array_equal(lp1[~bad], lp0[~bad])
    np.testing.assert_array_equal(g1[~bad], g0[~bad])
