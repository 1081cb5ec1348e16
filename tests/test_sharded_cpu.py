"""World-size-2 gloo test of the component-sharded train_iter orchestration (gmmvi_amd/sharded.py) on CPU.

The exchange logic (which rows go to which rank, how partial log-sum-exps and gradients are recombined, how the
replicated weight / reward state stays in sync) is back-end agnostic; here it runs on an oracle-backed ``ops`` object
(NumPy fp64, test infrastructure) with a torch.distributed gloo exchange, and must reproduce the unsharded oracle
trajectory.  On the GPU the same class runs with HipOps + RcclExchange (tests/test_hip_sharded.py covers R = 1)."""
import os
import sys
import tempfile

import numpy as np
import pytest
from scipy.special import logsumexp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import philox, gmm as ogmm, stein as ostein, updaters as oupd, weights as oweights, stepsizes as osteps  # noqa: E402
from helpers import samtron_config, make_oracle  # noqa: E402


class OracleOps:
    """The ops interface of ShardedGMMVI on NumPy arrays (fp64), built from oracle/ functions."""

    def __init__(self, target):
        self.target = target

    def asarray(self, x, dtype=np.float32):
        return np.array(x, dtype=np.float64 if np.dtype(dtype) == np.float32 else dtype)

    def full(self, shape, v):
        return np.full(shape, v, np.float64)

    def rows(self, a, lo, hi):
        return a[lo:hi]

    def to_host(self, a):
        return np.asarray(a)

    def copy_into(self, dst, src):
        dst[...] = src

    def sample(self, means, chols, counts, seed, first_index):
        n = int(np.sum(counts))
        eps = philox.normals(seed, first_index, n, means.shape[1])
        mapping = np.repeat(np.arange(len(counts)), counts)
        return means[mapping] + np.einsum('nij,nj->ni', chols[mapping], eps)

    def target_eval(self, x):
        return self.target.log_density_and_grad(x)

    def pack(self, means, chols):
        return means.copy(), chols.copy()

    def _model(self, packed, logw):
        m = object.__new__(ogmm.FullCovGMM)
        m.dtype, m.means, m.chol_cov, m.log_weights = np.float64, packed[0], packed[1], np.asarray(logw)
        m.num_dimensions = packed[0].shape[1]
        return m

    def mixture(self, packed, logw, x, d, want_ld=False, want_grad=False):
        m = self._model(packed, logw)
        if want_grad:
            lq, g, cld = m.log_density_and_grad(x)
            return (cld if want_ld else None), lq, g
        lq, cld = m.log_densities_also_individual(x)
        return (cld if want_ld else None), lq, None

    def mixture_dual(self, packed, logw, logw2, x, d):
        ld, lq, g = self.mixture(packed, logw, x, d, want_ld=True, want_grad=True)
        _, bg, _ = self.mixture(packed, logw2, x, d)
        return ld, lq, g, bg

    def concat(self, parts):
        return np.concatenate([np.asarray(p).reshape(-1) for p in parts])

    def unpack(self, gathered, n_ranks, sizes):
        g = np.asarray(gathered).reshape(n_ranks, int(sum(sizes)))
        outs, off = [], 0
        for sz in sizes:
            outs.append(np.ascontiguousarray(g[:, off:off + sz]).reshape(-1))
            off += sz
        return outs

    def combine(self, lp_parts, grad_parts, d):
        lp = logsumexp(lp_parts, axis=0)
        g = None if grad_parts is None else np.einsum('rn,rnd->nd', np.exp(lp_parts - lp[None]), grad_parts)
        return lp, g

    def component_stepsize(self, steps, prev, last, c):
        steps[...] = osteps.component_stepsize_improvement(steps, np.stack([prev, last], 1), c["min_stepsize"],
                                                           c["max_stepsize"], c["stepsize_inc_factor"],
                                                           c["stepsize_dec_factor"])

    def stein(self, packed, x, ld, qgrad, bg, tgrad, d):
        hs, gs = [], []
        for i in range(packed[0].shape[0]):
            g, h = ostein.expected_gradient_and_hessian_self_normalized(packed[1][i], packed[0][i], ld[i], x, bg,
                                                                        tgrad - qgrad)
            hs.append(-h); gs.append(-g)
        return np.stack(hs), np.stack(gs)

    def update_kl(self, means, chols, h_neg, g_neg, steps, temperature, l2_init, last_eta, l2, nupd):
        class W:
            pass
        w = W()
        w.model = self._model((means.copy(), chols.copy()), np.zeros(means.shape[0]))
        w.model.replace_components = lambda m, c: (setattr(w.model, "means", m), setattr(w.model, "chol_cov", c))
        w.last_log_etas, w.l2_regularizers, w.num_received_updates = last_eta.copy(), l2.copy(), nupd.copy()
        w.initial_regularizer = l2_init
        succ, _, _, _ = oupd.apply_ng_update_kl(w, h_neg, g_neg, steps, temperature)
        means[...] = w.model.means; chols[...] = w.model.chol_cov
        last_eta[...] = w.last_log_etas; l2[...] = w.l2_regularizers; nupd[...] = w.num_received_updates
        return succ, self.pack(means, chols)

    def elr(self, ld, bg, tlp, logq, beta, logw_loc):
        lw = ld - bg[None]
        iw = np.exp(lw - logsumexp(lw, axis=1, keepdims=True))
        e = iw @ (tlp - beta * logq)
        return e, beta * logw_loc + e

    def weight_stepsize(self, logw, rewards_last, state, c):
        w = np.exp(logw)
        with np.errstate(over="ignore"):
            elbo = float(np.sum(w * rewards_last) - np.sum(w * logw))
            elbo = float(np.finfo(np.float32).min) if elbo <= -3.4028e38 else float(np.float32(elbo))
        if elbo > state[1]:
            state[0] = min(c["stepsize_inc_factor"] * state[0], c["max_stepsize"])
        else:
            state[0] = max(c["stepsize_dec_factor"] * state[0], c["min_stepsize"])
        state[1] = elbo

    def update_weights(self, logw, e, stepsize_view, beta):
        if logw.shape[0] > 1:
            _, _, nl = oweights.weights_bracketing_search(logw.copy(), e, float(stepsize_view[0]), beta)
            logw[...] = nl - logsumexp(nl)


class GlooExchange:
    def __init__(self, rank, world, init_file):
        import torch.distributed as dist
        self.dist = dist
        dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
        self.n_ranks, self.rank = world, rank

    def allgather(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        outs = [torch.empty_like(t) for _ in range(self.n_ranks)]
        self.dist.all_gather(outs, t)
        return np.concatenate([o.numpy() for o in outs], axis=0)

    def barrier(self):
        self.dist.barrier()

    def max_scalar(self, v):
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])


KIND, D, K, S, SEED, ITERS = "stm", 4, 6, 24, 13, 5


def _worker(rank, world, init_file, out_dir):
    from gmmvi_amd.sharded import ShardedGMMVI
    cfg = samtron_config(S)
    o = make_oracle(KIND, D, K, S, SEED, cfg)              # only used for the target and the initial mixture
    ex = GlooExchange(rank, world, init_file)
    kl = K // world
    om = o.model.model
    algo = ShardedGMMVI(OracleOps(o.target), ex, D, K, om.means[rank * kl:(rank + 1) * kl],
                        om.chol_cov[rank * kl:(rank + 1) * kl], S, SEED, cfg)
    for _ in range(ITERS):
        algo.train_iter()
    lw, means, chols = algo.gather_model()
    assert ex.max_scalar(float(rank)) == world - 1
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), lw=lw, means=means, chols=chols,
                 steps=ex.allgather(algo.stepsizes), etas=ex.allgather(algo.last_eta), wstate=algo.wstate)
    else:
        ex.allgather(algo.stepsizes); ex.allgather(algo.last_eta)
    ex.barrier()
    ex.dist.destroy_process_group()


def test_sharded_orchestration_matches_unsharded_oracle():
    # plain multiprocessing (spawn): torch is imported only inside the two workers, never in the pytest process,
    # which may already hold libgmmvi_hip.so (two HIP runtimes in one process abort at exit)
    import multiprocessing as mp
    mpc = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rdzv")
        procs = [mpc.Process(target=_worker, args=(r, 2, init_file, tmp)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        got = np.load(os.path.join(tmp, "sharded.npz"))
        cfg = samtron_config(S)
        ref = make_oracle(KIND, D, K, S, SEED, cfg)
        for _ in range(ITERS):
            ref.train_iter()
        np.testing.assert_allclose(got["means"], ref.model.means, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(got["chols"], ref.model.chol_cov, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(got["lw"], ref.model.log_weights, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(got["steps"], ref.model.stepsizes, rtol=1e-12)
        np.testing.assert_allclose(got["etas"], ref.model.last_log_etas, rtol=1e-8)
        np.testing.assert_allclose(got["wstate"][0], ref.weight_stepsize_adapter.stepsize, rtol=1e-12)


def test_single_rank_sharded_equals_unsharded_oracle():
    from gmmvi_amd.sharded import ShardedGMMVI, LocalExchange
    cfg = samtron_config(S)
    o = make_oracle(KIND, D, K, S, SEED, cfg)
    om = o.model.model
    algo = ShardedGMMVI(OracleOps(o.target), LocalExchange(), D, K, om.means.copy(), om.chol_cov.copy(), S, SEED, cfg)
    for _ in range(ITERS):
        algo.train_iter()
        o.train_iter()
    with pytest.raises(RuntimeError):
        algo.log_weights                                   # the last weight step still rides with the next exchange
    algo.flush()
    np.testing.assert_allclose(algo.means, o.model.means, rtol=1e-10)
    np.testing.assert_allclose(algo.log_weights, o.model.log_weights, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("change,word", [
    (dict(adaptive={"del_iters": 6, "add_iters": 3, "max_components": 6, "thresholds_for_add_heuristic": [50.],
                    "min_weight_for_del_heuristic": 1e-6, "num_database_samples": 10, "num_prior_samples": 0}), "adaptive"),
    (dict(reuse_ratio=2.0), "reuse"), (dict(estimator="MORE"), "MORE"), (dict(updater="direct"), "direct"),
    (dict(diag=True), "diagonal")])
def test_sharded_path_refuses_configurations_outside_its_scope(change, word):
    """The fixed-block sharded path says what it does not cover instead of running something else (an adaptive number of
    components -- component_adaptation.py:177-300 -- has its own class, gmmvi_amd/sharded_adaptive.py)."""
    from gmmvi_amd.sharded import ShardedGMMVI, LocalExchange
    cfg = samtron_config(S, **change)
    o = make_oracle(KIND, D, K, S, SEED, samtron_config(S))
    om = o.model.model
    with pytest.raises(NotImplementedError, match=word):
        ShardedGMMVI(OracleOps(o.target), LocalExchange(), D, K, om.means.copy(), om.chol_cov.copy(), S, SEED, cfg)


def test_partition_tables_restore_the_global_order_through_adds_and_deletions():
    """Host logic of the adaptive sharded path (gmmvi_amd/sharded_adaptive.py): components are owned by id (initial blocks, then
    id mod R), deletions leave the ranks with different counts.  Concatenating every rank's local list (what an all-gather
    delivers, padding stripped) and gathering with ``rm_of_g`` must give the global (ascending id) order at every step."""
    from gmmvi_amd.sharded_adaptive import partition_tables
    rng = np.random.default_rng(4)
    for n_ranks in (2, 3, 8):
        k0 = 2 * n_ranks + 1
        ids = np.arange(k0)
        owner = (np.arange(k0) * n_ranks // k0).astype(np.int32)
        next_id = k0
        for step in range(60):
            if step % 3 != 2 or len(ids) <= n_ranks + 2:                                 # add (gmm_wrapper.py:106: ids grow by one)
                ids, owner = np.append(ids, next_id), np.append(owner, np.int32(next_id % n_ranks))
                next_id += 1
            else:                                                                        # delete anywhere but a rank's last one
                cand = [g for g in range(len(ids)) if np.count_nonzero(owner == owner[g]) > 1]
                g = int(rng.choice(cand))
                ids, owner = np.delete(ids, g), np.delete(owner, g)
            tables = [partition_tables(owner, n_ranks, r) for r in range(n_ranks)]
            counts, _, rm_of_g = tables[0]
            assert counts.sum() == len(ids) and all(np.array_equal(t[2], rm_of_g) for t in tables)
            rank_major = np.concatenate([ids[tables[r][1]] for r in range(n_ranks)])       # every rank sends its ids in local order
            assert all(len(tables[r][1]) == counts[r] for r in range(n_ranks))
            np.testing.assert_array_equal(rank_major[rm_of_g], ids)
            assert np.all(np.diff(ids) > 0)
            # the padded exchange itself, emulated on the host: every rank packs [values of width 3 | values of width 1] of its
            # components, pads to the largest rank, the buffers are concatenated (= all-gather) and taken apart by gather_layout
            from gmmvi_amd.sharded_adaptive import gather_layout
            a_of = lambda i: np.stack([i * 10.0, i * 10.0 + 1, i * 10.0 + 2], axis=1).reshape(-1)      # width 3, from the id
            sent = []
            for r in range(n_ranks):
                mine = ids[tables[r][1]].astype(np.float64)
                chunk, pad, where = gather_layout(counts, [3, 1], r)
                buf = np.concatenate([a_of(mine), -mine, np.zeros(pad)])
                assert buf.shape[0] == chunk
                sent.append(buf)
            gathered = np.concatenate(sent)
            for j, (w, expect) in enumerate([(3, a_of(ids.astype(np.float64))), (1, -ids.astype(np.float64))]):
                rm = np.concatenate([gathered[lo:lo + n] for lo, n in where[j]]).reshape(len(ids), w)
                np.testing.assert_array_equal(rm[rm_of_g].reshape(-1), expect)
