"""GPU tests of the sharded path: single-rank HipOps/LocalExchange equals the modular GMMVI on the same seed, and the
RCCL communicator initialises and moves data (one rank per GPU; multi-rank runs happen in the driver's scaling bench)."""
import os
import numpy as np
import pytest

from helpers import samtron_config, make_oracle, make_device

pytestmark = pytest.mark.gpu


def test_single_rank_sharded_equals_modular_gmmvi():
    from gmmvi_amd.device import get_context
    from gmmvi_amd.sharded import ShardedGMMVI, HipOps, LocalExchange
    kind, d, k, s, seed = "stm", 6, 8, 40, 17
    cfg = samtron_config(s)
    o = make_oracle(kind, d, k, s, seed, cfg)
    g = make_device(kind, d, k, s, seed, cfg, o)
    ctx = get_context()
    sh = ShardedGMMVI(HipOps(ctx, g.sample_selector.target_distribution), LocalExchange(), d, k,
                      g.model.means.numpy(), g.model.chol_cov.numpy(), s, seed, cfg)
    for _ in range(6):
        g.train_iter()
        sh.train_iter()
    sh.flush()
    # not bitwise: the modular path takes the parameter blocks emitted by the update kernel, the sharded path re-packs
    # them (log-normaliser summed in a different order); 6 iterations amplify the 1e-7 difference to ~1e-5
    np.testing.assert_allclose(sh.means.numpy(), g.model.means.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(sh.chols.numpy(), g.model.chol_cov.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(sh.log_weights.numpy(), g.model.log_weights.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(sh.stepsizes.numpy(), g.model.stepsizes.numpy(), rtol=1e-6)


def test_rccl_single_rank_communicator():
    from gmmvi_amd.device import get_context
    from gmmvi_amd.sharded import RcclExchange
    ctx = get_context()
    ex = RcclExchange(ctx, 1, 0, tag=f"test_{os.getpid()}")
    a = ctx.asarray(np.arange(12, dtype=np.float32).reshape(4, 3))
    np.testing.assert_array_equal(ex.allgather(a).numpy(), a.numpy())
    assert ex.max_scalar(3.5) == 3.5
    ctx.check(ctx.lib.gmmvi_comm_destroy(ctx.handle))


def test_concat_unpack_gathered_roundtrip():
    from gmmvi_amd.device import get_context
    from gmmvi_amd import hip_ops
    ctx = get_context()
    rng = np.random.default_rng(3)
    ranks, sizes = 3, [35, 7, 70]
    parts = [[rng.normal(size=sz).astype(np.float32) for sz in sizes] for _ in range(ranks)]
    chunks = [hip_ops.concat(ctx, [ctx.asarray(p) for p in ps]).numpy() for ps in parts]
    for ps, c in zip(parts, chunks):
        np.testing.assert_array_equal(c, np.concatenate(ps))
    outs = hip_ops.unpack_gathered(ctx, ctx.asarray(np.concatenate(chunks)), ranks, sizes)
    for j, o in enumerate(outs):
        np.testing.assert_array_equal(o.numpy(), np.concatenate([parts[r][j] for r in range(ranks)]))


class _ThreadExchange:
    """Two virtual ranks in ONE process on ONE GPU: each rank is a thread, a turn lock lets exactly one of them issue
    work at a time (they share the context and its stream), and the exchange goes through the host.  Exercises the
    device-side packing, de-interleaving and partial-mixture recombination of the R > 1 code path that the single-rank
    tests never reach; RCCL itself is only initialised with one rank here."""

    def __init__(self, ctx, rank, world, slots, barrier, turn):
        self.ctx, self.rank, self.n_ranks, self.slots, self.barrier, self.turn = ctx, rank, world, slots, barrier, turn

    def _rendezvous(self):
        self.turn.release()
        self.barrier.wait()
        self.turn.acquire()

    def allgather(self, arr):
        self.slots[self.rank] = arr.numpy().reshape(-1)
        self._rendezvous()
        out = np.concatenate(self.slots)
        self._rendezvous()
        shape = (self.n_ranks * arr.shape[0],) + tuple(arr.shape[1:])
        return self.ctx.asarray(out.reshape(shape))

    def allgather_inplace(self, buf, count):
        """The in-place form the phased C iteration uses (gmmvi_train_iter_sharded_phase): own part -> host, all parts back."""
        self.slots[self.rank] = buf.rows(self.rank * count, (self.rank + 1) * count).numpy()
        self._rendezvous()
        out = np.concatenate(self.slots)
        self._rendezvous()
        buf.set(out)

    def barrier_(self):
        self._rendezvous()

    def max_scalar(self, v):
        return v


@pytest.mark.parametrize("kind,d,k,s,iters", [("stm", 6, 8, 40, 5), ("gmm", 72, 4, 48, 3), ("stm", 20, 32, 64, 4),
                                              ("planar", 10, 8, 50, 4)])
def test_two_virtual_ranks_match_single_rank(kind, d, k, s, iters):
    """D = 6 / 20 / 10: the phased C iteration (gmmvi_train_iter_sharded_phase: the single-call iteration's launches with the
    three exchanges between its phases; 16 local components at D = 20, N = 2048: component-chunked sweeps, their partials merged
    before they travel); D = 72: blocked
    kernels, module-by-module path."""
    import threading
    from gmmvi_amd.device import get_context
    from gmmvi_amd.sharded import ShardedGMMVI, HipOps, LocalExchange
    from gmmvi_amd import hip_ops
    seed = 23
    cfg = samtron_config(s)
    o = make_oracle(kind, d, k, s, seed, cfg)
    g = make_device(kind, d, k, s, seed, cfg, o)
    target = g.sample_selector.target_distribution
    means0, chols0 = g.model.means.numpy(), g.model.chol_cov.numpy()
    ref = ShardedGMMVI(HipOps(get_context(), target), LocalExchange(), d, k, means0, chols0, s, seed, cfg)
    for _ in range(iters):
        ref.train_iter()
    ref.flush()

    world, slots, barrier, turn = 2, [None, None], threading.Barrier(2), threading.Lock()
    results, errors = [None, None], []
    ctx = get_context()

    def run(rank):
        turn.acquire()
        try:
            kl = k // world
            sh = ShardedGMMVI(HipOps(ctx, target), _ThreadExchange(ctx, rank, world, slots, barrier, turn), d, k,
                              means0[rank * kl:(rank + 1) * kl], chols0[rank * kl:(rank + 1) * kl], s, seed, cfg)
            for _ in range(iters):
                sh.train_iter()
            sh.flush()
            results[rank] = (sh.log_weights.numpy(), sh.means.numpy(), sh.chols.numpy())
        except Exception as e:                                      # pragma: no cover - surfaced below
            errors.append(e)
            barrier.abort()
        finally:
            turn.release()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    logw = results[0][0]
    np.testing.assert_array_equal(logw, results[1][0])              # replicated state stays identical on both ranks
    means = np.concatenate([results[0][1], results[1][1]])
    chols = np.concatenate([results[0][2], results[1][2]])
    # partial log-sum-exps are recombined in a different order than the single-rank sweep
    np.testing.assert_allclose(logw, ref.log_weights.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(means, ref.means.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(chols, ref.chols.numpy(), rtol=2e-4, atol=2e-4)


# ---- adaptive number of components on shards (gmmvi_amd/sharded_adaptive.py) ------------------------------------------------
# (tools/debug_sharded_adaptive.py: on this case 9 components are deleted in 45 iterations, K ends at the cap of 14, and several
# of the added components carry weight -- 0.34 / 0.25 / 0.07 ...)
ADAPTIVE = {"del_iters": 6, "add_iters": 2, "max_components": 14, "thresholds_for_add_heuristic": [500., 100., 20.],
            "min_weight_for_del_heuristic": 1e-3, "num_database_samples": 300, "num_prior_samples": 0}
ADAPTIVE_CASE = ("stm", 6, 2, 60, 3, 45)                  # target, D, initial K, samples / component, seed, iterations


def _assert_same_mixture(a, b):
    """Weights as probabilities (log-weights near the -69 floor of a component that carries nothing are noise), parameters of
    the components that carry weight (one without weight follows its own sampling noise)."""
    (lwa, ma, ca), (lwb, mb, cb) = a, b
    np.testing.assert_allclose(np.exp(lwa), np.exp(lwb), rtol=2e-3, atol=1e-6)
    heavy, live = np.exp(lwb) > 0.01, np.exp(lwb) > 1e-4
    assert heavy.any()
    scale = np.abs(mb).max()                               # (45 iterations of re-sampling from slightly different mixtures)
    np.testing.assert_allclose(ma[heavy], mb[heavy], rtol=2e-3, atol=2e-3 * scale)
    np.testing.assert_allclose(ca[heavy], cb[heavy], rtol=2e-3, atol=2e-3 * scale)
    assert np.abs(ma[live] - mb[live]).max() <= 0.02 * scale and np.abs(ca[live] - cb[live]).max() <= 0.02 * scale

def _adaptive_setup(kind, d, k, s, seed):
    cfg = samtron_config(s, adaptive=ADAPTIVE)
    o = make_oracle(kind, d, k, s, seed, cfg)
    g = make_device(kind, d, k, s, seed, cfg, o)          # also gives the product's target object and the initial mixture
    cfg = dict(cfg, model_initialization=dict(cfg["model_initialization"], prior_mean=0.0,
                                              initial_cov=g.num_component_adapter.prior_var.tolist()))
    return cfg, g


def test_single_rank_adaptive_sharded_equals_modular_gmmvi():
    """One rank of the adaptive sharded path against the module-by-module GMMVI on the same seed: the same components are
    added and deleted at the same iterations (component_adaptation.py:177-300), the mixtures agree."""
    from gmmvi_amd.device import get_context
    from gmmvi_amd.sharded import HipOps, LocalExchange
    from gmmvi_amd.sharded_adaptive import ShardedAdaptiveGMMVI
    kind, d, k, s, seed, iters = ADAPTIVE_CASE
    cfg, g = _adaptive_setup(kind, d, k, s, seed)
    g._fast_path.enabled = False
    sh = ShardedAdaptiveGMMVI(HipOps(get_context(), g.sample_selector.target_distribution), LocalExchange(), d,
                              g.model.means.numpy(), g.model.chol_cov.numpy(), s, seed, cfg, history_length=400)
    deleted = 0
    for it in range(iters):
        before = set(sh.unique_component_ids.tolist())
        g.train_iter()
        sh.train_iter()
        deleted += len(before - set(sh.unique_component_ids.tolist()))
        np.testing.assert_array_equal(sh.unique_component_ids, g.model.unique_component_ids, err_msg=f"iteration {it}")
    assert deleted >= 3 and sh.num_components > k, (deleted, sh.num_components)
    added_alive = np.exp(sh.log_weights.numpy())[sh.unique_component_ids >= k]
    assert np.count_nonzero(added_alive > 0.01) >= 2, added_alive
    lw, means, chols = sh.gather_model()
    _assert_same_mixture((lw, means, chols), (g.model.log_weights.numpy(), g.model.means.numpy(), g.model.chol_cov.numpy()))


def test_two_virtual_ranks_adaptive_match_single_rank():
    """Two virtual ranks (threads, one GPU, exchange through the host): components added later go to rank id mod 2, deletions
    leave the ranks with different numbers of components -- padded exchanges, the rank-major -> global order gather.  Same
    ids after every iteration and the same mixture as one rank."""
    import threading
    from gmmvi_amd.device import get_context
    from gmmvi_amd.sharded import HipOps, LocalExchange
    from gmmvi_amd.sharded_adaptive import ShardedAdaptiveGMMVI
    kind, d, k, s, seed, iters = ADAPTIVE_CASE
    cfg, g = _adaptive_setup(kind, d, k, s, seed)
    target = g.sample_selector.target_distribution
    means0, chols0 = g.model.means.numpy(), g.model.chol_cov.numpy()
    ctx = get_context()
    ref = ShardedAdaptiveGMMVI(HipOps(ctx, target), LocalExchange(), d, means0, chols0, s, seed, cfg, history_length=400)
    ref_ids = []
    for _ in range(iters):
        ref.train_iter()
        ref_ids.append(ref.unique_component_ids.copy())
    ref_model = ref.gather_model()

    world, slots, barrier, turn = 2, [None, None], threading.Barrier(2), threading.Lock()
    results, errors, uneven = [None, None], [], [False]

    def run(rank):
        turn.acquire()
        try:
            sh = ShardedAdaptiveGMMVI(HipOps(ctx, target), _ThreadExchange(ctx, rank, world, slots, barrier, turn), d,
                                      means0, chols0, s, seed, cfg, history_length=400)
            for it in range(iters):
                sh.train_iter()
                np.testing.assert_array_equal(sh.unique_component_ids, ref_ids[it], err_msg=f"rank {rank}, iteration {it}")
                counts = np.bincount(sh._owner, minlength=world)
                uneven[0] = uneven[0] or counts[0] != counts[1]
            results[rank] = sh.gather_model()
        except BaseException as e:                                   # pragma: no cover - surfaced below
            errors.append(e)
            barrier.abort()
        finally:
            turn.release()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert uneven[0], "the ranks never held different numbers of components: the padded exchange was not exercised"
    for a, b in zip(results[0], results[1]):
        np.testing.assert_array_equal(a, b)
    _assert_same_mixture(results[0], ref_model)


def test_adaptive_sharded_on_the_example6_workload():
    """bench.py's c4_adaptive workload (examples/6_samtron_planar4.py:19-26: a component added every iteration, deletions from
    iteration 11) through ShardedAdaptiveGMMVI.build at one rank: K follows the adds, components get deleted, and the deletion
    decisions are the reference rule's on the replicated histories (the adaptation module is the single-GPU one)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from gmmvi_amd.sharded_adaptive import ShardedAdaptiveGMMVI
    w = bench.build("c4_adaptive", 1, 0)
    sh = ShardedAdaptiveGMMVI.build(w, 1, 0)
    k0, ids_seen = sh.num_components, set(sh.unique_component_ids.tolist())
    for _ in range(30):
        sh.train_iter()
        ids_seen |= set(sh.unique_component_ids.tolist())
    assert sh.max_component_id == k0 - 1 + 29                      # adds at iterations 2..30
    deleted = len(ids_seen) - sh.num_components
    assert deleted >= 1 and sh.num_components == k0 + 29 - deleted
    assert np.isfinite(sh.log_weights.numpy()).all() and abs(np.exp(sh.log_weights.numpy().astype(np.float64)).sum() - 1) < 1e-4
