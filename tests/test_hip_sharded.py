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
