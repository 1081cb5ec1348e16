"""Component-sharded train_iter with an ADAPTIVE number of components (both multi-GPU BASELINE examples run
``num_component_adapter_type: adaptive`` -- examples/6_samtron_planar4.py:19-26 adds a component every iteration).

``sharded.ShardedGMMVI`` keeps a fixed contiguous block of components per rank.  Here the partition follows the components'
unique ids (gmm_wrapper.py:76,:106): an initial component belongs to the rank of its block, a component added later to rank
``id % R`` -- round robin, since ids grow by one per add -- and a deletion only removes a row on its owner.  No component ever
moves between GPUs.  What the reference keeps per mixture and not per component is REPLICATED and updated identically on every
rank from all-gathered values: the log-weights, the reward / weight histories the deletion rule reads
(component_adaptation.py:261-300), the weight stepsize state, and the candidate samples of the add heuristic
(component_adaptation.py:228-249: every rank holds all N samples after the first exchange anyway and draws the same candidates
from the same host generator).  So every rank takes the same add / delete decisions without a broadcast.

Order of the replicated arrays = the single-GPU order (ascending id).  Exchanged arrays arrive rank-major; a row gather with a
small index table restores the global order, so a run on R ranks sees the samples of the single-GPU run on the same seed (the
Philox normals are generated for the global draw and the local rows picked out of them).  Ranks hold different numbers of
components: every exchange pads the per-rank part to the largest one (RCCL has no all-gather-v).

Exchanges per iteration (one all-gather each): E1 samples / target values of the local draws; E2 partial background and model
densities + gradient; E3 partial post-update density; E4 expected log-ratios and rewards of the local components (the fixed-K
path lets these ride with the next E1; here the adaptation step needs the new weights and rewards now).  The add heuristic
costs two more small ones (partial candidate densities, component entropies).

Scope: Stein estimator, KL trust-region component and weight updates, reuse ratio 0, full covariances -- the configuration of
the two multi-GPU examples.  Module-by-module launches (the single-call phases of ``ShardedGMMVI`` assume a fixed K).
"""
import numpy as np

from . import hip_ops
from .optimization.gmmvi_modules.component_adaptation import VipsComponentAdaptation
from .optimization.sample_db import SampleDB, _Growable
from .sharded import HipOps, ShardedGMMVI

FLOAT32_MIN = float(np.finfo(np.float32).min)


def partition_tables(owner, n_ranks, rank):
    """Host tables of a partition by owner (``owner[g]`` = rank of the component at global position g):
    counts [R], the global positions of ``rank``'s components (ascending), and for every global position its index in the
    rank-major concatenation of the ranks' component lists -- what turns an all-gathered array back into the global order."""
    owner = np.asarray(owner)
    counts = np.bincount(owner, minlength=n_ranks)
    base = np.concatenate([[0], np.cumsum(counts)])[:-1]
    local_pos = np.zeros(owner.shape[0], np.int64)
    for r in range(n_ranks):
        sel = owner == r
        local_pos[sel] = np.arange(int(sel.sum()))
    return counts, np.nonzero(owner == rank)[0].astype(np.int32), (base[owner] + local_pos).astype(np.int32)


def gather_layout(counts, widths, rank):
    """Layout of one padded all-gather of per-component arrays: every rank sends its arrays back to back (``counts[r]`` rows of
    ``widths[j]`` floats for array j) padded to the largest rank's size.  -> (floats per rank in the gathered buffer, padding this
    rank appends, and for every array the (offset, length) of each rank's piece in the gathered buffer, rank by rank)."""
    counts = [int(c) for c in counts]
    total_w = sum(widths)
    chunk = max(counts) * total_w
    where = []
    for j, w in enumerate(widths):
        before = sum(widths[:j])
        where.append([(r * chunk + counts[r] * before, counts[r] * w) for r in range(len(counts))])
    return chunk, (max(counts) - counts[rank]) * total_w, where


class _CandidateStore:
    """What the add heuristic needs of the sample database (sample_db.py:137-152): all samples so far with their target
    log-densities, replicated on every rank, and the counter of written samples."""

    def __init__(self, ctx, d):
        self.ctx = ctx
        self._samples = _Growable(ctx, (d,))
        self._target_lnpdfs = _Growable(ctx, ())
        self._num_samples_written = 0

    samples = property(lambda self: self._samples.view())
    target_lnpdfs = property(lambda self: self._target_lnpdfs.view())
    num_samples_written = property(lambda self: SampleDB._Counter(self))
    get_random_sample = SampleDB.get_random_sample

    def append(self, x, tlp):
        self._samples.append(x)
        self._target_lnpdfs.append(tlp)
        self._num_samples_written += int(x.shape[0])


class ShardedAdaptiveGMMVI:
    def __init__(self, ops: HipOps, exchange, d, means0, chols0, samples_per_component, seed, cfg, history_length=None):
        """``means0`` [K0, D] / ``chols0`` [K0, D, D]: the WHOLE initial mixture on the host (every rank passes the same
        arrays and keeps its block)."""
        self._check_scope(cfg)
        self.ops, self.exchange, self.ctx = ops, exchange, ops.ctx
        self.R, self.rank = exchange.n_ranks, exchange.rank
        self.d, self.S, self.seed = int(d), int(samples_per_component), seed
        self.num_dimensions, self.diagonal_covs = self.d, False
        means0, chols0 = np.asarray(means0, np.float32), np.asarray(chols0, np.float32)
        k0 = means0.shape[0]
        if k0 < self.R:
            raise ValueError("fewer initial components than ranks")
        self.temperature = float(cfg["temperature"])
        self.cs = cfg["component_stepsize_adapter_config"]
        self.ws = cfg["weight_stepsize_adapter_config"]
        nca = cfg["num_component_adapter_config"]
        self.H = int(history_length) if history_length else 2 * max(2, int(nca["del_iters"]))
        ctx = self.ctx
        # replicated bookkeeping (host): ids in global order, their owners
        self.unique_component_ids = np.arange(k0, dtype=np.int32)
        self._owner = (np.arange(k0) * self.R // k0).astype(np.int32)             # contiguous blocks, as ShardedGMMVI
        self.max_component_id = k0 - 1
        self._logw = ctx.asarray(np.full(k0, -np.log(k0), np.float32))
        self.reward_history = np.full((k0, self.H), FLOAT32_MIN, np.float32)
        self.weight_history = np.full((k0, self.H), FLOAT32_MIN, np.float32)
        self.wstate = ctx.asarray(np.array([self.ws["initial_stepsize"], FLOAT32_MIN], np.float32))
        # local components
        mine = np.nonzero(self._owner == self.rank)[0]
        self.means, self.chols = ctx.asarray(means0[mine]), ctx.asarray(chols0[mine])
        kl = len(mine)
        self.initial_stepsize = float(self.cs["initial_stepsize"])
        self.initial_regularizer = float(cfg["ng_estimator_config"].get("initial_l2_regularizer", 1e-12))
        self.stepsizes = ctx.full((kl,), self.initial_stepsize)
        self.last_eta = ctx.full((kl,), -1.0)
        self.l2 = ctx.full((kl,), self.initial_regularizer)
        self.num_received_updates = ctx.full((kl,), 0.0)
        self.packed = None
        self.last_success = None
        self.num_updates = 0
        self.store = _CandidateStore(ctx, self.d)
        self._tables = None
        init = cfg.get("model_initialization", {})
        self.adapter = VipsComponentAdaptation(self, self.store, ops.target, init.get("prior_mean"), init.get("initial_cov"),
                                               **nca)

    @staticmethod
    def _check_scope(cfg):
        fixed = dict(cfg, num_component_adapter_type="fixed")
        ShardedGMMVI._check_scope(fixed)                   # same estimator / updater / reuse / covariance scope
        if cfg.get("num_component_adapter_type") != "adaptive":
            raise ValueError("ShardedAdaptiveGMMVI is the adaptive-K path; use ShardedGMMVI for a fixed number of components")
        if cfg.get("weight_updater_type", "trust-region") != "trust-region":
            raise NotImplementedError("ShardedAdaptiveGMMVI: trust-region weight updates only")

    # ---- partition tables (host, rebuilt when K changes) ------------------------------------------------------------------
    @property
    def num_components(self):
        return int(self.unique_component_ids.shape[0])

    def _tab(self):
        if self._tables is None:
            counts, gpos_loc, rm_of_g = partition_tables(self._owner, self.R, self.rank)
            if counts.min() == 0:
                raise RuntimeError("ShardedAdaptiveGMMVI: a rank is left without components (deletions emptied its share)")
            t = type("Tables", (), {})()
            t.counts, t.kmax = counts, int(counts.max())
            t.gpos_loc = gpos_loc                    # global positions of the local components
            t.rm_of_g = rm_of_g                      # rank-major index of global position g
            t.rm_of_g_dev = self.ctx.asarray(t.rm_of_g, np.int32)
            t.gpos_loc_dev = self.ctx.asarray(t.gpos_loc, np.int32)
            self._tables = t
        return self._tables

    def _gather_by_component(self, parts):
        """``parts``: [(device array holding K_local rows of ``width`` floats, width)] -> the same arrays for ALL components in
        global order ([K * width] each).  One all-gather; per-rank parts padded to the largest rank."""
        if self.R == 1:
            return [a.reshape(-1) for a, _ in parts]
        ctx, t = self.ctx, self._tab()
        widths = [w for _, w in parts]
        chunk, pad, where = gather_layout(t.counts, widths, self.rank)
        pieces = [a.reshape(-1) for a, _ in parts]
        if pad:
            pieces.append(ctx.zeros((pad,)))
        gathered = self.exchange.allgather(hip_ops.concat(ctx, pieces))
        out = []
        for j, w in enumerate(widths):
            rank_major = hip_ops.concat(ctx, [gathered.rows(lo, lo + n) for lo, n in where[j]]).reshape((self.num_components, w))
            out.append(hip_ops.gather_rows(ctx, rank_major, t.rm_of_g_dev).reshape(-1))
        return out

    def _combine(self, lp_part, grad_part=None):
        """Partial mixtures over the local components -> the full mixture (one all-gather of equal parts)."""
        if self.R == 1:
            return lp_part, grad_part
        ctx, d, n = self.ctx, self.d, lp_part.shape[0]
        if grad_part is None:
            g = self.exchange.allgather(lp_part)
            return hip_ops.combine_partials(ctx, g.reshape((self.R, n)), None, d)[0], None
        g = self.exchange.allgather(hip_ops.concat(ctx, [lp_part, grad_part]))
        lp, gr = hip_ops.unpack_gathered(ctx, g, self.R, [n, n * d])
        return hip_ops.combine_partials(ctx, lp.reshape((self.R, n)), gr.reshape((self.R, n, d)), d)

    def _local(self, global_dev):
        """Rows of a replicated [K] device array that belong to the local components."""
        if self.R == 1:
            return global_dev
        return hip_ops.gather_rows(self.ctx, global_dev, self._tab().gpos_loc_dev)

    def _packed_now(self):
        if self.packed is None:
            self.packed = self.ops.pack(self.means, self.chols)
        return self.packed

    # ---- one iteration (gmmvi.py:146-174, then component_adaptation.py:177-190) ------------------------------------------------
    def train_iter(self):
        ctx, o, d, S = self.ctx, self.ops, self.d, self.S
        t = self._tab()
        K, kl = self.num_components, int(t.counts[self.rank])
        N, nl = K * S, kl * S
        # draw: Philox normals of the GLOBAL draw (sample n of the run has one counter whatever the partition), local rows of it
        first = int(self.store.num_samples_written)
        eps = hip_ops.philox_normals(ctx, self.seed, first, N, d)
        if self.R > 1:
            eps = hip_ops.gather_rows(ctx, eps.reshape((K, S * d)), t.gpos_loc_dev).reshape((nl, d))
        offsets = ctx.cached_const(("offsets_adaptive", kl, S), lambda: ctx.asarray(np.arange(kl + 1, dtype=np.int32) * S, np.int32))
        x_loc = hip_ops.sample_components(ctx, self.means, self.chols, offsets, nl, seed=self.seed, first_index=first, eps=eps)[0]
        tlp_loc, tgrad_loc = o.target_eval(x_loc)
        x, tlp, tgrad = self._gather_by_component([(x_loc, S * d), (tlp_loc, S), (tgrad_loc, S * d)])       # E1
        x, tgrad = x.reshape((N, d)), tgrad.reshape((N, d))
        self.store.append(x, tlp)
        # partial background / model densities over the local components, E2
        packed = self._packed_now()
        logw_loc = self._local(self._logw)
        logc_loc = ctx.cached_const(("logc_adaptive", kl, K), lambda: ctx.asarray(np.full(kl, -np.log(K), np.float32)))
        ld, lq_part, qg_part, bg_part = o.mixture_dual(packed, logw_loc, logc_loc, x, d)
        bg, _ = self._combine(bg_part)
        logq, qgrad = self._combine(lq_part, qg_part)
        # component stepsizes from the last two reward columns of the local components, Stein estimate, KL-constrained update
        rh = self.reward_history[t.gpos_loc]
        o.component_stepsize(self.stepsizes, ctx.asarray(np.ascontiguousarray(rh[:, -2])), ctx.asarray(np.ascontiguousarray(rh[:, -1])),
                             self.cs)
        h_neg, g_neg = o.stein(packed, x, ld, qgrad, bg, tgrad, d)
        self.last_success, self.packed = o.update_kl(self.means, self.chols, h_neg, g_neg, self.stepsizes, self.temperature,
                                                     self.initial_regularizer, self.last_eta, self.l2, self.num_received_updates)
        # weight stepsize (replicated), post-update density (E3), expected log-ratios of the local components, E4
        o.weight_stepsize(self._logw, ctx.asarray(np.ascontiguousarray(self.reward_history[:, -1])), self.wstate, self.ws)
        ld2, lq2_part, _ = o.mixture(self.packed, logw_loc, x, d, want_ld=True)
        logq2, _ = self._combine(lq2_part)
        e_loc, reward_loc = o.elr(ld2, bg, tlp, logq2, self.temperature, logw_loc)
        e, reward = self._gather_by_component([(e_loc, 1), (reward_loc, 1)])
        self.reward_history = np.concatenate([self.reward_history[:, 1:], reward.numpy()[:, None]], axis=1)   # gmm_wrapper.py:150-158
        o.update_weights(self._logw, e, o.rows(self.wstate, 0, 1), self.temperature)
        self.weight_history = np.concatenate([self.weight_history[:, 1:], np.exp(self._logw.numpy())[:, None]], axis=1)   # :170-182
        self.num_updates += 1
        self.adapter.adapt_number_of_components(self.num_updates)

    def flush(self):
        """Nothing is pending after train_iter (kept for the interface bench.py drives ShardedGMMVI through)."""

    # ---- the model surface the adaptation module uses (gmm_wrapper.py:90-148, gmm.py) --------------------------------------------
    @property
    def log_weights(self):
        return self._logw

    def reward_window(self, n):
        return self.reward_history[:, -int(n):]

    def weight_window(self, n):
        return self.weight_history[:, -int(n):]

    def log_density(self, samples):
        """gmm.py:183-192 of the whole mixture (collective: every rank calls it with the same samples)."""
        x = self.ctx.asarray(samples)
        part = self.ops.mixture(self._packed_now(), self._local(self._logw), x, self.d)[1]
        return self._combine(part)[0]

    def component_entropies(self):
        """gmm.py:249-261 for ALL components (collective)."""
        diag = np.ascontiguousarray(np.diagonal(self.chols.numpy(), axis1=1, axis2=2))
        loc = (0.5 * self.d * (np.log(2 * np.pi) + 1) + np.sum(np.log(diag), axis=1)).astype(np.float32)
        return self._gather_by_component([(self.ctx.asarray(loc), 1)])[0].numpy()

    def get_average_entropy(self):
        """gmm.py:263-272."""
        w = np.exp(self._logw.numpy().astype(np.float64))
        return float(np.sum(w * self.component_entropies()))

    def _renormalised(self, lw):
        out = self.ctx.empty(lw.shape)
        self.ctx.check(self.ctx.lib.gmmvi_normalize_logw(self.ctx.handle, lw.ptr, int(lw.shape[0]), out.ptr))
        return out

    def add_component(self, initial_weight, initial_mean, initial_cov, adding_threshold, initial_entropy):
        """gmm_wrapper.py:90-127 / full_cov_gmm.py:64-68: the owner (new id mod R) appends the component, everybody the
        replicated entries."""
        ctx, d = self.ctx, self.d
        self.max_component_id += 1
        owner = self.max_component_id % self.R
        if owner == self.rank:
            cov = np.asarray(initial_cov, np.float32).reshape(d, d)
            if np.count_nonzero(cov - np.diag(np.diagonal(cov))) or not np.all(np.diagonal(cov) > 0):
                raise ValueError("add_component: a positive diagonal covariance is expected (component_adaptation.py:220-223)")
            chol = ctx.asarray(np.diag(np.sqrt(np.diagonal(cov))).reshape(1, d, d))
            mean = initial_mean if hasattr(initial_mean, "ptr") else ctx.asarray(np.asarray(initial_mean, np.float32).reshape(1, d))
            kl = self.means.shape[0]
            self.means = hip_ops.concat(ctx, [self.means, mean.reshape((1, d))]).reshape((kl + 1, d))
            self.chols = hip_ops.concat(ctx, [self.chols, chol]).reshape((kl + 1, d, d))
            tail = ctx.asarray(np.array([self.initial_regularizer, -1.0, 0.0, self.initial_stepsize], np.float32))
            app = lambda dev, j: hip_ops.concat(ctx, [dev, tail.rows(j, j + 1)])
            self.l2, self.last_eta = app(self.l2, 0), app(self.last_eta, 1)
            self.num_received_updates, self.stepsizes = app(self.num_received_updates, 2), app(self.stepsizes, 3)
            self.packed = None
        self.unique_component_ids = np.append(self.unique_component_ids, np.int32(self.max_component_id))
        self._owner = np.append(self._owner, np.int32(owner))
        new_lw = ctx.asarray(np.array([np.log(np.float64(initial_weight))], np.float32))
        self._logw = self._renormalised(hip_ops.concat(ctx, [self._logw, new_lw]))
        self.reward_history = np.concatenate([self.reward_history, np.full((1, self.H), FLOAT32_MIN, np.float32)])     # :121-122
        self.weight_history = np.concatenate([self.weight_history, np.full((1, self.H), initial_weight, np.float32)])  # :123-124
        self._tables = None

    def remove_component(self, idx):
        """gmm_wrapper.py:129-148 / gmm.py:388-398."""
        ctx, idx = self.ctx, int(idx)
        k_old = self.num_components
        if self._owner[idx] == self.rank:
            lpos = int(np.count_nonzero(self._owner[:idx] == self.rank))
            keep = ctx.asarray(np.delete(np.arange(self.means.shape[0], dtype=np.int32), lpos), np.int32)
            rm = lambda dev: hip_ops.gather_rows(ctx, dev, keep)
            self.means, self.chols = rm(self.means), rm(self.chols)
            self.l2, self.last_eta = rm(self.l2), rm(self.last_eta)
            self.num_received_updates, self.stepsizes = rm(self.num_received_updates), rm(self.stepsizes)
            self.packed = None
        keep_g = ctx.asarray(np.delete(np.arange(k_old, dtype=np.int32), idx), np.int32)
        self._logw = self._renormalised(hip_ops.gather_rows(ctx, self._logw, keep_g))
        self.unique_component_ids = np.delete(self.unique_component_ids, idx)
        self._owner = np.delete(self._owner, idx)
        self.reward_history = np.delete(self.reward_history, idx, axis=0)
        self.weight_history = np.delete(self.weight_history, idx, axis=0)
        self._tables = None

    # ---- bench.py factory ------------------------------------------------------------------------------------------------------
    @staticmethod
    def build(w, n_ranks, rank):
        """From bench.py's workload dict (the whole initial mixture on every rank) and an RCCL exchange."""
        from .device import get_context
        from .sharded import RcclExchange, LocalExchange
        ctx = get_context()
        chols, ok = hip_ops.cholesky(ctx, ctx.asarray(w["covs"]))
        exchange = RcclExchange(ctx, n_ranks, rank) if n_ranks > 1 else LocalExchange()
        return ShardedAdaptiveGMMVI(HipOps(ctx, w["target"]), exchange, w["d"], w["means"], chols.numpy(), w["s"], w["seed"],
                                    w["cfg"])

    # ---- host views (tests / metrics) ---------------------------------------------------------------------------------------
    def gather_model(self):
        """(log_weights [K], means [K, D], chols [K, D, D]) on the host in global order (collective)."""
        d = self.d
        m, c = self._gather_by_component([(self.means, d), (self.chols, d * d)])
        k = self.num_components
        return self._logw.numpy(), m.numpy().reshape(k, d), c.numpy().reshape(k, d, d)
