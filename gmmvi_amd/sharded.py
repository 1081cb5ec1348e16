"""Component-sharded train_iter over the GPUs of one node (the reference has no multi-GPU code).

Partition: rank r owns components [r*Kl, (r+1)*Kl) -- their means, Cholesky factors, stepsizes, etas and the
samples they draw.  Every component needs all N samples (``only_use_own_samples: False``), so one iteration has
three exchange steps, each an RCCL all-gather over xGMI followed by a local combine:

  E1  samples, target log-densities and target gradients of the locally drawn samples      [N/R, 2D+1] per rank
  E2  per-sample partial mixtures over the local components: background (max,sum) folded into one log value,
      model log q partial and its responsibility-weighted gradient partial                  [N, D+2]   per rank
  E3  post-update log q partial                                                              [N]        per rank
Each step is ONE all-gather: the arrays of a step travel back to back in one buffer (xGMI collectives at these sizes are
latency-bound, ~tens of microseconds each whatever the payload) and one de-interleave launch restores them.
The per-component expected log-ratios and rewards of the weight step ([2 Kl] per rank) need no collective of their own:
nothing reads the new weights or the new reward column before the NEXT iteration's density sweep and stepsize rule, so
they ride in the next iteration's E1 buffer and the replicated weight step is applied right after that gather
(``flush()`` sends a pending pair on its own -- call it on every rank before reading ``log_weights``).

Stein estimate and the KL-constrained component update are local to the owner.  The mixture weights [K], the
reward history and the weight trust-region step are replicated (every rank computes the identical [K]-sized update
from the gathered expected log-ratios), so no broadcast is needed.  Philox sample indices are global (rank offset
folded into the counter), hence a sharded run draws exactly the samples of the single-GPU run on the same seed.

Scope: the SAMTRON hot path with a fixed number of components and reuse ratio 0 (the configuration bench.py scales).
The orchestration is written against a small ``ops`` / ``exchange`` interface so that the exchange logic is
exercised by world-size-2 gloo tests on CPU (tests/test_sharded_cpu.py); on the GPU ``HipOps`` + ``RcclExchange``.
"""
import os
import time

import numpy as np

FLOAT32_MIN = float(np.finfo(np.float32).min)


import ctypes as _C

_f, _i, _p = _C.c_float, _C.c_int32, _C.c_void_p


class ShardedPlan(_C.Structure):
    """struct gmmvi_sharded_plan (include/gmmvi_hip.h)."""
    _fields_ = [
        ("n_ranks", _i), ("rank", _i), ("K", _i), ("D", _i), ("N", _i),
        ("target_kind", _i), ("target_family", _i), ("target_K", _i), ("target_nu", _f),
        ("target_packed", _p), ("target_logw", _p), ("planar_prior_std", _p), ("planar_goals", _p),
        ("planar_goals_count", _i), ("planar_likelihood_std", _f),
        ("means", _p), ("chols", _p), ("packed", _p), ("packed_new", _p),
        ("stepsizes", _p), ("last_eta", _p), ("l2", _p), ("num_updates", _p), ("success_out", _p),
        ("logw_all", _p), ("bg_logw", _p), ("offsets", _p), ("max_per_component", _i),
        ("seed", _C.c_uint64), ("first_index", _C.c_uint64),
        ("e1", _p), ("e2", _p), ("e3", _p),
        ("x_all", _p), ("tlp_all", _p), ("tgrad_all", _p), ("E_all", _p), ("reward_all", _p),
        ("has_pending", _i), ("reward_col_pending", _p), ("reward_prev", _p), ("reward_last", _p), ("reward_last_all", _p),
        ("wstate", _p), ("temperature", _f), ("l2_init", _f),
        ("component_stepsize_mode", _i), ("cs_min", _f), ("cs_max", _f), ("cs_inc", _f), ("cs_dec", _f),
        ("weight_stepsize_mode", _i), ("ws_min", _f), ("ws_max", _f), ("ws_inc", _f), ("ws_dec", _f),
        ("stein_flags", _i), ("presample_next", _i), ("presampled", _i), ("scratch", _p),
    ]


# ---------------------------------------------------------------------------------------------------------------------
# GPU back end
# ---------------------------------------------------------------------------------------------------------------------
class HipOps:
    """hip_ops behind the interface ShardedGMMVI uses (DeviceArrays in HBM)."""

    def __init__(self, ctx, target):
        from . import hip_ops
        self.ctx, self.h, self.target = ctx, hip_ops, target

    def asarray(self, x, dtype=np.float32):
        return self.ctx.asarray(x, dtype)

    def full(self, shape, v):
        return self.ctx.full(shape, v)

    def rows(self, a, lo, hi):
        return a.rows(lo, hi)

    def to_host(self, a):
        return a.numpy()

    def copy_into(self, dst, src):
        dst.copy_from(src)

    def sample(self, means, chols, counts, seed, first_index):
        offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        od = self.ctx.cached_const(("offsets", offsets.tobytes()), lambda: self.ctx.asarray(offsets, np.int32))
        return self.h.sample_components(self.ctx, means, chols, od, int(offsets[-1]), seed=seed,
                                        first_index=first_index)[0]

    def target_eval(self, x):
        lp, grad = self.target.log_density_and_grad(x)
        return self.ctx.asarray(lp), self.ctx.asarray(grad)

    def pack(self, means, chols):
        return self.h.pack_components(self.ctx, means, chols)[0]

    def mixture(self, packed, logw, x, d, want_ld=False, want_grad=False):
        return self.h.mixture_eval(self.ctx, packed, logw, x, d, want_ld=want_ld, want_lp=True, want_grad=want_grad)

    def mixture_dual(self, packed, logw, logw2, x, d):
        """(ld, partial log q, partial gradient, partial background) in one sweep over the local components."""
        return self.h.mixture_eval_dual(self.ctx, packed, logw, logw2, x, d)

    def combine(self, lp_parts, grad_parts, d):
        return self.h.combine_partials(self.ctx, lp_parts, grad_parts, d)

    def concat(self, parts):
        return self.h.concat(self.ctx, parts)

    def unpack(self, gathered, n_ranks, sizes):
        return self.h.unpack_gathered(self.ctx, gathered, n_ranks, sizes)

    def component_stepsize(self, steps, prev, last, c):
        self.h.component_stepsize_improvement(self.ctx, steps, prev, last, c["min_stepsize"], c["max_stepsize"],
                                              c["stepsize_inc_factor"], c["stepsize_dec_factor"])

    def stein(self, packed, x, ld, qgrad, bg, tgrad, d):
        return self.h.stein(self.ctx, packed, x, ld, qgrad, bg, tgrad, d)

    def update_kl(self, means, chols, h_neg, g_neg, steps, temperature, l2_init, last_eta, l2, nupd):
        """-> (success, packed blocks of the updated components)."""
        out = self.h.update_components_kl(self.ctx, means, chols, h_neg, g_neg, steps, temperature, l2_init, last_eta,
                                          l2, nupd, want_packed=True)
        return out[0], out[3]

    def elr(self, ld, bg, tlp, logq, beta, logw_loc):
        k = ld.shape[0]
        reward = self.ctx.empty((k,))
        e, _ = self.h.expected_log_ratios(self.ctx, ld, bg, tlp, logq, beta, logw_loc, True, reward_out=reward)
        return e, reward

    def weight_stepsize(self, logw, rewards_last, state, c):
        self.h.weight_stepsize_improvement(self.ctx, logw, rewards_last, state, c["min_stepsize"], c["max_stepsize"],
                                           c["stepsize_inc_factor"], c["stepsize_dec_factor"])

    def update_weights(self, logw, e, stepsize_view, beta):
        self.h.update_weights(self.ctx, "trust-region", logw, e, stepsize_view, beta)


class RcclExchange:
    """RCCL all-gather / all-reduce over the ranks of one node (one process per GPU).  The 128-byte ncclUniqueId
    travels through a file in the node-local temp directory keyed by the launcher's MASTER_PORT / run id."""

    def __init__(self, ctx, n_ranks, rank, tag=None, timeout=120.0):
        import ctypes as C
        self.ctx, self.n_ranks, self.rank = ctx, n_ranks, rank
        # all ranks of one launch share the launcher as parent process: a stale file of an earlier run never matches
        tag = tag or f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"gmmvi_rccl_id_{tag}_{n_ranks}.bin")
        if rank == 0:
            buf = C.create_string_buffer(128)
            rc = ctx.lib.gmmvi_comm_unique_id(buf)
            if rc != 0:
                ctx.check(rc)
            tmp = path + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(buf.raw)
            os.replace(tmp, path)
            uid = buf.raw
        else:
            t0 = time.time()
            while True:
                try:
                    if os.path.getsize(path) == 128 and os.path.getmtime(path) > time.time() - 3600:
                        with open(path, "rb") as f:
                            uid = f.read()
                        break
                except OSError:
                    pass
                if time.time() - t0 > timeout:
                    raise TimeoutError(f"rank {rank}: no RCCL id at {path}")
                time.sleep(0.02)
        ctx.check(ctx.lib.gmmvi_comm_init(ctx.handle, uid, n_ranks, rank))
        ctx.n_ranks, ctx.rank = n_ranks, rank
        self._scalar = ctx.empty((1,))
        self.barrier()
        if rank == 0:
            try:
                os.remove(path)
            except OSError:
                pass

    def allgather(self, arr):
        """[m, ...] per rank -> [R*m, ...] in rank order."""
        out = self.ctx.empty((self.n_ranks * arr.shape[0],) + tuple(arr.shape[1:]), arr.dtype)
        self.ctx.check(self.ctx.lib.gmmvi_allgather_f32(self.ctx.handle, arr.ptr, out.ptr, arr.size))
        return out

    def allgather_inplace(self, buf, count):
        """``buf`` holds n_ranks parts of ``count`` floats; this rank's part is filled: gather the others in place."""
        self.ctx.check(self.ctx.lib.gmmvi_allgather_f32(self.ctx.handle, buf.ptr + 4 * count * self.rank, buf.ptr, count))

    def barrier(self):
        self._scalar.set(np.zeros(1, np.float32))
        self.ctx.check(self.ctx.lib.gmmvi_allreduce_f32(self.ctx.handle, self._scalar.ptr, 1, 0))
        self.ctx.sync()

    def max_scalar(self, v):
        self._scalar.set(np.array([v], np.float32))
        self.ctx.check(self.ctx.lib.gmmvi_allreduce_f32(self.ctx.handle, self._scalar.ptr, 1, 1))
        return float(self._scalar.numpy()[0])


class LocalExchange:
    """Single-rank stand-in (n_ranks == 1): the sharded code path without a communicator."""
    n_ranks, rank = 1, 0

    def allgather(self, arr):
        return arr

    def allgather_inplace(self, buf, count):
        pass

    def barrier(self):
        pass

    def max_scalar(self, v):
        return v


# ---------------------------------------------------------------------------------------------------------------------
# orchestration (back-end agnostic)
# ---------------------------------------------------------------------------------------------------------------------
class ShardedGMMVI:
    def __init__(self, ops, exchange, d, k_total, means_loc, chols_loc, samples_per_component, seed, cfg,
                 history_length=64):
        self.ops, self.exchange = ops, exchange
        self.R, self.rank = exchange.n_ranks, exchange.rank
        self._check_scope(cfg)
        if k_total % self.R:
            raise ValueError("the number of components must be divisible by the number of ranks")
        self.d, self.K, self.Kl = d, k_total, k_total // self.R
        self.S = int(samples_per_component)
        self.Nl, self.N = self.Kl * self.S, k_total * self.S
        self.lo, self.hi = self.rank * self.Kl, (self.rank + 1) * self.Kl
        self.seed = seed
        self.temperature = float(cfg["temperature"])
        self.cs = cfg["component_stepsize_adapter_config"]
        self.ws = cfg["weight_stepsize_adapter_config"]
        o = ops
        self.means, self.chols = o.asarray(means_loc), o.asarray(chols_loc)
        self._logw = o.asarray(np.full(k_total, -np.log(k_total), np.float32))            # replicated
        self._pending = None           # (expected log-ratios, rewards) of the local components, weight step not applied yet
        self.stepsizes = o.full((self.Kl,), float(self.cs["initial_stepsize"]))
        self.last_eta = o.full((self.Kl,), -1.0)
        self.l2 = o.full((self.Kl,), 1e-12)
        self.num_received_updates = o.full((self.Kl,), 0.0)
        self.H = int(history_length)
        self.reward_ring = o.full((self.H, k_total), FLOAT32_MIN)                         # replicated
        self.t_reward = 0
        self.wstate = o.asarray(np.array([self.ws["initial_stepsize"], FLOAT32_MIN], np.float32))
        self.logc_loc = o.asarray(np.full(self.Kl, np.log(self.S / self.N), np.float32))   # background log-counts
        self.counts_loc = np.full(self.Kl, self.S, np.int64)
        self.num_samples_written = 0
        self.num_updates = 0
        self.last_success = None
        self.packed = None                         # packed blocks of the local components (kept up to date by update_kl)
        self._fast = None
        if isinstance(ops, HipOps) and os.environ.get("GMMVI_FAST_PATH", "1") != "0":
            self._setup_phased(cfg)

    @staticmethod
    def _check_scope(cfg):
        """The sharded iteration covers the configuration bench.py scales; anything else is refused up front instead of
        silently running something different.  An adaptive number of components has its own class,
        ``gmmvi_amd.sharded_adaptive.ShardedAdaptiveGMMVI`` (partition by component id, replicated add / delete decisions)."""
        def refuse(what):
            raise NotImplementedError(f"ShardedGMMVI: {what} is not supported on the component-sharded path "
                                      "(supported: Stein estimator, KL trust-region updates, fixed number of components, "
                                      "reuse ratio 0, full covariances); run it on one GPU with gmmvi_amd.optimization.gmmvi.GMMVI")
        if cfg.get("num_component_adapter_type", "fixed") != "fixed":
            refuse("an adaptive number of components (num_component_adapter_type = "
                   f"{cfg['num_component_adapter_type']!r}: use gmmvi_amd.sharded_adaptive.ShardedAdaptiveGMMVI)")
        if float(cfg.get("sample_selector_config", {}).get("ratio_reused_samples_to_desired", 0.0)) != 0.0:
            refuse("sample reuse (ratio_reused_samples_to_desired > 0)")
        if cfg.get("ng_estimator_type", "Stein") != "Stein":
            refuse(f"the {cfg['ng_estimator_type']} estimator")
        if cfg.get("ng_based_updater_type", "trust-region") != "trust-region":
            refuse(f"the {cfg['ng_based_updater_type']} component updater")
        if cfg.get("model_initialization", {}).get("use_diagonal_covs", False):
            refuse("a diagonal-covariance model")

    # reward ring helpers (same convention as GmmWrapper)
    def _slot(self, back):
        s = (self.t_reward - 1 - back) % self.H
        return self.ops.rows(self.reward_ring, s, s + 1)

    @property
    def log_weights(self):
        """Replicated mixture log-weights [K].  The weight step of the last iteration travels with the next exchange:
        ``flush()`` (a collective: every rank must call it) applies it."""
        if self._pending is not None:
            raise RuntimeError("ShardedGMMVI.log_weights: the last weight step is still pending; call flush() on "
                               "every rank first")
        return self._logw

    def _apply_weight_step(self, e, reward):
        """Replicated on every rank from the gathered [K] expected log-ratios / rewards (weight_updater.py:56-100,
        gmm_wrapper.py:150-160): new reward column, then the weight trust-region step."""
        o = self.ops
        s = self.t_reward % self.H
        o.copy_into(self._row1d(o.rows(self.reward_ring, s, s + 1)), reward)
        self.t_reward += 1
        o.update_weights(self._logw, e, o.rows(self.wstate, 0, 1), self.temperature)
        self._pending = None

    def flush(self):
        """Apply a pending weight step (one small all-gather of [2 Kl] floats per rank).  Collective."""
        if self._pending is None:
            return
        o, ex = self.ops, self.exchange
        e_loc, reward_loc = self._pending
        if self.R > 1:
            e, reward = o.unpack(ex.allgather(o.concat([e_loc, reward_loc])), self.R, [self.Kl, self.Kl])
        else:
            e, reward = e_loc, reward_loc
        self._apply_weight_step(e, reward)

    # ---- the iteration as four C calls with an all-gather between them (gmmvi_train_iter_sharded_phase) --------------------------
    def _setup_phased(self, cfg):
        """Buffers and the plan of the phased C iteration: the SAME launches as the single-GPU single-call iteration (packed
        sweeps with carried merges, Stein slab whitened inside the update kernel, the next draw as riders), the exchanges issued
        from here.  Built-in targets and register-path dimensions only; anything else keeps the module-by-module path below."""
        from . import _lib, hip_ops
        ctx, tgt = self.ops.ctx, self.ops.target
        if not hasattr(tgt, "_fast_path_target") or self.d > _lib.blocked_above() or self.d >= _lib.MAX_DIM:
            return
        if not hasattr(self.exchange, "allgather_inplace"):
            return
        snis = bool(cfg["ng_estimator_config"].get("use_self_normalized_importance_weights", True))
        if bool(cfg["weight_updater_config"].get("use_self_normalized_importance_weights", True)) != snis:
            return
        if cfg.get("weight_updater_type", "trust-region") != "trust-region":
            return
        R, K, D, N, Nl, Kt = self.R, self.Kl, self.d, self.N, self.Nl, self.K
        s1, s2 = Nl * (2 * D + 1) + 2 * K, N * (D + 2)
        f = type("PhasedState", (), {})()
        f.s1, f.s2 = s1, s2
        f.e1, f.e2, f.e3 = ctx.zeros((R * s1,)), ctx.empty((R * s2,)), ctx.empty((R * N,))
        part = f.e1.rows(self.rank * s1, (self.rank + 1) * s1)
        f.E_loc, f.reward_loc = part.rows(Nl * (2 * D + 1), Nl * (2 * D + 1) + K), part.rows(Nl * (2 * D + 1) + K, s1)
        if R == 1:                                     # the parts ARE the gathered arrays
            f.x_all, f.tlp_all = part.rows(0, N * D), part.rows(N * D, N * D + N)
            f.tgrad_all = part.rows(N * D + N, N * (2 * D + 1))
            f.E_all, f.reward_all = f.E_loc, f.reward_loc
        else:
            f.x_all, f.tlp_all, f.tgrad_all = ctx.empty((N * D,)), ctx.empty((N,)), ctx.empty((N * D,))
            f.E_all, f.reward_all = ctx.empty((Kt,)), ctx.empty((Kt,))
        f.scratch = ctx.empty((int(ctx.lib.gmmvi_sharded_scratch_floats(K, D, N)),))
        stride = int(hip_ops.packed_stride(D))
        f.packed_next = ctx.empty((K, stride))
        f.success = ctx.empty((K,), np.int32)
        offsets = np.concatenate([[0], np.cumsum(self.counts_loc)]).astype(np.int32)
        f.offsets = ctx.asarray(offsets, np.int32)
        t = tgt._fast_path_target()
        f.target_keepalive = t
        p = ShardedPlan()
        p.n_ranks, p.rank, p.K, p.D, p.N = R, self.rank, K, D, N
        p.target_kind = t["kind"]
        p.target_family, p.target_K, p.target_nu = t.get("family", 0), t.get("K", 0), t.get("nu", 0.0)
        p.target_packed, p.target_logw = t.get("packed"), t.get("logw")
        p.planar_prior_std, p.planar_goals = t.get("prior_std"), t.get("goals")
        p.planar_goals_count, p.planar_likelihood_std = t.get("G", 0), t.get("lik_std", 0.0)
        p.e1, p.e2, p.e3 = f.e1.ptr, f.e2.ptr, f.e3.ptr
        p.x_all, p.tlp_all, p.tgrad_all = f.x_all.ptr, f.tlp_all.ptr, f.tgrad_all.ptr
        p.E_all, p.reward_all = f.E_all.ptr, f.reward_all.ptr
        p.bg_logw, p.offsets, p.max_per_component = self.logc_loc.ptr, f.offsets.ptr, int(self.counts_loc.max())
        p.seed = int(self.seed) & 0xFFFFFFFFFFFFFFFF
        p.wstate = self.wstate.ptr
        p.temperature, p.l2_init = self.temperature, 1e-12
        p.component_stepsize_mode = 1
        p.cs_min, p.cs_max = self.cs["min_stepsize"], self.cs["max_stepsize"]
        p.cs_inc, p.cs_dec = self.cs["stepsize_inc_factor"], self.cs["stepsize_dec_factor"]
        p.weight_stepsize_mode = 1
        p.ws_min, p.ws_max = self.ws["min_stepsize"], self.ws["max_stepsize"]
        p.ws_inc, p.ws_dec = self.ws["stepsize_inc_factor"], self.ws["stepsize_dec_factor"]
        p.stein_flags = _lib.SELF_NORMALIZED if snis else 0
        p.scratch = f.scratch.ptr
        f.plan, f.presampled = p, False
        f.presample = os.environ.get("GMMVI_PRESAMPLE", "1") != "0"
        self._fast = f

    def _train_iter_phased(self):
        f, ex, ctx = self._fast, self.exchange, self.ops.ctx
        p = f.plan
        K, Kt, H = self.Kl, self.K, self.H
        if self.packed is None:
            self.packed = self.ops.pack(self.means, self.chols)
        p.means, p.chols, p.logw_all = self.means.ptr, self.chols.ptr, self._logw.ptr
        p.packed, p.packed_new = self.packed.ptr, f.packed_next.ptr
        p.stepsizes, p.last_eta, p.l2 = self.stepsizes.ptr, self.last_eta.ptr, self.l2.ptr
        p.num_updates, p.success_out = self.num_received_updates.ptr, f.success.ptr
        p.first_index = self.num_samples_written + self.rank * self.Nl
        pending = self._pending is not None
        p.has_pending = int(pending)
        ring = self.reward_ring.ptr
        if pending:                                    # the gathered rewards become the next column of the (replicated) history
            p.reward_col_pending = ring + 4 * Kt * (self.t_reward % H)
            self.t_reward += 1
        last, prev = (self.t_reward - 1) % H, (self.t_reward - 2) % H
        p.reward_last_all = ring + 4 * Kt * last
        p.reward_last = ring + 4 * (Kt * last + self.lo)
        p.reward_prev = ring + 4 * (Kt * prev + self.lo)
        p.presampled, p.presample_next = int(f.presampled), int(f.presample)
        call = lambda phase: ctx.check(ctx.lib.gmmvi_train_iter_sharded_phase(ctx.handle, _C.byref(p), phase))
        call(1)
        ex.allgather_inplace(f.e1, f.s1)
        call(2)
        ex.allgather_inplace(f.e2, f.s2)
        call(3)
        ex.allgather_inplace(f.e3, self.N)
        call(4)
        self.packed, f.packed_next = f.packed_next, self.packed
        self.last_success = f.success
        self._pending = (f.E_loc, f.reward_loc)        # travel with the next iteration's first exchange (or flush())
        f.presampled = f.presample
        self.num_samples_written += self.N
        self.num_updates += 1

    def train_iter(self):
        if self._fast is not None:
            return self._train_iter_phased()
        o, ex, d, R, N, Nl = self.ops, self.exchange, self.d, self.R, self.N, self.Nl
        # ---- sampling + target (local components); exchange E1: [x | log p~ | grad log p~ (| pending E, reward)] in ONE
        # all-gather; neither sampling nor the target reads the mixture weights, so the previous weight step may still be open
        first = self.num_samples_written + self.rank * self.Nl
        x_loc = o.sample(self.means, self.chols, self.counts_loc, self.seed, first)
        tlp_loc, tgrad_loc = o.target_eval(x_loc)
        parts, sizes = [x_loc, tlp_loc, tgrad_loc], [Nl * d, Nl, Nl * d]
        if self._pending is not None:
            parts += list(self._pending)
            sizes += [self.Kl, self.Kl]
        if R > 1:
            outs = o.unpack(ex.allgather(o.concat(parts)), R, sizes)
            x, tlp, tgrad = outs[0].reshape((N, d)), outs[1], outs[2].reshape((N, d))
        else:
            outs = parts
            x, tlp, tgrad = x_loc, tlp_loc, tgrad_loc
        if self._pending is not None:
            self._apply_weight_step(outs[3], outs[4])
        self.num_samples_written += N
        # ---- partial background / model densities over the local components (one sweep); exchange E2 ------------------------
        if self.packed is None:
            self.packed = o.pack(self.means, self.chols)
        packed = self.packed
        logw_loc = o.rows(self._logw, self.lo, self.hi)
        ld, lq_part, qg_part, bg_part = o.mixture_dual(packed, logw_loc, self.logc_loc, x, d)
        if R > 1:
            g2 = ex.allgather(o.concat([bg_part, lq_part, qg_part]))
            bgp, lqp, qgp = o.unpack(g2, R, [N, N, N * d])
            bg, _ = o.combine(bgp.reshape((R, N)), None, d)
            logq, qgrad = o.combine(lqp.reshape((R, N)), qgp.reshape((R, N, d)), d)
        else:
            bg, logq, qgrad = bg_part, lq_part, qg_part
        # ---- component updates (local) ----------------------------------------------------------------------------------
        prev, last = (o.rows(self._row1d(self._slot(1)), self.lo, self.hi),
                      o.rows(self._row1d(self._slot(0)), self.lo, self.hi))
        o.component_stepsize(self.stepsizes, prev, last, self.cs)
        h_neg, g_neg = o.stein(packed, x, ld, qgrad, bg, tgrad, d)
        self.last_success, self.packed = o.update_kl(self.means, self.chols, h_neg, g_neg, self.stepsizes,
                                                     self.temperature, 1e-12, self.last_eta, self.l2,
                                                     self.num_received_updates)
        # ---- weight update: post-update density (E3); the expected log-ratios + rewards wait for the next E1 -------------------
        o.weight_stepsize(self._logw, self._row1d(self._slot(0)), self.wstate, self.ws)
        ld2, lq2_part, _ = o.mixture(self.packed, logw_loc, x, d, want_ld=True)
        logq2 = o.combine(self._stack(ex.allgather(lq2_part), N), None, d)[0] if R > 1 else lq2_part
        self._pending = o.elr(ld2, bg, tlp, logq2, self.temperature, logw_loc)
        self.num_updates += 1

    @staticmethod
    def _row1d(a):
        return a.reshape(-1)

    @staticmethod
    def _stack(a, n, d=None):
        """[R*n(,d)] gathered buffer -> [R, n(,d)] view."""
        r = a.shape[0] // n
        return a.reshape((r, n) if d is None else (r, n, d))

    # ---- bench.py factory ------------------------------------------------------------------------------------------------
    @staticmethod
    def build(w, n_ranks, rank):
        """From bench.py's workload dict: shard the K initial components over the ranks and connect RCCL."""
        from .device import get_context
        from . import hip_ops
        ctx = get_context()
        k_total = w["k_total"]
        kl = k_total // n_ranks
        lo, hi = rank * kl, (rank + 1) * kl
        chols, ok = hip_ops.cholesky(ctx, ctx.asarray(w["covs"][lo:hi]))
        exchange = RcclExchange(ctx, n_ranks, rank) if n_ranks > 1 else LocalExchange()
        return ShardedGMMVI(HipOps(ctx, w["target"]), exchange, w["d"], k_total, w["means"][lo:hi], chols, w["s"],
                            w["seed"], w["cfg"])

    # ---- host views (tests / metrics) -----------------------------------------------------------------------------------
    def gather_model(self):
        """(log_weights [K], means [K,D], chols [K,D,D]) on the host, gathered from all ranks."""
        o, ex = self.ops, self.exchange
        self.flush()
        return (o.to_host(self._logw), o.to_host(ex.allgather(self.means)), o.to_host(ex.allgather(self.chols)))
