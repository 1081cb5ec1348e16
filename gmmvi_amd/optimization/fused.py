"""Single-call fast path of GMMVI.train_iter() (reference: src/gmmvi/optimization/gmmvi.py:146-174).

When every plug-in module is one of the built-in SAMTRON-style choices (component-based selector, Stein estimator,
KL-constrained component updater, improvement-based or fixed stepsizes, trust-region or direct weight updater, built-in
target), the iteration is issued by ONE C call (``gmmvi_train_iter_samtron``) that composes the same entry points the
modules call, on the same state arrays.  Python only does the bookkeeping the reference keeps in ``tf.Variable``s (DB
length, ring positions, counters).
Sample reuse (``ratio_reused_samples_to_desired`` > 0, the reference's default ``component-based.yml:3-4``): the number
of new samples per component follows from the effective sample sizes of the reused ones (sample_selector.py:160-202);
those K numbers are computed on the device by the same three launches the selector module issues and read back (the one
host synchronisation of such an iteration, as upstream's ``tf.floor`` implies), then the rest of the iteration is the one
C call.  Anything else -- MORE, direct/iBLR updaters, own-samples-only, user targets, diagonal GMMs, want_info, an
iteration in which the database has to be thinned out -- takes the modular path.  Disable with ``GMMVI_FAST_PATH=0``.
"""
import ctypes as C
import os

import numpy as np

from .. import _lib, hip_ops
from .gmmvi_modules.sample_selector import VipsSampleSelector
from .gmmvi_modules.ng_estimator import SteinNgEstimator
from .gmmvi_modules.ng_based_component_updater import KLConstrainedNgBasedComponentUpdater
from .gmmvi_modules.component_stepsize_adaptation import (FixedComponentStepsizeAdaptation,
                                                          ImprovementBasedComponentStepsizeAdaptation)
from .gmmvi_modules.weight_stepsize_adaptation import (FixedWeightStepsizeAdaptation,
                                                       ImprovementBasedWeightStepsizeAdaptation)
from .gmmvi_modules.weight_updater import DirectWeightUpdater, TrustRegionBasedWeightUpdater
from .gmmvi_modules.component_adaptation import FixedComponentAdaptation

_f, _i, _p = C.c_float, C.c_int32, C.c_void_p


class SamtronPlan(C.Structure):
    """struct gmmvi_samtron_plan (include/gmmvi_hip.h)."""
    _fields_ = [
        ("K", _i), ("D", _i), ("N", _i), ("target_kind", _i), ("target_family", _i), ("target_K", _i),
        ("target_nu", _f), ("target_packed", _p), ("target_logw", _p), ("planar_prior_std", _p), ("planar_goals", _p),
        ("planar_goals_count", _i), ("planar_likelihood_std", _f),
        ("means", _p), ("chols", _p), ("logw", _p), ("packed", _p), ("packed_new", _p),
        ("stepsizes", _p), ("last_eta", _p), ("l2", _p), ("num_updates", _p), ("success_out", _p),
        ("offsets", _p), ("max_per_component", _i), ("n_old", _i), ("bg_K", _i), ("bg_packed", _p), ("bg_logw", _p),
        ("bg_old", _p), ("bg_logw_new", _p), ("bg_log_share_old", _f), ("bg_log_share_new", _f),
        ("seed", C.c_uint64), ("first_index", C.c_uint64),
        ("db_samples", _p), ("db_tlp", _p), ("db_tgrad", _p), ("db_mapping", _p), ("mapping_base", _i),
        ("db_means", _p), ("db_chols", _p), ("db_packed", _p),
        ("reward_prev", _p), ("reward_last", _p), ("reward_next", _p), ("weight_slot", _p), ("wstate", _p),
        ("temperature", _f), ("l2_init", _f),
        ("component_stepsize_mode", _i), ("cs_min", _f), ("cs_max", _f), ("cs_inc", _f), ("cs_dec", _f),
        ("weight_stepsize_mode", _i), ("ws_min", _f), ("ws_max", _f), ("ws_inc", _f), ("ws_dec", _f),
        ("weight_update_mode", _i), ("stein_flags", _i), ("presample_next", _i), ("presampled", _i), ("phase", _i),
    ]


class SamtronFastPath:
    def __init__(self, gmmvi):
        self.g = gmmvi
        self.plan = SamtronPlan()
        self.enabled = os.environ.get("GMMVI_FAST_PATH", "1") != "0"
        self._static_ok = None
        # True: the Stein estimate is materialised exactly as the modules do it (bit-equal trajectories); False: the update
        # kernel whitens the moment sums directly where it can (csrc/update_kl.hip), same mathematics with fewer roundings
        self.explicit_estimate = os.environ.get("GMMVI_EXPLICIT_ESTIMATE", "0") == "1"
        # the NEXT iteration's draw rides in this iteration's post-update sweep (csrc/riders.h) when nothing can change the
        # components, the counts or the database in between: fixed number of components, reuse ratio 0
        self.presample = os.environ.get("GMMVI_PRESAMPLE", "1") != "0"
        self._presample_token = None
        # sample reuse: the effective sample sizes of the NEXT iteration's window need the updated components but not the weights;
        # they are queued between the component update and the weight update (the iteration as two C calls) and read back
        # asynchronously, so the next iteration does not start with a wait for the device
        self.prefetch_counts = os.environ.get("GMMVI_PREFETCH_COUNTS", "1") != "0"
        self._prefetch = None
        self._pinned = None              # (pinned host pointer, floats, NumPy view)
        self._event = None
        lib = _lib.load()
        lib.gmmvi_train_iter_samtron.restype = C.c_int
        lib.gmmvi_train_iter_samtron.argtypes = [C.c_void_p, C.POINTER(SamtronPlan)]
        self._fn = lib.gmmvi_train_iter_samtron

    # ---- eligibility -----------------------------------------------------------------------------------------------------
    def _check_static(self):
        g = self.g
        sel, est, upd = g.sample_selector, g.ng_estimator, g.ng_based_updater
        wu, cs, ws = g.weight_updater, g.component_stepsize_adapter, g.weight_stepsize_adapter
        tgt = sel.target_distribution
        ok = (type(sel) is VipsSampleSelector and type(est) is SteinNgEstimator
              and type(upd) is KLConstrainedNgBasedComponentUpdater
              and type(cs) in (FixedComponentStepsizeAdaptation, ImprovementBasedComponentStepsizeAdaptation)
              and type(ws) in (FixedWeightStepsizeAdaptation, ImprovementBasedWeightStepsizeAdaptation)
              and type(wu) in (DirectWeightUpdater, TrustRegionBasedWeightUpdater)
              and hasattr(tgt, "_fast_path_target")
              and not est._only_use_own_samples
              and est._use_self_normalized_importance_weights == wu.use_self_normalized_importance_weights
              and g.sample_db.keep_samples and g.model.num_dimensions < _lib.MAX_DIM
              and g.model.num_dimensions <= _lib.blocked_above()
              and not g.model.diagonal_covs)
        return bool(ok)

    def eligible(self):
        ok = self._eligible()
        if not ok and self._prefetch is not None:
            self.drop_prefetch()
        return ok

    def _eligible(self):
        if not self.enabled:
            return False
        if self._static_ok is None:
            self._static_ok = self._check_static()
        if not self._static_ok:
            return False
        g = self.g
        sel, db = g.sample_selector, g.sample_db
        if sel.eps_override is not None or not sel.fuse_background:
            return False
        if g.ng_based_updater.want_info or g.weight_updater.want_info:
            return False
        # worst case of this iteration's append: every component draws its full share
        n_new = sel.desired_samples_per_component * g.model.num_components
        if db.max_samples is not None and n_new + db._samples.n > db.max_samples:
            return False                                   # the modular path thins the DB out first (sample_db.py:111-112)
        return sel.desired_samples_per_component >= 1

    # ---- effective sample sizes of a reuse window, fetched ahead -----------------------------------------------------------
    def _state_token(self):
        """What the prefetched effective sample sizes depend on: the database as it is, the components, the selector."""
        g = self.g
        sel, db, model = g.sample_selector, g.sample_db, g.model.model
        return (db._samples.n, db._means.n, db._epoch, int(db._num_samples_written), db._samples.buf.ptr, model.num_components,
                id(model._packed), model.means.ptr, model.chol_cov.ptr, sel.desired_samples_per_component,
                sel.reused_samples_per_component)

    def drop_prefetch(self):
        """The next iteration will not use what was fetched ahead (another path runs it, or something changed): the database's
        window caches go back to where they were."""
        pf, self._prefetch = self._prefetch, None
        if pf is not None:
            db = self.g.sample_db
            db._bg_cache, db._pd = pf["db_caches"]

    def _pinned_floats(self, n):
        ctx = self.g.model.model.ctx
        if self._pinned is None or self._pinned[1] < n:
            if self._pinned is not None:
                ctx.sync()
                ctx.check(ctx.lib.gmmvi_host_free(ctx.handle, self._pinned[0]))
            cap = max(256, 2 * n)
            ptr = C.c_void_p()
            ctx.check(ctx.lib.gmmvi_host_alloc(ctx.handle, 4 * cap, C.byref(ptr)))
            self._pinned = (ptr.value, cap, np.ctypeslib.as_array((C.c_float * cap).from_address(ptr.value)))
        if self._event is None:
            self._event = ctx.event()
        return self._pinned

    def _issue_counts(self):
        """The selector's launches for the effective sample sizes of the newest reuse window (sample_selector.py:140-202):
        background density of the window, component log densities of the current model on it, effective sample sizes.
        -> (ess [K] on the device, window background density, its mixture, window length)."""
        g = self.g
        sel, db, model = g.sample_selector, g.sample_db, g.model.model
        n_reuse = sel.reused_samples_per_component * model.num_components
        bg_old, xs_old, _, _, _ = db.get_newest_samples(n_reuse)
        ld_old = model.component_log_densities(xs_old)
        return sel.get_effective_samples(ld_old, bg_old), bg_old, db._bg_cache["mix"], int(xs_old.shape[0])

    def _new_sample_counts(self):
        """Per-component numbers of new samples and the window of reused ones (sample_selector.py:160-219).
        -> (counts [K] int64, n_old).  Reuse ratio 0: every component draws its full share, nothing is read back."""
        g = self.g
        sel, db, model = g.sample_selector, g.sample_db, g.model.model
        k, s = model.num_components, sel.desired_samples_per_component
        n_reuse = sel.reused_samples_per_component * k
        if n_reuse == 0 or db._samples.n == 0:
            return np.full(k, s, np.int64), 0
        pf = self._prefetch
        if pf is not None:
            if pf["token"] == self._state_token():
                # fetched ahead by the previous iteration, behind its component update: normally long since there
                self._prefetch = None
                ctx = model.ctx
                # the window's mixture (its snapshot blocks, its log weights) does not depend on the counts either: its device
                # work is queued before the wait
                mix = pf["bg_mix"]() if callable(pf["bg_mix"]) else pf["bg_mix"]
                mix.packed, mix.logw_dev
                pf["bg_mix"] = mix
                ctx.check(ctx.lib.gmmvi_event_synchronize(ctx.handle, self._event))
                n_eff = np.floor(self._pinned[2][:k]).astype(np.int64)
                self._bg_old, self._bg_mix = pf["bg_old"], pf["bg_mix"]
                return np.maximum(1, s - n_eff), pf["n_old"]
            self.drop_prefetch()
        # exactly the selector's launches: background density of the newest n_reuse samples, component log-densities of the
        # current model on them, effective sample sizes; the [K] result is read back (host synchronisation)
        ess, bg_old, bg_mix, n_old = self._issue_counts()
        n_eff = np.floor(ess.numpy()).astype(np.int64)
        self._bg_old = bg_old            # the window's background density so far: the call extends it instead of redoing it
        self._bg_mix = bg_mix            # ... and the mixture it belongs to
        return np.maximum(1, s - n_eff), n_old

    def _fill_static(self, p, packed_cur, packed_new, success):
        """The plan fields that do not depend on the iteration's sample counts."""
        g = self.g
        m = g.model
        model = m.model
        sel = g.sample_selector
        tgt = sel.target_distribution._fast_path_target()
        p.target_kind = tgt["kind"]
        p.target_family, p.target_K, p.target_nu = tgt.get("family", 0), tgt.get("K", 0), tgt.get("nu", 0.0)
        p.target_packed, p.target_logw = tgt.get("packed"), tgt.get("logw")
        p.planar_prior_std, p.planar_goals = tgt.get("prior_std"), tgt.get("goals")
        p.planar_goals_count, p.planar_likelihood_std = tgt.get("G", 0), tgt.get("lik_std", 0.0)
        p.means, p.chols, p.logw = model.means.ptr, model.chol_cov.ptr, model.log_weights.ptr
        p.packed, p.packed_new = packed_cur.ptr, packed_new.ptr
        p.stepsizes, p.last_eta, p.l2 = m.stepsizes.ptr, m.last_log_etas.ptr, m.l2_regularizers.ptr
        p.num_updates, p.success_out = m.num_received_updates.ptr, success.ptr
        p.seed = int(model.seed) & 0xFFFFFFFFFFFFFFFF
        p.reward_prev, p.reward_last = m.reward_slot(1).ptr, m.reward_slot(0).ptr
        p.reward_next = m.next_reward_slot().ptr
        p.weight_slot = m.next_weight_slot().ptr
        ws, cs, wu = g.weight_stepsize_adapter, g.component_stepsize_adapter, g.weight_updater
        p.wstate = ws._state.ptr
        p.temperature, p.l2_init = float(g.temperature), float(m.initial_regularizer)
        if type(cs) is ImprovementBasedComponentStepsizeAdaptation:
            p.component_stepsize_mode = 1
            p.cs_min, p.cs_max = cs.min_stepsize, cs.max_stepsize
            p.cs_inc, p.cs_dec = cs.stepsize_inc_factor, cs.stepsize_dec_factor
        else:
            p.component_stepsize_mode = 0
        if type(ws) is ImprovementBasedWeightStepsizeAdaptation:
            p.weight_stepsize_mode = 1
            p.ws_min, p.ws_max = ws.min_stepsize, ws.max_stepsize
            p.ws_inc, p.ws_dec = ws.stepsize_inc_factor, ws.stepsize_dec_factor
        else:
            p.weight_stepsize_mode = 0
        p.weight_update_mode = 0 if type(wu) is TrustRegionBasedWeightUpdater else 1
        p.stein_flags = _lib.SELF_NORMALIZED if g.ng_estimator._use_self_normalized_importance_weights else 0
        if self.explicit_estimate:
            p.stein_flags |= _lib.EXPLICIT_ESTIMATE

    # ---- one iteration -----------------------------------------------------------------------------------------------------
    def step(self):
        g = self.g
        m = g.model                      # GmmWrapper
        model = m.model
        ctx = model.ctx
        sel, db = g.sample_selector, g.sample_db
        k, d = model.num_components, model.num_dimensions
        p = self.plan

        # everything that does not depend on this iteration's sample counts comes first: with counts fetched ahead the device
        # may still be on its way to them, and this is host time it hides
        stride = db._packed.inner[0]
        packed_cur = model.packed
        packed_new = ctx.empty((k, stride))
        success = ctx.empty((k,), np.int32)
        self._fill_static(p, packed_cur, packed_new, success)

        counts, n_old = self._new_sample_counts()
        n = int(counts.sum())
        key = counts.tobytes()
        offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        offsets_dev = ctx.cached_const(("offsets", offsets.tobytes()), lambda: ctx.asarray(offsets, np.int32))
        if getattr(sel, "_mapping_key", None) != key:
            sel._mapping_key = key
            sel._last_mapping_host = np.repeat(np.arange(k, dtype=np.int32), counts)
        sel._last_counts = counts

        # may this call draw the next iteration's samples as well?  (sample_selector.py:204-219: the draw needs the updated
        # components only)  Only when the next iteration is bound to be this same call on the same shapes.
        presample_next = (self.presample and n_old == 0 and sel.reused_samples_per_component == 0
                          and type(g.num_component_adapter) is FixedComponentAdaptation
                          and (db.max_samples is None or db._samples.n + 2 * n <= db.max_samples))
        # SampleDB: reserve room (for the early draw too), remember the append position (sample_db.py:113-124)
        for grow, rows in ((db._samples, 2 * n if presample_next else n), (db._target_lnpdfs, n), (db._target_grads, n),
                           (db._mapping_dev, 2 * n if presample_next else n), (db._means, k), (db._chols, k), (db._packed, k)):
            grow.reserve(rows)
        s0, c0 = db._samples.n, db._means.n

        # background mixture of the active window (sample_db.py:216-227): with nothing reused its components are the
        # model's own and share the model's sweep; otherwise the snapshots of the window's sampling components (the new
        # ones, appended by this call at row c0, included), weighted by their sample counts
        # (the host mirror of the mapping and the append log get this iteration's entry first: the window below includes it)
        db._mapping_host.append(sel._last_mapping_host + np.int32(c0))
        db._segments.append((s0, c0, counts))
        if n_old == 0:
            bg_logw = ctx.cached_const(("bg_logw", key), lambda: ctx.asarray(np.log(counts / counts.sum()).astype(np.float32)))
            p.n_old, p.bg_K, p.bg_packed = 0, 0, None
        else:
            # the components the reused samples came from, weighted by their counts among the reused samples: exactly the mixture
            # get_newest_samples(n_old) evaluated for bg_old (same window; this iteration's entry not counted)
            mix = self._bg_mix() if callable(self._bg_mix) else self._bg_mix
            assert int(mix.active.max()) < c0, "the reused samples come from earlier appends"
            self._bg_keepalive = (mix.packed, mix.logw_dev)
            bg_packed_ptr, bg_logw = mix.packed.ptr, mix.logw_dev
            bg_logw_new = ctx.cached_const(("bg_logw", key), lambda: ctx.asarray(np.log(counts / counts.sum()).astype(np.float32)))
            self._bg_logw_new_keepalive = bg_logw_new
            p.n_old, p.bg_K, p.bg_packed = n_old, len(mix.active), bg_packed_ptr
            p.bg_old, p.bg_logw_new = self._bg_old.ptr, bg_logw_new.ptr
            p.bg_log_share_old, p.bg_log_share_new = db.log_shares(n_old, n)
        p.max_per_component = int(counts.max())

        p.K, p.D, p.N = k, d, n
        p.offsets, p.bg_logw = offsets_dev.ptr, bg_logw.ptr
        p.first_index = int(db._num_samples_written)
        p.db_samples = db._samples.buf.ptr + s0 * d * 4
        p.db_tlp = db._target_lnpdfs.buf.ptr + s0 * 4
        p.db_tgrad = db._target_grads.buf.ptr + s0 * d * 4
        p.db_mapping, p.mapping_base = db._mapping_dev.buf.ptr + s0 * 4, c0
        p.db_means = db._means.buf.ptr + c0 * d * 4
        p.db_chols = db._chols.buf.ptr + c0 * d * d * 4
        p.db_packed = db._packed.buf.ptr + c0 * stride * 4

        # this iteration's samples may already be there: drawn by the previous call behind its component update
        here = (db._samples.buf.ptr, db._mapping_dev.buf.ptr, s0, c0, n, k, key, int(db._num_samples_written),
                int(model.seed), model.means.ptr, model.chol_cov.ptr, offsets_dev.ptr)
        tok, self._presample_token = self._presample_token, None
        p.presampled = int(tok is not None and n_old == 0 and tok[0] == here and tok[1] is model._packed)
        p.presample_next = int(presample_next)

        prefetch_next = self.prefetch_counts and sel.reused_samples_per_component > 0
        p.phase = 1 if prefetch_next else 0
        try:
            ctx.check(self._fn(ctx.handle, C.byref(p)))
        except Exception:
            # the host mirror of the mapping and the append log are one entry ahead of the device buffers: take it back
            db._mapping_host.n -= len(sel._last_mapping_host)
            db._segments.pop()
            db._seg_start, db._seg_size, db._seg_info = [], [], {}
            raise

        # bookkeeping the modules would have done
        for grow, rows in ((db._samples, n), (db._target_lnpdfs, n), (db._target_grads, n), (db._mapping_dev, n),
                           (db._means, k), (db._chols, k), (db._packed, k)):
            grow.n += rows
        db._num_samples_written += n
        if n_old > 0:
            db._bg_cache = None          # the window's density was extended inside the call (SampleDB.get_newest_samples does the same)
        model._packed = packed_new
        model._eval_cache = None
        if prefetch_next:
            # between the component update and the weight update: the next iteration's effective sample sizes (they need the
            # appended samples and the updated components, both there; not the weights), read back without waiting
            caches = (db._bg_cache, db._pd)
            ess, bg_old_next, bg_mix_next, n_old_next = self._issue_counts()
            pinned = self._pinned_floats(k)
            ctx.check(ctx.lib.gmmvi_download_async(ctx.handle, pinned[0], ess.ptr, 4 * k))
            ctx.record(self._event)
            p.phase = 2
            ctx.check(self._fn(ctx.handle, C.byref(p)))
            p.phase = 0
            self._prefetch = {"ess": ess, "bg_old": bg_old_next, "bg_mix": bg_mix_next, "n_old": n_old_next,
                              "db_caches": caches, "token": None}
        m.commit_rewards()
        if k > 1:
            m._t_weight += 1
        if presample_next:
            self._presample_token = ((db._samples.buf.ptr, db._mapping_dev.buf.ptr, s0 + n, c0 + k, n, k, key,
                                      int(db._num_samples_written), int(model.seed), model.means.ptr, model.chol_cov.ptr,
                                      offsets_dev.ptr), packed_new)
        g.ng_based_updater.last_success = success
        g.num_updates.assign_add(1)
        if self._prefetch is not None:
            self._prefetch["token"] = self._state_token()
