"""GmmWrapper: per-component learner state beside the model (reference: src/gmmvi/models/gmm_wrapper.py:4-182).

State the kernels read/write every iteration lives in HBM (stepsizes, l2_regularizers, last_log_etas,
num_received_updates, and the reward / weight histories as *ring buffers* [H, Kcap] instead of the reference's
concat-shift of a [K, 10000] tensor twice per iteration, gmm_wrapper.py:158,:182).  Kcap >= K is a capacity: adding a
component fills one column, removing one shifts the columns on the device -- no host round trip of the history.
``reward_history`` / ``weight_history`` materialise the reference's [K, H] layout (newest entry last) on demand.
"""
import numpy as np

from .. import hip_ops
from .gmm import GMM
from .full_cov_gmm import FullCovGMM

FLOAT32_MIN = float(np.finfo(np.float32).min)


class GmmWrapper:
    @staticmethod
    def build_from_config(model: GMM, config: dict):
        """gmm_wrapper.py:33-58."""
        nca = config["num_component_adapter_config"]
        max_reward_history_length = 2 * max(2, nca["del_iters"]) if "del_iters" in nca else 2
        initial_regularizer = config["ng_estimator_config"].get("initial_l2_regularizer", 1e-12)
        initial_stepsize = config["component_stepsize_adapter_config"]["initial_stepsize"]
        return GmmWrapper(model, initial_stepsize, initial_regularizer, max_reward_history_length)

    def __init__(self, model: GMM, initial_stepsize: float, initial_regularizer: float,
                 max_reward_history_length: int):
        self.model = model
        ctx = model.ctx
        k = model.num_components
        self.initial_regularizer = float(initial_regularizer)
        self.initial_last_eta = -1
        self.initial_stepsize = float(initial_stepsize)
        self.max_reward_history_length = int(max_reward_history_length)
        self.l2_regularizers = ctx.full((k,), self.initial_regularizer)          # :68
        self.last_log_etas = ctx.full((k,), float(self.initial_last_eta))        # :69 (stores eta)
        self.num_received_updates = ctx.zeros((k,))                              # :70
        self.stepsizes = ctx.full((k,), self.initial_stepsize)                   # :71
        h = self.max_reward_history_length
        self._kcap = max(64, 2 * k)
        self._reward_ring = ctx.full((h, self._kcap), FLOAT32_MIN)               # :72
        self._weight_ring = ctx.full((h, self._kcap), FLOAT32_MIN)               # :74
        self._t_reward = 0          # number of store_rewards() calls
        self._t_weight = 0          # number of replace_weights() calls
        self.unique_component_ids = np.arange(k, dtype=np.int32)                 # :76
        self.max_component_id = int(self.unique_component_ids.max()) if k else -1
        self.adding_thresholds = -np.ones(k, np.float32)                         # :79
        self.initial_entropies = model.component_entropies()                     # :80

    def __getattr__(self, name):                                                 # :83-88
        return getattr(self.__dict__["model"], name)

    # ---- ring helpers -----------------------------------------------------------------------------------------
    def _slot(self, t):
        return t % self.max_reward_history_length

    def _row(self, ring, slot):
        """Device view [K] of one time slot."""
        return ring.rows(slot, slot + 1).reshape(-1).rows(0, self.model.num_components)

    def reward_slot(self, back):
        """Device view [K] of reward_history[:, -1-back] (back = 0: newest)."""
        return self._row(self._reward_ring, self._slot(self._t_reward - 1 - back))

    def next_reward_slot(self):
        """Device view the next store_rewards() writes to; advance with commit_rewards()."""
        return self._row(self._reward_ring, self._slot(self._t_reward))

    def next_weight_slot(self):
        return self._row(self._weight_ring, self._slot(self._t_weight))

    def commit_rewards(self):
        self._t_reward += 1

    def _materialise(self, ring, t, last_n=None):
        h = self.max_reward_history_length
        k = self.model.num_components
        n = h if last_n is None else min(int(last_n), h)
        start = (t - n) % h                                  # oldest requested slot
        if start + n <= h:
            host = ring.rows(start, start + n).numpy()
        else:                                                # wraps around the ring: two contiguous pieces
            host = np.concatenate([ring.rows(start, h).numpy(), ring.rows(0, start + n - h).numpy()])
        return np.ascontiguousarray(host[:, :k].T)           # [K, n], newest last

    @property
    def reward_history(self):
        return self._materialise(self._reward_ring, self._t_reward)

    @property
    def weight_history(self):
        return self._materialise(self._weight_ring, self._t_weight)

    def reward_window(self, n):
        return self._materialise(self._reward_ring, self._t_reward, n)

    def weight_window(self, n):
        return self._materialise(self._weight_ring, self._t_weight, n)

    # ---- reference API ------------------------------------------------------------------------------------------
    def store_rewards(self, rewards):
        """:150-158."""
        self.next_reward_slot().copy_from(self.model.ctx.asarray(rewards))
        self.commit_rewards()

    def update_stepsizes(self, new_stepsizes):
        """:160-168."""
        new = self.model.ctx.asarray(new_stepsizes)
        if new is not self.stepsizes:
            self.stepsizes = new

    def record_weights(self):
        """weight_history shift of :182 for weights already normalised on the device."""
        hip_ops.exp_into(self.model.ctx, self.next_weight_slot(), self.model.log_weights)
        self._t_weight += 1

    def replace_weights(self, new_log_weights):
        """:170-182."""
        self.model.replace_weights(new_log_weights)
        self.record_weights()

    def _grow_rings(self, k_needed):
        if k_needed <= self._kcap:
            return
        ctx = self.model.ctx
        new_cap = max(2 * self._kcap, k_needed)
        h = self.max_reward_history_length
        for name in ("_reward_ring", "_weight_ring"):
            old = getattr(self, name).numpy()                 # rare (capacity doubling): through the host
            new = np.full((h, new_cap), FLOAT32_MIN, np.float32)
            new[:, :self._kcap] = old
            setattr(self, name, ctx.asarray(new))
        self._kcap = new_cap

    def add_component(self, initial_weight, initial_mean, initial_cov, adding_threshold, initial_entropy):
        """:90-127."""
        ctx = self.model.ctx
        k_old = self.model.num_components
        self._grow_rings(k_old + 1)
        # seven arrays get a row: all copies in two launches (hip_ops.copy_batch takes eight at a time) instead of one per array
        pairs = []
        batched = isinstance(self.model, FullCovGMM)
        if batched:
            self.model.add_component(initial_weight, initial_mean, initial_cov, pairs)
        else:
            self.model.add_component(initial_weight, initial_mean, initial_cov)
        self.max_component_id += 1
        self.unique_component_ids = np.append(self.unique_component_ids, np.int32(self.max_component_id))
        # the four per-component vectors get their new entry on the device (one small upload: nothing read back)
        tail = ctx.asarray(np.array([self.initial_regularizer, self.initial_last_eta, 0.0, self.initial_stepsize], np.float32))
        app = lambda dev, j: self.model._append_rows(dev, tail.rows(j, j + 1), pairs)
        self.l2_regularizers = app(self.l2_regularizers, 0)
        self.last_log_etas = app(self.last_log_etas, 1)
        self.num_received_updates = app(self.num_received_updates, 2)
        self.stepsizes = app(self.stepsizes, 3)
        hip_ops.copy_batch(ctx, pairs)
        if batched:
            self.model.log_weights = self.model._renormalised(self.model.log_weights)
        h = self.max_reward_history_length
        # the new component's history column: rewards float32.min (:121-122), weights initial_weight (:123-124)
        ctx.check(ctx.lib.gmmvi_fill_strided_f32(ctx.handle, self._reward_ring.ptr + 4 * k_old, self._kcap, h,
                                                 FLOAT32_MIN))
        ctx.check(ctx.lib.gmmvi_fill_strided_f32(ctx.handle, self._weight_ring.ptr + 4 * k_old, self._kcap, h,
                                                 float(initial_weight)))
        self.adding_thresholds = np.append(self.adding_thresholds,
                                           np.asarray(adding_threshold, np.float32).reshape(-1))
        self.initial_entropies = np.append(self.initial_entropies,
                                           np.asarray(initial_entropy, np.float32).reshape(-1))

    def remove_component(self, idx):
        """:129-148."""
        ctx = self.model.ctx
        idx = int(idx)
        k_old = self.model.num_components
        self.model.remove_component(idx)
        self.unique_component_ids = np.delete(self.unique_component_ids, idx)
        keep = ctx.asarray(np.delete(np.arange(k_old, dtype=np.int32), idx), np.int32)
        rm = lambda dev: hip_ops.gather_rows(ctx, dev, keep)
        self.l2_regularizers = rm(self.l2_regularizers)
        self.last_log_etas = rm(self.last_log_etas)
        self.num_received_updates = rm(self.num_received_updates)
        self.stepsizes = rm(self.stepsizes)
        h = self.max_reward_history_length
        for ring in (self._reward_ring, self._weight_ring):
            ctx.check(ctx.lib.gmmvi_remove_column_f32(ctx.handle, ring.ptr, h, self._kcap, k_old, idx))
        self.adding_thresholds = np.delete(self.adding_thresholds, idx)
        self.initial_entropies = np.delete(self.initial_entropies, idx)
