"""SampleDB on the MI355X (reference: src/gmmvi/optimization/sample_db.py:30-228).

Samples, target log-densities/gradients and the (mean, chol) snapshot of every sampling component stay in HBM in
capacity-doubling buffers (the reference re-concatenates every tensor on each add, sample_db.py:115-124).  The
int32 ``mapping`` is mirrored on the host: it is produced deterministically from the per-component sample counts,
so ``unique_with_counts`` (sample_db.py:221) costs no device round trip; for the common case that the requested
window ends on an append boundary the answer is read off the append log without touching the mapping at all.
Background densities are evaluated by the same fused density kernel as the model (log-weights = log(count / N),
sample_db.py:221-227).  Steady-state iterations issue no host<->device synchronisation.
"""
import os

import numpy as np

from .. import hip_ops
from ..device import DeviceArray, get_context


def random_subset(rng, total, size):
    """A uniformly random ``size``-subset of range(total), ascending -- the SET that shuffle(range(total))[:size] is
    (sample_db.py:137-152); its order is irrelevant to the consumer, an arg-max over the candidates.  Oversampled independent
    draws, sorted, duplicates dropped, surplus values removed at random: the distinct values of independent uniform draws form a
    uniformly random subset of their number, and removing uniformly chosen members keeps it one.  Vectorised (NumPy's own
    choice(replace=False) walks a hash set element by element: 1.9 ms against 0.85 ms for 1e5 of 3e6 here) and
    ascending, so the gather that follows reads the database front to back."""
    total, size = int(total), min(int(size), int(total))
    if 4 * size > total or total >= 2 ** 31:                 # dense draws (small databases): NumPy's own algorithm
        return np.sort(rng.choice(total, size=size, replace=False, shuffle=False)).astype(np.int32)
    have = None
    while have is None or have.shape[0] < size:
        missing = size - (0 if have is None else have.shape[0])
        draw = rng.integers(0, total, size=missing + max(64, missing // 16), dtype=np.int32)
        have = np.sort(draw if have is None else np.concatenate([have, draw]))
        keep = np.empty(have.shape[0], bool)
        keep[0] = True
        np.not_equal(have[1:], have[:-1], out=keep[1:])
        have = have[keep]
    if have.shape[0] > size:
        have = np.delete(have, rng.choice(have.shape[0], size=have.shape[0] - size, replace=False, shuffle=False))
    return have


class _MappedRange:
    """A reserved device address range with physical memory mapped behind its first ``mapped`` bytes (gmmvi_vmm_*)."""

    def __init__(self, ctx, reserve_bytes):
        import ctypes as C
        base, chunk = C.c_void_p(), C.c_size_t()
        ctx.check(ctx.lib.gmmvi_vmm_reserve(ctx.handle, int(reserve_bytes), C.byref(base), C.byref(chunk)))
        self.ctx, self.base, self.chunk, self.reserved, self.mapped = ctx, base.value, int(chunk.value), int(reserve_bytes), 0

    def ensure(self, nbytes):
        want = (int(nbytes) + self.chunk - 1) // self.chunk * self.chunk
        if want > self.mapped:
            if want > (self.reserved + self.chunk - 1) // self.chunk * self.chunk:
                raise MemoryError(f"sample database: a buffer outgrew its reserved address range of {self.reserved >> 30} GiB "
                                  "(GMMVI_DB_RESERVE_GB)")
            self.ctx.check(self.ctx.lib.gmmvi_vmm_grow(self.ctx.handle, self.base, self.chunk, self.mapped, want))
            self.mapped = want

    def __del__(self):
        try:
            if self.base:
                self.ctx.lib.gmmvi_vmm_release(self.ctx.handle, self.base, self.chunk, self.mapped, self.reserved)
                self.base = None
        except Exception:
            pass


class _Growable:
    """Device buffer with logical length ``n`` along axis 0 and amortised O(1) append.  Small buffers double (allocate, copy,
    release); once a buffer would pass ``MAPPED_FROM`` bytes it moves -- one last copy -- into a reserved address range and
    grows IN PLACE from then on, physical memory mapped chunk by chunk (csrc/api.hip: gmmvi_vmm_*): a doubling of a multi-gigabyte
    buffer costs seconds (hipMalloc + copy + a hipFree that waits for the stream), and the database of a long run is meant to
    fill a good part of the 288 GB."""

    def __init__(self, ctx, inner, dtype=np.float32):
        self.ctx, self.inner, self.dtype = ctx, tuple(inner), dtype
        self.buf = ctx.empty((0,) + self.inner, dtype)
        self.n = 0
        self._range = None

    def view(self, start=0, stop=None):
        stop = self.n if stop is None else stop
        return self.buf.rows(start, stop)

    # the first allocation holds FIRST_APPENDS appends of the first one's size (at most FIRST_BYTES): a run of a few hundred
    # iterations then never re-allocates -- a doubling is a hipMalloc (host-synchronous) plus a copy of everything so far, and a
    # short timed window that happens to contain one reads several percent slower (bench.py: 20 steps against 200)
    FIRST_APPENDS, FIRST_BYTES = 256, 1 << 30
    MAPPED_FROM = 1 << 30                                     # bytes from which a buffer lives in a mapped range
    RESERVE = int(os.environ.get("GMMVI_DB_RESERVE_GB", "64")) << 30      # address range per buffer (costs nothing until mapped)

    def _row_bytes(self):
        return int(np.prod(self.inner, dtype=np.int64)) * np.dtype(self.dtype).itemsize

    def _map_rows(self, rows):
        """Rows [0, rows) of the mapped range usable; ``buf`` re-pointed at everything that is mapped."""
        rb = self._row_bytes()
        self._range.ensure(rows * rb)
        self.buf = DeviceArray._external(self.ctx, self._range.base, (self._range.mapped // rb,) + self.inner, self.dtype, self._range)

    def reserve(self, m):
        if self.n + m <= self.buf.shape[0]:
            return
        if self._range is not None:
            self._map_rows(self.n + m)
            return
        cap = max(2 * self.buf.shape[0], self.n + m, 1024)
        row_bytes = self._row_bytes()
        if self.buf.shape[0] == 0:
            cap = max(cap, min(self.FIRST_APPENDS * m, self.FIRST_BYTES // max(1, row_bytes)))
        if cap * row_bytes >= self.MAPPED_FROM and np.dtype(self.dtype).itemsize == 4:
            old = self.buf
            try:
                self._range = _MappedRange(self.ctx, max(self.RESERVE, 2 * (self.n + m) * row_bytes))
            except Exception as e:                              # no virtual memory management on this stack: keep doubling
                import warnings
                warnings.warn(f"sample database: grow-in-place buffers are not available ({e}); large buffers will double by copy")
                _Growable.MAPPED_FROM = float("inf")
            else:
                self._map_rows(self.n + m)
                if self.n:
                    self.buf.rows(0, self.n).copy_from(old.rows(0, self.n))
                return
        new = self.ctx.empty((cap,) + self.inner, self.dtype)
        if self.buf.shape[0] == 0 and new.size and np.dtype(self.dtype).itemsize == 4:
            # touch the pages now (one fill on the stream) instead of while the appends walk through them
            self.ctx.check(self.ctx.lib.gmmvi_fill_f32(self.ctx.handle, new.ptr, 0.0, new.size))
        if self.n:
            new.rows(0, self.n).copy_from(self.buf.rows(0, self.n))
        self.buf = new

    def append(self, arr):
        m = arr.shape[0]
        self.reserve(m)
        if m:
            self.buf.rows(self.n, self.n + m).copy_from(arr)
        self.n += m

    def append_deferred(self, arr, pairs):
        """Reserve room and queue the copy in ``pairs`` (executed by one hip_ops.copy_batch launch)."""
        m = arr.shape[0]
        self.reserve(m)
        if m:
            pairs.append((self.buf.rows(self.n, self.n + m), arr))
        self.n += m

    def assign(self, arr):
        if self._range is not None:                            # (thinning a large database: the rows move to the front of the range)
            self.n = 0
            self.reserve(arr.shape[0])
            if arr.shape[0]:
                self.buf.rows(0, arr.shape[0]).copy_from(arr)
            self.n = arr.shape[0]
            return
        self.buf = arr
        self.n = arr.shape[0]


class _HostGrowable:
    def __init__(self, dtype):
        self.buf = np.zeros(1024, dtype)
        self.n = 0

    def view(self, start=0):
        return self.buf[start:self.n]

    def append(self, arr):
        m = arr.shape[0]
        if self.n + m > self.buf.shape[0]:
            # room for 256 appends of the first one's size at the first growth (as the device buffers), doubling afterwards: a
            # doubling copies everything so far on the host (1.3 ms at 2.6 M entries: one slow iteration)
            first = self.buf.shape[0] <= 1024 and self.n == 0
            new = np.empty(max(2 * self.buf.shape[0], self.n + m, 256 * m if first else 0), self.buf.dtype)
            new[:self.n] = self.buf[:self.n]
            self.buf = new
        self.buf[self.n:self.n + m] = arr
        self.n += m

    def assign(self, arr):
        self.buf = np.array(arr, self.buf.dtype)
        self.n = self.buf.shape[0]


class SampleDB:
    def __init__(self, dim, diagonal_covariances, keep_samples, max_samples=None, ctx=None):
        self.ctx = ctx if ctx is not None else get_context()
        self._dim = int(dim)
        self.diagonal_covariances = bool(diagonal_covariances)                                 # :32
        self.keep_samples = keep_samples
        self.max_samples = max_samples
        d = self._dim
        self._samples = _Growable(self.ctx, (d,))
        self._target_lnpdfs = _Growable(self.ctx, ())
        self._target_grads = _Growable(self.ctx, (d,))
        self._mapping_dev = _Growable(self.ctx, (), np.int32)
        self._means = _Growable(self.ctx, (d,))
        self._chols = _Growable(self.ctx, (d,) if self.diagonal_covariances else (d, d))       # :36-41
        # component blocks of the snapshots: the diagonal kernels' blocks for a diagonal database (csrc/diag_sweep.hip)
        self._packed = _Growable(self.ctx, (hip_ops.diag_packed_stride(d) if self.diagonal_covariances
                                            else hip_ops.packed_stride(d),))
        self._mapping_host = _HostGrowable(np.int32)
        self._num_samples_written = 0
        # append log: (first sample, first component, per-component counts); cleared when the DB is thinned out
        self._segments = []
        # the last get_newest_samples answer: window [start, stop), its components / counts, its background density
        self._bg_cache = None
        # per-append partial densities of the sliding window (_sliding_background)
        self._pd = None
        self._epoch = 0                   # bumped whenever samples are re-indexed (thinning, non-keeping assign)
        self._seg_start, self._seg_size, self._seg_info = [], [], {}      # search keys / per-append component lists of the append log

    @staticmethod
    def build_from_config(config, num_dimensions):
        """sample_db.py:48-61."""
        return SampleDB(num_dimensions, config["model_initialization"]["use_diagonal_covs"],
                        config["use_sample_database"], config["max_database_size"])

    # ---- reference attributes (device views) ----------------------------------------------------------------
    @property
    def samples(self):
        return self._samples.view()

    @property
    def means(self):
        return self._means.view()

    @property
    def chols(self):
        return self._chols.view()

    @property
    def inv_chols(self):
        """sample_db.py:121: explicit inverses (API compatibility; the density kernels solve with L directly)."""
        if self.diagonal_covariances:                                                          # :119,:130
            return hip_ops.reciprocal(self.ctx, self.chols)
        if self._means.n == 0:
            return self.ctx.empty((0, self._dim, self._dim))
        return hip_ops.pack_components(self.ctx, self.means, self.chols, want_inverse=True)[1]

    @property
    def target_lnpdfs(self):
        return self._target_lnpdfs.view()

    @property
    def target_grads(self):
        return self._target_grads.view()

    @property
    def mapping(self):
        return self._mapping_dev.view()

    class _Counter:
        """num_samples_written with the ``.numpy()`` / ``assign_add`` the reference callers use."""
        def __init__(self, db):
            self._db = db
        def numpy(self):
            return self._db._num_samples_written
        def assign_add(self, n):
            self._db._num_samples_written += int(n)
        def __int__(self):
            return self._db._num_samples_written
        def __index__(self):
            return self._db._num_samples_written

    @property
    def num_samples_written(self):
        return SampleDB._Counter(self)

    def _pack(self, means, chols):
        """Component blocks of (means, chols) in the layout this database's density kernels read."""
        if self.diagonal_covariances:
            if chols.ndim != 2:
                raise ValueError("SampleDB(diagonal_covariances=True) got [K,D,D] Cholesky factors")
            return hip_ops.diag_pack(self.ctx, means, chols)
        return hip_ops.pack_components(self.ctx, means, self._dense(chols))[0]

    def _mixture_lp(self, packed, logw, xs):
        """log sum_j exp(logw_j) N(x; component j) over packed blocks of this database's layout."""
        if self.diagonal_covariances:
            return hip_ops.diag_mixture_eval(self.ctx, packed, logw, xs, self._dim, want_lp=True)[1]
        return hip_ops.mixture_eval(self.ctx, packed, logw, xs, self._dim, want_lp=True)[1]

    def _dense(self, chols):
        """Dense lower-triangular factors for the density kernels ([K,D] standard deviations -> diag(sigma))."""
        if chols.ndim == 2:
            if not self.diagonal_covariances:
                raise ValueError("SampleDB(diagonal_covariances=False) got [K,D] Cholesky factors")
            return hip_ops.diag_embed(self.ctx, chols)
        if self.diagonal_covariances:
            raise ValueError("SampleDB(diagonal_covariances=True) got [K,D,D] Cholesky factors")
        return chols

    # ---- mutation --------------------------------------------------------------------------------------------------
    def remove_every_nth_sample(self, N):
        """sample_db.py:63-79."""
        n = self._samples.n
        keep = np.arange(0, n, int(N), dtype=np.int32)
        idx = self.ctx.asarray(keep, np.int32)
        self._samples.assign(hip_ops.gather_rows(self.ctx, self.samples, idx))
        self._target_lnpdfs.assign(hip_ops.gather_rows(self.ctx, self.target_lnpdfs, idx))
        self._target_grads.assign(hip_ops.gather_rows(self.ctx, self.target_grads, idx))
        mp = self._mapping_host.view()[keep]
        uniq, first = np.unique(mp, return_index=True)
        used = uniq[np.argsort(first)].astype(np.int32)                       # tf.unique: first-occurrence order
        remap = np.full(int(mp.max()) + 1 if mp.size else 0, -1, np.int64)
        remap[used] = np.arange(len(used))
        new_mp = remap[mp].astype(np.int32)
        self._mapping_host.assign(new_mp)
        self._mapping_dev.assign(self.ctx.asarray(new_mp, np.int32))
        uidx = self.ctx.asarray(used, np.int32)
        self._means.assign(hip_ops.gather_rows(self.ctx, self.means, uidx))
        self._chols.assign(hip_ops.gather_rows(self.ctx, self.chols, uidx))
        self._packed.assign(hip_ops.gather_rows(self.ctx, self._packed.view(), uidx))
        self._segments = []
        self._bg_cache = None
        self._pd = None
        self._epoch += 1
        self._seg_start, self._seg_size, self._seg_info = [], [], {}

    def add_samples(self, samples, means, chols, target_lnpdfs, target_grads, mapping, mapping_host=None,
                    packed=None, counts=None):
        """sample_db.py:81-135.  ``mapping_host`` (NumPy int32), ``packed`` (component blocks of ``means``/``chols``)
        and ``counts`` (samples per component, component order) are optional accelerators used by the built-in
        sample selectors."""
        ctx = self.ctx
        samples = ctx.asarray(samples)
        n_new = samples.shape[0]
        if self.max_samples is not None and n_new + self._samples.n > self.max_samples:
            self.remove_every_nth_sample(2)                                                   # :111-112
        self._num_samples_written += n_new                                                     # :113
        means = ctx.asarray(means); chols = ctx.asarray(chols)
        if mapping_host is None:
            mapping_host = np.asarray(mapping.numpy() if isinstance(mapping, DeviceArray) else mapping, np.int32)
        if packed is None:
            packed = self._pack(means, chols)
        tl = ctx.asarray(target_lnpdfs); tg = ctx.asarray(target_grads)
        mapping = ctx.asarray(mapping, np.int32)
        if self.keep_samples:
            offset = self._means.n                                                             # :115
            if counts is not None:
                self._segments.append((self._samples.n, offset, np.asarray(counts, np.int64)))
            else:
                self._segments = []
                self._pd = None
                self._seg_start, self._seg_size, self._seg_info = [], [], {}
            self._mapping_host.append(mapping_host + offset)
            self._mapping_dev.reserve(n_new)
            if n_new:
                dst = self._mapping_dev.buf.rows(self._mapping_dev.n, self._mapping_dev.n + n_new)
                ctx.check(ctx.lib.gmmvi_add_scalar_i32(ctx.handle, dst.ptr, mapping.ptr, int(offset), n_new))
            self._mapping_dev.n += n_new
            pairs = []
            for grow, arr in ((self._means, means), (self._chols, chols), (self._packed, packed),
                              (self._samples, samples), (self._target_lnpdfs, tl), (self._target_grads, tg)):
                grow.append_deferred(arr, pairs)
            hip_ops.copy_batch(ctx, pairs)
        else:                                                                                  # :125-135
            self._segments = [(0, 0, np.asarray(counts, np.int64))] if counts is not None else []
            self._bg_cache = None
            self._pd = None
            self._epoch += 1
            self._seg_start, self._seg_size, self._seg_info = [], [], {}
            self._mapping_host.assign(mapping_host)
            self._mapping_dev.assign(mapping.copy())
            self._means.assign(means.copy()); self._chols.assign(chols.copy()); self._packed.assign(packed)
            self._samples.assign(samples); self._target_lnpdfs.assign(tl); self._target_grads.assign(tg)

    def get_random_sample(self, N, rng=None):
        """sample_db.py:137-152 (tf.random.shuffle + slice -> ``random_subset``: the same set law, NumPy generator)."""
        rng = np.random.default_rng() if rng is None else rng
        idx = random_subset(rng, self._samples.n, N)
        # the index list goes up in pieces that fit the pinned staging ring (gmmvi_upload: no wait for the stream); one 400 KB
        # copy would be synchronous and park the host until the iteration's kernels have drained
        didx = self.ctx.empty((idx.shape[0],), np.int32)
        piece = 16000
        for lo in range(0, idx.shape[0], piece):
            didx.rows(lo, min(lo + piece, idx.shape[0])).set(idx[lo:lo + piece])
        return hip_ops.gather_rows(self.ctx, self.samples, didx), hip_ops.gather_rows(self.ctx, self.target_lnpdfs, didx)

    # ---- background density ------------------------------------------------------------------------------------------
    def evaluate_background(self, weights, means, chols, inv_chols, samples):
        """sample_db.py:164-192: log sum_j w_j N(x; mu_j, Sigma_j).  ``inv_chols`` is accepted for signature
        compatibility; the kernel solves with ``chols``."""
        ctx = self.ctx
        means = ctx.asarray(means); chols = ctx.asarray(chols)
        packed = self._pack(means, chols)
        w = np.asarray(weights.numpy() if hasattr(weights, "numpy") else weights, np.float64)
        return self._mixture_lp(packed, ctx.asarray(np.log(w).astype(np.float32)), ctx.asarray(samples))

    def _active_components(self, start):
        """unique_with_counts of mapping[start:] in first-occurrence order (sample_db.py:221)."""
        # fast path: the window is a whole number of recorded appends
        acc_active, acc_counts = [], []
        for s0, c0, counts in reversed(self._segments):
            if s0 < start:
                break
            nz = counts > 0
            acc_active.append(c0 + np.nonzero(nz)[0])
            acc_counts.append(counts[nz])
            if s0 == start:
                return np.concatenate(acc_active[::-1]), np.concatenate(acc_counts[::-1])
        mp = self._mapping_host.view(start)
        uniq, first, counts = np.unique(mp, return_index=True, return_counts=True)
        order = np.argsort(first)
        return uniq[order], counts[order]

    def _window_components(self, start, stop):
        """unique_with_counts of mapping[start:stop] in first-occurrence order."""
        acc_active, acc_counts = [], []
        pos = start
        for s0, c0, counts in self._segments:
            if s0 == pos and s0 + int(counts.sum()) <= stop:
                nz = counts > 0
                acc_active.append(c0 + np.nonzero(nz)[0])
                acc_counts.append(counts[nz])
                pos = s0 + int(counts.sum())
                if pos == stop:
                    return np.concatenate(acc_active), np.concatenate(acc_counts)
        mp = self._mapping_host.view(start)[:stop - start]
        uniq, first, counts = np.unique(mp, return_index=True, return_counts=True)
        order = np.argsort(first)
        return uniq[order], counts[order]

    @staticmethod
    def log_shares(n_old, n_new):
        """float32 log(n_old / (n_old + n_new)), log(n_new / (n_old + n_new)): the weights that join the two halves of a window's
        background density."""
        tot = float(n_old + n_new)
        return float(np.float32(np.log(n_old / tot))), float(np.float32(np.log(n_new / tot)))

    def _rows_of(self, active):
        """Snapshot blocks of the components ``active`` (ascending DB rows), as a view when they are consecutive."""
        lo, hi = int(active[0]), int(active[-1]) + 1
        if hi - lo == len(active):
            return self._packed.view(lo, hi)
        return hip_ops.gather_rows(self.ctx, self._packed.view(), active.astype(np.int32))

    class _Mixture:
        """The mixture a window's samples came from: components (DB rows), their sample counts inside the window, and -- built
        on first use -- the device arrays the kernels take (log weights count / window size, snapshot blocks)."""
        def __init__(self, db, active, counts, logw_dev=None):
            self.db, self.active, self.counts, self._logw, self._packed = db, active, counts, logw_dev, None

        @property
        def logw_dev(self):
            if self._logw is None:
                c = self.counts
                self._logw = self.db.ctx.cached_const(
                    ("bg_logw", c.tobytes()), lambda: self.db.ctx.asarray(np.log(c.astype(np.float64) / c.sum()).astype(np.float32)))
            return self._logw

        @property
        def packed(self):
            if self._packed is None:
                a = self.active
                if len(a) > 1 and not np.all(np.diff(a) > 0):          # first-occurrence order that is not ascending
                    self._packed = hip_ops.gather_rows(self.db.ctx, self.db._packed.view(), a.astype(np.int32))
                else:
                    self._packed = self.db._rows_of(a)
            return self._packed

    def _extend_background(self, start, stop):
        """Background density of the window [start, stop) from the one of [start, cached stop) returned by the previous call
        (the effective-sample-size step of the sample selectors asks for the reused samples first, then -- after the append --
        for the reused and the new ones: sample_selector.py:160-219).  In exact arithmetic identical to evaluating the whole
        window's mixture on all of its samples (sample_db.py:216-227): that mixture is the share-weighted sum of the mixture the
        old samples came from (known for the old samples, evaluated for the new ones) and of the new samples' components.
        -> DeviceArray, or None when the cache does not apply."""
        c = self._bg_cache
        if c is None or c["start"] != start or not (c["stop"] < stop) or c["stop"] <= start:
            return None
        old = c["mix"]() if callable(c["mix"]) else c["mix"]
        new_active, new_counts = self._window_components(c["stop"], stop)
        if new_active.min() <= old.active.max():                     # a component with samples in both halves: no clean split
            return None
        ctx = self.ctx
        n_old, n_new = c["stop"] - start, stop - c["stop"]
        new = SampleDB._Mixture(self, new_active, new_counts)
        part_old = hip_ops.concat(ctx, [c["bg"], self._mixture_lp(old.packed, old.logw_dev, self._samples.view(c["stop"]))])
        part_new = self._mixture_lp(new.packed, new.logw_dev, self._samples.view(start))
        ca, cb = self.log_shares(n_old, n_new)
        return hip_ops.logaddexp(ctx, part_old, ca, part_new, cb)

    # the partial densities of a window hold at most this many floats of component log densities at a time
    _PD_MAX_LD_FLOATS = 1 << 27

    def _segment_info(self, idx, logW):
        """(DB rows, counts, float32 log(count / W)) of the components with samples in append ``idx``."""
        key = (idx, logW)
        info = self._seg_info.get(key)
        if info is None:
            s0, c0, counts = self._segments[idx]
            nz = np.nonzero(counts > 0)[0]
            cnt = counts[nz]
            info = (c0 + nz, cnt, (np.log(cnt.astype(np.float64)) - logW).astype(np.float32))
            self._seg_info[key] = info
        return info

    def _sliding_background(self, start, stop):
        """Background density of the window [start, stop) of FIXED length (the reuse window once the database is longer than it)
        from per-append partial densities.  The window's mixture weights a component by its samples inside the window
        (sample_db.py:221-226); for an append g that lies wholly inside, those are the append's own counts, so
        P_g[n] = log sum_{j in g} count_j / W  N_j(x_n) does not depend on where the window starts: the rows of such appends are
        kept from call to call (the surviving block is moved, the new samples' columns and the new appends' rows are evaluated);
        only the append the window cuts through is evaluated again with its counts inside the window.  The answer is the
        log-sum-exp of the rows: the same mixture, summed in groups.  -> (DeviceArray, mixture) or None (caller evaluates from
        scratch).  Host work per call is O(appends entering or leaving the window), not O(appends inside)."""
        ctx, W = self.ctx, stop - start
        segs = self._segments
        if self.diagonal_covariances or not segs or W <= 0:
            return None
        if len(self._seg_start) != len(segs):                           # append log grew: extend the search keys
            for s0, c0, counts in segs[len(self._seg_start):]:
                self._seg_start.append(s0)
                self._seg_size.append(int(counts.sum()))
        hi = len(segs) - 1
        if self._seg_start[hi] + self._seg_size[hi] != stop:
            return None
        import bisect
        lo = bisect.bisect_left(self._seg_start, start)                 # first append that starts inside the window
        cut = None
        if lo == 0:
            if self._seg_start[0] != start:
                return None                                             # samples in front of the recorded appends
        elif self._seg_start[lo - 1] + self._seg_size[lo - 1] > start:
            cut = lo - 1
        if lo > hi and cut is None:
            return None
        logW = float(np.log(float(W)))
        if len(self._seg_info) > 4 * (hi - lo + 8):
            self._seg_info = {k: v for k, v in self._seg_info.items() if k[0] >= lo}
        pd = self._pd
        reuse = (pd is not None and pd["epoch"] == self._epoch and pd["W"] == W and pd["start"] <= start
                 and pd["stop"] <= stop and pd["stop"] > start and pd["hi"] >= lo and pd["lo"] <= lo)
        n_inside = max(0, hi - lo + 1)
        n_rows = n_inside + (1 if cut is not None else 0)
        P = ctx.empty((n_rows, W))
        xs = self._samples.view(start)
        if reuse:
            n_kept = pd["hi"] - lo + 1
            drop_segs = lo - pd["lo"]
            drop_comps = int(pd["nseg"][:drop_segs].sum())
            k_active, k_counts, k_nseg = pd["active"][drop_comps:], pd["counts"][drop_comps:], pd["nseg"][drop_segs:]
            k_logw = pd["logw_dev"].rows(drop_comps, pd["logw_dev"].shape[0])
            keep_cols = pd["stop"] - start
            hip_ops.copy_2d(ctx, P, 0, 0, pd["P"], drop_segs, start - pd["start"], n_kept, keep_cols)
            new_cols = W - keep_cols
            if new_cols > 0:
                if len(k_active) * new_cols > self._PD_MAX_LD_FLOATS:
                    return None
                offsets = np.concatenate([[0], np.cumsum(k_nseg)]).astype(np.int32)
                ld = hip_ops.mixture_eval(ctx, self._rows_of(k_active), k_logw, self._samples.view(pd["stop"]), self._dim,
                                          want_ld=True, want_lp=False)[0]
                hip_ops.segment_lse_into(ctx, P, keep_cols, ctx.asarray(offsets, np.int32), k_logw, ld)
            fresh = range(pd["hi"] + 1, hi + 1)
        else:
            n_kept = 0
            k_active = k_counts = np.zeros(0, np.int64)
            k_nseg = np.zeros(0, np.int64)
            k_logw = None
            fresh = range(lo, hi + 1)
        # rows of the appends that are new to the window (normally one): their mixture on all window samples
        infos = [self._segment_info(i, logW) for i in fresh]
        row = n_kept
        f_logw_dev = []
        if len(infos) > 2:
            active = np.concatenate([i[0] for i in infos])
            if len(active) * W > self._PD_MAX_LD_FLOATS:
                return None
            offsets = np.concatenate([[0], np.cumsum([len(i[0]) for i in infos])]).astype(np.int32)
            lw = ctx.asarray(np.concatenate([i[2] for i in infos]))
            f_logw_dev.append(lw)
            ld = hip_ops.mixture_eval(ctx, self._rows_of(active), lw, xs, self._dim, want_ld=True, want_lp=False)[0]
            hip_ops.segment_lse_into(ctx, P.rows(row, row + len(infos)), 0, ctx.asarray(offsets, np.int32), lw, ld)
            row += len(infos)
        else:
            for act, cnt, lw_host in infos:
                lw = ctx.asarray(lw_host)
                f_logw_dev.append(lw)
                P.rows(row, row + 1).reshape(W).copy_from(self._mixture_lp(self._rows_of(act), lw, xs))
                row += 1
        # the append the window cuts through, with its counts inside the window
        cut_part = None
        if cut is not None:
            mp = self._mapping_host.view(start)[:self._seg_start[cut] + self._seg_size[cut] - start]
            uniq, cnt = np.unique(mp, return_counts=True)
            lw = ctx.asarray((np.log(cnt.astype(np.float64)) - logW).astype(np.float32))
            P.rows(row, row + 1).reshape(W).copy_from(self._mixture_lp(self._rows_of(uniq), lw, xs))
            cut_part = (uniq.astype(np.int64), cnt.astype(np.int64), lw)
            row += 1
        assert row == n_rows
        active = np.concatenate([k_active] + [i[0] for i in infos]) if infos else k_active
        counts = np.concatenate([k_counts] + [i[1] for i in infos]) if infos else k_counts
        nseg = np.concatenate([k_nseg, np.array([len(i[0]) for i in infos], np.int64)])
        parts = ([k_logw] if k_logw is not None and k_logw.shape[0] else []) + f_logw_dev
        logw_dev = parts[0] if len(parts) == 1 else (hip_ops.concat(ctx, parts) if parts else None)
        self._pd = {"epoch": self._epoch, "W": W, "start": start, "stop": stop, "lo": lo, "hi": hi, "P": P, "active": active,
                    "counts": counts, "nseg": nseg, "logw_dev": logw_dev}
        bg = P.reshape(W).copy() if n_rows == 1 else hip_ops.combine_partials(ctx, P, None, self._dim)[0]

        def mixture():
            # oldest first: the cut append's components, then the appends inside
            if cut_part is None:
                return SampleDB._Mixture(self, active, counts, logw_dev)
            return SampleDB._Mixture(self, np.concatenate([cut_part[0], active]), np.concatenate([cut_part[1], counts]),
                                     hip_ops.concat(ctx, [cut_part[2], logw_dev]) if logw_dev is not None else cut_part[2])
        return bg, mixture

    def get_newest_samples(self, N, fuse_with_model=None):
        """sample_db.py:194-228 -> (log_pdfs, samples, mapping, target_lnpdfs, target_grads).
        ``fuse_with_model``: a model whose CURRENT components are exactly the snapshot components of the requested
        window (the caller guarantees it: newest append, reuse ratio 0); the background density is then computed
        together with the model's density/gradient in one sweep (GMM.eval_with_background)."""
        ctx, d = self.ctx, self._dim
        N = int(N)
        if self._samples.n == 0 or N == 0:                                                     # :213-214
            return (ctx.empty((0,)), ctx.empty((0, d)), ctx.empty((0,), np.int32), ctx.empty((0,)),
                    ctx.empty((0, d)))
        start = max(0, self._samples.n - N)                                                    # :216
        stop = self._samples.n
        xs = self._samples.view(start)
        whole = None                    # the window's components and counts, worked out only by the routes that need them

        def window_mixture():
            nonlocal whole
            if whole is None:
                whole = SampleDB._Mixture(self, *self._window_components(start, stop))         # :221, :225-226
            return whole

        bg, mix, extended = None, window_mixture, False
        if fuse_with_model is not None and self._segments and self._segments[-1][0] == start:
            if len(window_mixture().active) == fuse_with_model.num_components:
                bg = fuse_with_model.eval_with_background(xs, window_mixture().logw_dev)
        if bg is None:
            bg = self._extend_background(start, stop)
            extended = bg is not None
        if bg is None and start > 0:                           # a full-length window that slides: per-append partial densities
            ans = self._sliding_background(start, stop)
            if ans is not None:
                bg, mix = ans
        if bg is None:
            m = window_mixture()
            bg = self._mixture_lp(m.packed, m.logw_dev, xs)                                    # :227
        # an extended answer is not extended again (the single-call iteration, which extends inside its C call, keeps no
        # array to extend from either: both paths then recompute the next window from scratch and stay bit-equal)
        self._bg_cache = None if extended else {"start": start, "stop": stop, "bg": bg, "mix": mix}
        return (bg, xs, self._mapping_dev.view(start), self._target_lnpdfs.view(start),
                self._target_grads.view(start))

    def newest_mapping_host(self, N):
        start = max(0, self._samples.n - int(N))
        return self._mapping_host.view(start)
