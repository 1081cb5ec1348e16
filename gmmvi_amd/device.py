"""Device context and device-resident arrays for the gmmvi hot path.

``DeviceArray`` plays the role the reference's ``tf.Tensor`` / ``tf.Variable`` plays at the plug-in boundary: the
modules hand each other device-resident arrays and only ``.numpy()`` copies to the host
(reference callers do the same, e.g. gmmvi_runner.py:110-116, examples/4_...:29-30).
"""
import ctypes as C
import os

import math

import numpy as np

from . import _lib

_CTX = None


class Context:
    """One HIP context (device + stream + workspace [+ RCCL communicator]) per process."""

    def __init__(self, device=None):
        lib = _lib.load()
        if device is None:
            device = int(os.environ.get("GMMVI_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = lib.gmmvi_device_count()
            if n > 0 and device >= n:
                # never wrap: two ranks on one device would only fail later, inside RCCL, with "duplicate device"
                raise _lib.GmmviError(f"rank-local device index {device} (GMMVI_DEVICE / LOCAL_RANK) but this process sees only "
                                      f"{n} GPU(s): start one rank per visible GPU")
        h = C.c_void_p()
        rc = lib.gmmvi_ctx_create(C.byref(h), int(device))
        if rc != 0:
            msg = lib.gmmvi_last_error(None)
            raise _lib.GmmviError(f"cannot create HIP context on device {device}: {msg.decode() if msg else rc}")
        self.lib = lib
        self.handle = h
        self.device = device
        self.n_ranks = 1
        self.rank = 0
        # Stream-ordered caching allocator: every kernel of this context runs on ONE stream, so a block released
        # by the host can be handed to a later launch without synchronising (hipMalloc / hipFree would stall the
        # stream every iteration).  Blocks are bucketed by size rounded up to 512 B.
        self._pool = {}
        self._pool_bytes = 0
        self._small_cache = {}

    def _alloc_bytes(self, nbytes):
        size = max(512, (int(nbytes) + 511) // 512 * 512)
        free = self._pool.get(size)
        if free:
            self._pool_bytes -= size
            return free.pop(), size
        p = C.c_void_p()
        self.check(self.lib.gmmvi_malloc(self.handle, size, C.byref(p)))
        return p.value, size

    def _release_bytes(self, ptr, size):
        if self._pool_bytes + size > (8 << 30):           # keep at most 8 GiB parked
            self.lib.gmmvi_free(self.handle, ptr)
            return
        self._pool.setdefault(size, []).append(ptr)
        self._pool_bytes += size

    def cached_const(self, key, builder):
        """Small read-only device arrays (offset tables, log-count weights) uploaded once and reused."""
        a = self._small_cache.get(key)
        if a is None:
            if len(self._small_cache) > 256:
                self._small_cache.clear()
            a = self._small_cache[key] = builder()
        return a

    def check(self, rc):
        if rc != 0:
            _lib.check(self.handle, rc)

    def sync(self):
        self.check(self.lib.gmmvi_sync(self.handle))

    # ---- arrays ------------------------------------------------------------------------------------------
    def empty(self, shape, dtype=np.float32):
        return DeviceArray._alloc(self, shape, dtype)

    def zeros(self, shape, dtype=np.float32):
        a = self.empty(shape, dtype)
        if a.size:
            if a.dtype == np.float32:
                self.check(self.lib.gmmvi_fill_f32(self.handle, a.ptr, 0.0, a.size))
            else:
                a.set(np.zeros(a.shape, a.dtype))
        return a

    def full(self, shape, value, dtype=np.float32):
        a = self.empty(shape, dtype)
        if a.size:
            if a.dtype == np.float32:
                self.check(self.lib.gmmvi_fill_f32(self.handle, a.ptr, float(value), a.size))
            else:
                a.set(np.full(a.shape, value, a.dtype))
        return a

    def asarray(self, x, dtype=np.float32):
        """DeviceArray for x (no copy when x already is one of the right dtype)."""
        if isinstance(x, DeviceArray):
            if x.dtype == np.dtype(dtype):
                return x
            return self.asarray(x.numpy().astype(dtype), dtype)
        if hasattr(x, "numpy") and not isinstance(x, np.ndarray):
            x = x.numpy()
        host = np.ascontiguousarray(np.asarray(x), dtype=dtype)
        a = self.empty(host.shape, dtype)
        a.set(host)
        return a

    # ---- events (bench) ------------------------------------------------------------------------------------
    def event(self):
        e = C.c_void_p()
        self.check(self.lib.gmmvi_event_create(self.handle, C.byref(e)))
        return e

    def record(self, ev):
        self.check(self.lib.gmmvi_event_record(self.handle, ev))

    def elapsed_ms(self, start, stop):
        ms = C.c_float()
        self.check(self.lib.gmmvi_event_elapsed_ms(self.handle, start, stop, C.byref(ms)))
        return float(ms.value)


def get_context():
    """Process-wide context (created on first use; fails loudly without a GPU or the library)."""
    global _CTX
    if _CTX is None:
        _CTX = Context()
    return _CTX


_F32, _I32 = np.dtype(np.float32), np.dtype(np.int32)
_DTYPES = {np.float32: _F32, np.int32: _I32, _F32: _F32, _I32: _I32}


def _as_shape(shape):
    """Tuple of Python ints (NumPy integers would leak into pointer arithmetic)."""
    if type(shape) is tuple:
        for v in shape:
            if type(v) is not int:
                return tuple(int(v) for v in shape)
        return shape
    if isinstance(shape, (int, np.integer)):
        return (int(shape),)
    return tuple(int(v) for v in shape)


class DeviceArray:
    """Dense row-major fp32 / int32 array in HBM."""
    __slots__ = ("ctx", "ptr", "shape", "dtype", "_owner", "_base", "_cap", "__weakref__")
    __array_priority__ = 100

    def __init__(self):
        raise TypeError("use Context.empty / Context.asarray")

    # (these run a few dozen times per iteration on the module-by-module and the sample-reuse paths: the common cases -- a tuple
    # of Python ints, np.float32 / np.int32 -- skip the normalisation)
    @classmethod
    def _alloc(cls, ctx, shape, dtype):
        self = object.__new__(cls)
        shape = _as_shape(shape)
        dt = _DTYPES.get(dtype)
        if dt is None:
            dt = np.dtype(dtype)
            if dt != _F32 and dt != _I32:
                raise TypeError(f"DeviceArray supports float32 and int32, not {dt}")
        ptr, cap = ctx._alloc_bytes(math.prod(shape) * 4)
        self.ctx, self.ptr, self.shape, self.dtype, self._owner, self._base = ctx, ptr, shape, dt, True, None
        self._cap = cap
        return self

    @classmethod
    def _view(cls, base, ptr, shape):
        self = object.__new__(cls)
        self.ctx, self.ptr, self.shape, self.dtype = base.ctx, ptr, _as_shape(shape), base.dtype
        self._owner, self._base, self._cap = False, base, 0
        return self

    @classmethod
    def _external(cls, ctx, ptr, shape, dtype, keepalive):
        """Array over memory this class does not manage (a mapped range of optimization/sample_db.py's grow-in-place buffers);
        ``keepalive`` owns it and stays referenced by this array and every view of it."""
        self = object.__new__(cls)
        self.ctx, self.ptr, self.shape, self.dtype = ctx, ptr, _as_shape(shape), np.dtype(dtype)
        self._owner, self._base, self._cap = False, keepalive, 0
        return self

    def __del__(self):
        try:
            if getattr(self, "_owner", False) and self.ptr:
                self.ctx._release_bytes(self.ptr, self._cap)
                self.ptr = None
        except Exception:
            pass

    # ---- metadata -----------------------------------------------------------------------------------------
    @property
    def size(self):
        return math.prod(self.shape)

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def nbytes(self):
        return self.size * 4

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype})"

    # ---- host <-> device ----------------------------------------------------------------------------------
    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if out.size:
            self.ctx.check(self.ctx.lib.gmmvi_download(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.shape != self.shape:
            raise ValueError(f"shape mismatch: {host.shape} vs {self.shape}")
        if host.size:
            self.ctx.check(self.ctx.lib.gmmvi_upload(self.ctx.handle, self.ptr, host.ctypes.data, host.nbytes))
        return self

    def copy(self):
        out = self.ctx.empty(self.shape, self.dtype)
        if self.size:
            self.ctx.check(self.ctx.lib.gmmvi_copy(self.ctx.handle, out.ptr, self.ptr, self.nbytes))
        return out

    def copy_from(self, other):
        if other.shape != self.shape or other.dtype != self.dtype:
            raise ValueError("copy_from: shape/dtype mismatch")
        if self.size:
            self.ctx.check(self.ctx.lib.gmmvi_copy(self.ctx.handle, self.ptr, other.ptr, self.nbytes))
        return self

    # ---- views --------------------------------------------------------------------------------------------
    def rows(self, start, stop=None):
        """View of rows [start, stop) along the first axis (no copy)."""
        n = self.shape[0]
        stop = n if stop is None else stop
        if not (0 <= start <= stop <= n):
            raise IndexError(f"rows({start}, {stop}) out of range for {self.shape}")
        inner = math.prod(self.shape[1:])
        return DeviceArray._view(self, self.ptr + start * inner * 4, (stop - start,) + self.shape[1:])

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and not isinstance(shape[0], (int, np.integer)) else shape
        shape = _as_shape(shape)
        if -1 in shape and shape.count(-1) == 1:
            known = math.prod(s for s in shape if s != -1)
            shape = tuple(self.size // max(known, 1) if s == -1 else s for s in shape)
        if math.prod(shape) != self.size:
            raise ValueError("reshape: size mismatch")
        return DeviceArray._view(self, self.ptr, shape)

    def __getitem__(self, idx):
        # convenience for callers of the plug-in surface (examples index samples with NumPy syntax)
        return self.numpy()[idx]

    # arithmetic falls back to NumPy on the host (metrics / plots only; never on the hot path)
    def _np(self, other):
        return other.numpy() if isinstance(other, DeviceArray) else other

    def __add__(self, o): return self.numpy() + self._np(o)
    def __radd__(self, o): return self._np(o) + self.numpy()
    def __sub__(self, o): return self.numpy() - self._np(o)
    def __rsub__(self, o): return self._np(o) - self.numpy()
    def __mul__(self, o): return self.numpy() * self._np(o)
    def __rmul__(self, o): return self._np(o) * self.numpy()
    def __truediv__(self, o): return self.numpy() / self._np(o)
    def __neg__(self): return -self.numpy()


def concat_rows(ctx, arrays):
    """Concatenate along axis 0 on the device."""
    arrays = [a for a in arrays if a.shape[0] > 0]
    if not arrays:
        raise ValueError("concat_rows: nothing to concatenate")
    inner = arrays[0].shape[1:]
    total = sum(a.shape[0] for a in arrays)
    out = ctx.empty((total,) + inner, arrays[0].dtype)
    ofs = 0
    for a in arrays:
        out.rows(ofs, ofs + a.shape[0]).copy_from(a)
        ofs += a.shape[0]
    return out
