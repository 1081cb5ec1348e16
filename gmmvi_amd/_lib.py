"""ctypes binding of libgmmvi_hip.so (C ABI: include/gmmvi_hip.h).

The product path has no CPU fallback: if the shared library is missing, or a call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMMVI_HIP_LIB", os.path.join(_HERE, "libgmmvi_hip.so"))    # override: experiments only

GAUSS, STUDENT_T = 0, 1
SELF_NORMALIZED, OWN_SAMPLES_ONLY, EXPLICIT_ESTIMATE = 1, 2, 4
MAX_DIM = 64
MORE_REGISTER_MAX_DIM = 21     # gmmvi_more: register-resident ridge system up to here, the tiled route above (D <= 63)
BLOCKED_ABOVE_DEFAULT = 50     # csrc/blocked.h: D > 50 runs the blocked (MFMA) kernels


_blocked_above = None


def _atoi(text):
    """C atoi: optional sign and leading digits, anything else ends the number (no digits: 0)."""
    import re
    m = re.match(r"\s*([+-]?\d+)", text)
    return int(m.group(1)) if m else 0


def blocked_above():
    """Dimensions above this take the blocked (MFMA) path.  Mirrors csrc/blocked.h gmmvi_blocked_above(): the environment
    knob GMMVI_BLOCKED_ABOVE is read ONCE per process (the library keeps it in a static), parsed as atoi does, clamped to
    16..64."""
    global _blocked_above
    if _blocked_above is None:
        raw = os.environ.get("GMMVI_BLOCKED_ABOVE")
        t = BLOCKED_ABOVE_DEFAULT if raw is None else _atoi(raw)
        _blocked_above = min(max(t, 16), MAX_DIM)
    return _blocked_above


class GmmviError(RuntimeError):
    pass


_lib = None

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_sz = C.c_size_t
_u64 = C.c_uint64

_PROTOS = {
    "gmmvi_device_count": (_i, []),
    "gmmvi_ctx_create": (_i, [C.POINTER(_p), _i]),
    "gmmvi_ctx_destroy": (None, [_p]),
    "gmmvi_last_error": (C.c_char_p, [_p]),
    "gmmvi_sync": (_i, [_p]),
    "gmmvi_malloc": (_i, [_p, _sz, C.POINTER(_p)]),
    "gmmvi_free": (_i, [_p, _p]),
    "gmmvi_upload": (_i, [_p, _p, _p, _sz]),
    "gmmvi_download": (_i, [_p, _p, _p, _sz]),
    "gmmvi_copy": (_i, [_p, _p, _p, _sz]),
    "gmmvi_fill_f32": (_i, [_p, _p, _f, _sz]),
    "gmmvi_gather_rows": (_i, [_p, _p, _p, _i, _i, _p]),
    "gmmvi_exp_f32": (_i, [_p, _p, _p, _sz]),
    "gmmvi_logaddexp_f32": (_i, [_p, _p, _p, _f, _p, _f, _sz]),
    "gmmvi_segment_lse_f32": (_i, [_p, _i, _p, _p, _p, _i, _p, _sz, _i]),
    "gmmvi_copy_2d_f32": (_i, [_p, _p, _sz, _p, _sz, _i, _i]),
    "gmmvi_copy_batch": (_i, [_p, _i, C.POINTER(_p), C.POINTER(_p), C.POINTER(_sz)]),
    "gmmvi_normalize_logw": (_i, [_p, _p, _i, _p]),
    "gmmvi_add_heuristic_argmax": (_i, [_p, _p, _p, _i, C.c_double, _p]),
    "gmmvi_unpack_gathered": (_i, [_p, _p, _i, _sz, _i, C.POINTER(_sz), C.POINTER(_p)]),
    "gmmvi_fill_strided_f32": (_i, [_p, _p, _sz, _sz, _f]),
    "gmmvi_remove_column_f32": (_i, [_p, _p, _i, _sz, _i, _i]),
    "gmmvi_add_scalar_i32": (_i, [_p, _p, _p, C.c_int32, _sz]),
    "gmmvi_event_create": (_i, [_p, C.POINTER(_p)]),
    "gmmvi_event_destroy": (_i, [_p, _p]),
    "gmmvi_event_record": (_i, [_p, _p]),
    "gmmvi_event_synchronize": (_i, [_p, _p]),
    "gmmvi_host_alloc": (_i, [_p, _sz, C.POINTER(_p)]),
    "gmmvi_vmm_reserve": (_i, [_p, _sz, C.POINTER(_p), C.POINTER(_sz)]),
    "gmmvi_vmm_grow": (_i, [_p, _p, _sz, _sz, _sz]),
    "gmmvi_vmm_release": (_i, [_p, _p, _sz, _sz, _sz]),
    "gmmvi_host_free": (_i, [_p, _p]),
    "gmmvi_download_async": (_i, [_p, _p, _p, _sz]),
    "gmmvi_event_elapsed_ms": (_i, [_p, _p, _p, C.POINTER(_f)]),
    "gmmvi_profile_enable": (_i, [_p, _i]),
    "gmmvi_profile_report": (_i, [_p, C.c_char_p, _sz]),
    "gmmvi_packed_stride": (_sz, [_i]),
    "gmmvi_pack_components": (_i, [_p, _i, _f, _i, _i, _p, _p, _p, _p]),
    "gmmvi_cholesky": (_i, [_p, _i, _i, _p, _p, _p]),
    "gmmvi_mixture_eval": (_i, [_p, _i, _f, _i, _i, _p, _p, _p, _i, _p, _p, _p]),
    "gmmvi_mixture_eval_dual": (_i, [_p, _i, _f, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "gmmvi_target_planar": (_i, [_p, _i, _p, _i, _p, _f, _p, _i, _p, _p]),
    "gmmvi_sample_components": (_i, [_p, _i, _i, _p, _p, _p, _i, _u64, _u64, _i, _p, _p, _p]),
    "gmmvi_philox_normals": (_i, [_p, _u64, _u64, _i, _i, _i, _p]),
    "gmmvi_philox_uniforms": (_i, [_p, _u64, _u64, _i, _i, _p]),
    "gmmvi_stein": (_i, [_p, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p]),
    "gmmvi_more": (_i, [_p, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "gmmvi_update_components_kl": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p]),
    "gmmvi_update_components_kl_reference": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p]),
    "gmmvi_update_components_direct": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p]),
    "gmmvi_update_components_iblr": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p]),
    "gmmvi_expected_log_ratios": (_i, [_p, _i, _i, _p, _p, _p, _p, _f, _p, _i, _p, _p, _p]),
    "gmmvi_update_weights_kl": (_i, [_p, _i, _p, _p, _p, _f, _p]),
    "gmmvi_update_weights_direct": (_i, [_p, _i, _p, _p, _p, _f]),
    "gmmvi_component_stepsize_improvement": (_i, [_p, _i, _p, _p, _p, _f, _f, _f, _f]),
    "gmmvi_weight_stepsize_improvement": (_i, [_p, _i, _p, _p, _p, _f, _f, _f, _f]),
    "gmmvi_train_iter_samtron": (_i, [_p, _p]),
    "gmmvi_sharded_scratch_floats": (_sz, [_i, _i, _i]),
    "gmmvi_train_iter_sharded_phase": (_i, [_p, _p, _i]),
    "gmmvi_comm_unique_id": (_i, [C.c_char_p]),
    "gmmvi_comm_init": (_i, [_p, C.c_char_p, _i, _i]),
    "gmmvi_comm_destroy": (_i, [_p]),
    "gmmvi_allgather_f32": (_i, [_p, _p, _p, _sz]),
    "gmmvi_allreduce_f32": (_i, [_p, _p, _sz, _i]),
    "gmmvi_combine_partials": (_i, [_p, _i, _i, _i, _p, _p, _p, _p]),
    "gmmvi_diag_packed_stride": (_sz, [_i]),
    "gmmvi_diag_pack": (_i, [_p, _i, _i, _p, _p, _p]),
    "gmmvi_diag_mixture_eval": (_i, [_p, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "gmmvi_diag_sample": (_i, [_p, _i, _i, _p, _p, _p, _i, _u64, _u64, _i, _p, _p, _p]),
    "gmmvi_diag_stein": (_i, [_p, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p]),
    "gmmvi_diag_embed": (_i, [_p, _i, _i, _p, _p]),
    "gmmvi_diag_extract": (_i, [_p, _i, _i, _p, _p]),
    "gmmvi_reciprocal_f32": (_i, [_p, _p, _sz, _p]),
    "gmmvi_update_components_diag_kl": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p]),
    "gmmvi_update_components_diag_iblr": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p]),
    "gmmvi_mmd_scratch_doubles": (_sz, [_i, _i]),
    "gmmvi_mmd_pair_sum": (_i, [_p, _p, _i, _p, _i, _i, _p, _p, _p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS.keys())


def load():
    """Load the shared library (once).  Raises GmmviError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GmmviError(
            f"{LIB_PATH} not found: build it with `make -C gmmvi_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback for the gmmvi hot path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def device_count():
    return int(load().gmmvi_device_count())


def check(ctx_handle, rc):
    if rc != 0:
        msg = load().gmmvi_last_error(ctx_handle)
        raise GmmviError(f"libgmmvi_hip error {rc}: {msg.decode() if msg else '?'}")
