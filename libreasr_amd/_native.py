"""
ctypes binding of liblasr_hip.so (include/lasr.h).  There is NO CPU fallback: if the HIP
library is missing or cannot be loaded this module raises, loudly.

torch is imported first on purpose: its bundled libamdhip64.so (SONAME libamdhip64.so.7) is then
already mapped, so liblasr_hip.so binds to the SAME HIP runtime and device pointers / streams are
interchangeable with PyTorch-ROCm tensors.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below, see docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LASR_LIB") or os.path.join(_HERE, "csrc", "liblasr_hip.so")   # LASR_LIB: A/B builds

LASR_OK, LASR_EINVAL, LASR_ENOMEM, LASR_EHIP, LASR_ESTATE, LASR_EFULL = 0, -1, -2, -3, -4, -5
LASR_PUSH_PINNED_NOCOPY = 1
LASR_PUSH_DEVICE_STABLE = 2
LASR_RESET_IF_DECODED = 16


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "feat", "hidden", "enc_layers", "pred_layers", "pred_cell", "embed", "joint", "vocab",
        "blank", "bos", "n_fft", "win", "hop", "n_mels", "n_stack", "stride", "n_buffer",
        "n_window", "chunk", "sample_rate", "dtype", "max_streams", "max_iters_offline",
        "max_iters_stream", "beam")]


class StepStats(C.Structure):
    _fields_ = [("frontend_ms", C.c_double), ("encoder_ms", C.c_double), ("decode_ms", C.c_double),
                ("decode_iters", C.c_int32), ("frames", C.c_int32), ("cell_ms", C.c_double),
                ("cell_launches", C.c_int32)]


# every symbol include/lasr.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("lasr_default_desc", None, [C.POINTER(ModelDesc)]),
    ("lasr_weight_count", C.c_size_t, [C.POINTER(ModelDesc)]),
    ("lasr_create", C.c_int, [C.c_int, C.POINTER(ModelDesc), _P, C.c_size_t, _P, C.POINTER(_P)]),
    ("lasr_destroy", None, [_P]),
    ("lasr_last_error", C.c_char_p, [_P]),
    ("lasr_stream_open", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("lasr_stream_reset", C.c_int, [_P, C.c_int, C.c_int]),
    ("lasr_stream_reset_many", C.c_int, [_P, _P, C.c_int, C.c_int]),
    ("lasr_stream_close", C.c_int, [_P, C.c_int]),
    ("lasr_push_pcm", C.c_int, [_P, _P, C.c_int, _P]),
    ("lasr_push_pcm_ex", C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.POINTER(C.c_longlong)]),
    ("lasr_push_consumed", C.c_int, [_P, C.c_longlong]),
    ("lasr_push_submit", C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.POINTER(C.c_longlong)]),
    ("lasr_push_submit_rows", C.c_int, [_P, _P, C.c_int, _P, C.POINTER(C.c_longlong)]),
    ("lasr_step_stream", C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    ("lasr_step_window", C.c_int, [_P, _P, C.c_int, _P, C.c_int64, C.c_int, C.POINTER(C.c_int)]),
    ("lasr_step_submit", C.c_int, [_P, _P, C.c_int]),
    ("lasr_step_wait", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("lasr_peek_slot", C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("lasr_peek_many", C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P, _P]),
    ("lasr_step_pending", C.c_int, [_P]),
    ("lasr_max_inflight", C.c_int, [_P]),
    ("lasr_transcribe_pcm", C.c_int, [_P, _P, C.c_int, _P, _P]),
    ("lasr_transcribe_feats", C.c_int, [_P, _P, C.c_int, _P, _P]),
    ("lasr_step_feats", C.c_int, [_P, _P, C.c_int, _P, C.c_int]),
    ("lasr_fetch", C.c_int, [_P, C.c_int, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                             C.POINTER(C.c_double)]),
    ("lasr_fetch_many", C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P]),
    ("lasr_logmel", C.c_int, [_P, _P, C.c_int, C.c_int64, _P]),
    ("lasr_stack", C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.POINTER(C.c_int)]),
    ("lasr_encoder", C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    ("lasr_predictor", C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    ("lasr_joint", C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P]),
    ("lasr_get_stats", C.c_int, [_P, C.POINTER(StepStats)]),
    ("lasr_set_profiling", C.c_int, [_P, C.c_int]),
    ("lasr_sync", C.c_int, [_P]),
    ("lasr_resample", C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, _P, C.POINTER(C.c_int64)]),
    ("lasr_lm_weight_count", C.c_size_t, [_P]),
    ("lasr_attach_lm", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("lasr_attach_lm_int8", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("lasr_front_create", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    ("lasr_front_stop", C.c_int, [_P]),
    ("lasr_front_destroy", None, [_P]),
    ("lasr_front_open", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("lasr_front_push", C.c_int, [_P, C.c_int, _P, C.c_int]),
    ("lasr_front_eof", C.c_int, [_P, C.c_int]),
    ("lasr_front_next", C.c_int, [_P, C.c_int, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    ("lasr_front_close", C.c_int, [_P, C.c_int]),
    ("lasr_front_set_empty_tokens", C.c_int, [_P, _P, C.c_int]),
    ("lasr_front_pause", C.c_int, [_P]),
    ("lasr_front_resume", C.c_int, [_P]),
    ("lasr_front_stats", C.c_int, [_P, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    ("lasr_front_error", C.c_char_p, [_P]),
]

# measurement / debug / experiment entry points: include/lasr_debug.h (not part of the drop-in surface)
DEBUG_SYMBOLS = [
    ("lasr_debug_timing", C.c_int, [_P, _P]),
    ("lasr_debug_read", C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("lasr_debug_enclog", C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_int)]),
    ("lasr_debug_fe_race", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("lasr_bench_cell", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("lasr_overlap_probe", C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    ("lasr_bench_neighbour", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("lasr_cell_prof", C.c_int, [_P, C.c_int]),
    ("lasr_cell_prof_kernel", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    ("lasr_trace", C.c_int, [_P, C.c_int]),
    ("lasr_trace_read", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    ("lasr_cell_prof_read", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    ("lasr_debug_config", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int)]),
    ("lasr_bench_front", C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, _P, C.POINTER(C.c_double), _P]),
]


class LmDesc(C.Structure):          # lasr_lm_desc
    _fields_ = [("vocab", C.c_int32), ("embed", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32),
                ("alpha", C.c_float), ("theta", C.c_float), ("min_val", C.c_float)]

_lib = None


def lib():
    """Load liblasr_hip.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  libreasr_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, res, args in SYMBOLS + DEBUG_SYMBOLS:
        f = getattr(L, name)       # AttributeError here == the .so does not export the header's symbol
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


class LasrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"liblasr_hip error {code}: {msg}")
        self.code = code
