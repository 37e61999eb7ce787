"""ctypes binding of libhipensemble.so (include/hipensemble.h).

There is no CPU fallback: if the shared library is missing, or no MI355X is
visible when a context is created, the product path raises.

Process-level note: PyTorch-ROCm bundles its own HIP runtime.  A process that uses both torch
and this library must import torch FIRST (bench.py, eryn_amd.ladder do); initialising the system
HIP runtime through this library and importing torch afterwards leaves torch without devices.
"""
import ctypes as C
import os

import numpy as np

from ._build import LIB_PATH

HENS_OK = 0
ERR_INVALID, ERR_HIP, ERR_STATE, ERR_TOO_FEW_WALKERS, ERR_NONFINITE, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
LIKE_GAUSS_DENSE, LIKE_GAUSS_DIAG, LIKE_ROSENBROCK, LIKE_HOST, LIKE_TEMPLATE = 0, 1, 2, 3, 4


class HensConfig(C.Structure):
    _fields_ = [
        ("ntemps", C.c_int32), ("nwalkers", C.c_int32), ("ndim", C.c_int32),
        ("rung_begin", C.c_int32), ("rung_end", C.c_int32), ("device_id", C.c_int32),
        ("likelihood_kind", C.c_int32), ("tempered", C.c_int32), ("live_dangerously", C.c_int32),
        ("adaptive", C.c_int32), ("adaptation_delay", C.c_int32), ("ndim_active", C.c_int32),
        ("stop_adaptation", C.c_int64), ("a", C.c_double), ("fill_value", C.c_double),
        ("adaptation_lag", C.c_double), ("adaptation_time", C.c_double), ("seed", C.c_uint64),
    ]


class HensTiming(C.Structure):
    _fields_ = [
        ("total_ms", C.c_double), ("stretch_ms", C.c_double), ("pt_ms", C.c_double), ("plan_ms", C.c_double),
        ("n_stretch", C.c_int64), ("n_pt", C.c_int64), ("n_plan", C.c_int64), ("n_iters", C.c_int64),
        ("fused_ms", C.c_double), ("n_fused", C.c_int64), ("clock", C.c_int64),
    ]


RJ_MOVE_MH, RJ_MOVE_BD, RJ_MOVE_BD_ALL, RJ_MOVE_STRETCH = 0, 1, 2, 3      # include/hipensemble.h: HENS_RJ_MOVE_*


class HensRjDraws(C.Structure):
    """struct hens_rj_draws (include/hipensemble.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("step", "change", "leaf", "birth", "labels", "rint", "u_zz", "u_acc")] + \
               [("branch", C.c_int32), ("split", C.c_int32)]


class HensPipeRegions(C.Structure):
    """struct hens_pipe_region_table (include/hipensemble.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("ldn_out", "ldn_rows_out", "ldn_in", "ldn_rows_in", "lup_out", "lup_in",
                                          "rows_out", "rows_in", "cnt_out", "cnt_in")] + \
               [("lp_doubles", C.c_int64), ("row_doubles", C.c_int64), ("cnt_words", C.c_int64), ("stream", C.c_void_p)]


class HensDeviceBuffers(C.Structure):
    _fields_ = [
        ("logl", C.c_void_p), ("gather_logl", C.c_void_p), ("send_rows", C.c_void_p), ("recv_rows", C.c_void_p),
        ("row_capacity", C.c_int64), ("row_doubles", C.c_int64), ("stream", C.c_void_p),
    ]


_P = C.c_void_p
# every symbol include/hipensemble.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "hens_create": (C.c_int, [C.POINTER(HensConfig), C.POINTER(_P)]),
    "hens_destroy": (None, [_P]),
    "hens_last_error": (C.c_char_p, [_P]),
    "hens_synchronize": (C.c_int, [_P]),
    "hens_set_prior_box": (C.c_int, [_P, _P, _P, C.c_double]),
    "hens_set_periodic": (C.c_int, [_P, _P]),
    "hens_set_gaussian": (C.c_int, [_P, _P, _P]),
    "hens_set_rosenbrock": (C.c_int, [_P, C.c_double, C.c_double]),
    "hens_upload_state": (C.c_int, [_P, _P, _P, _P, _P]),
    "hens_download_state": (C.c_int, [_P, _P, _P, _P, _P]),
    "hens_eval_state": (C.c_int, [_P]),
    "hens_stretch_split": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "hens_propose_split": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "hens_accept_split": (C.c_int, [_P, C.c_int32, _P, _P, _P]),
    "hens_pt_sweep": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P, _P]),
    "hens_step": (C.c_int, [_P, C.c_int64]),
    "hens_get_counters": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "hens_reset_counters": (C.c_int, [_P]),
    "hens_set_adapt_time": (C.c_int, [_P, C.c_int64]),
    "hens_set_profiling": (C.c_int, [_P, C.c_int32]),
    "hens_get_timing": (C.c_int, [_P, C.POINTER(HensTiming)]),
    "hens_get_device_buffers": (C.c_int, [_P, C.POINTER(HensDeviceBuffers)]),
    "hens_set_stream": (C.c_int, [_P, _P]),
    "hens_stretch_iter": (C.c_int, [_P]),
    "hens_pt_plan_sharded": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "hens_pt_finish_sharded": (C.c_int, [_P, C.c_int64]),
    "hens_mh_step": (C.c_int, [_P, _P, _P, _P]),
    "hens_set_mh_proposal": (C.c_int, [_P, C.c_int32, _P, C.c_double]),
    "hens_get_mh_counters": (C.c_int, [_P, _P, _P]),
    "hens_step_marked": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "hens_get_marked_counters": (C.c_int, [_P, _P, _P]),
    "hens_step_report": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P]),
    "hens_pipe_init": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    "hens_pipe_connect": (C.c_int, [_P, _P]),
    "hens_pipe_connect_local": (C.c_int, [_P, _P]),
    "hens_pipe_connect_staged": (C.c_int, [_P]),
    "hens_pipe_regions": (C.c_int, [_P, _P]),
    "hens_pipe_stage": (C.c_int, [_P, C.c_int32]),
    "hens_pipe_selftest": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_double]),
    "hens_pipe_debug_stats": (C.c_int, [_P, _P, C.c_int32]),
    "hens_debug_trace": (C.c_int, [_P, C.c_int32, _P, C.c_int64, _P]),
    "hens_debug_launch_times": (C.c_int, [_P, _P, C.c_int64, _P]),
    "hens_comm_unique_id": (C.c_int, [_P]),
    "hens_comm_init": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "hens_comm_destroy": (C.c_int, [_P]),
    "hens_comm_selfsend": (C.c_int, [_P, C.c_int64, _P, _P]),
    "hens_debug_permutation": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int64, _P]),
    "hens_rj_set_model": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P, C.c_double]),
    "hens_rj_set_model_general": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "hens_rj_set_mh_scale": (C.c_int, [_P, _P]),
    "hens_rj_mh_step": (C.c_int, [_P, _P, _P, _P]),
    "hens_rj_bd_step": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "hens_rj_step": (C.c_int, [_P, C.c_int64]),
    "hens_rj_get_counters": (C.c_int, [_P, _P, _P, _P]),
    "hens_rj_debug_draws": (C.c_int, [_P, C.c_int64] + [_P] * 11),
    "hens_rj_set_schedule": (C.c_int, [_P, C.c_int32]),
    "hens_rj_bd_all_step": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "hens_rj_propose": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P]),
    "hens_rj_accept": (C.c_int, [_P, _P, _P]),
    "hens_rj_stretch_split": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "hens_get_iteration": (C.c_int, [_P, _P]),
    "hens_set_iteration": (C.c_int, [_P, C.c_int64]),
    "hens_set_nsplits": (C.c_int, [_P, C.c_int32]),
    "hens_set_stretch_scale": (C.c_int, [_P, C.c_double]),
    "hens_debug_draws": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hens_version": (C.c_char_p, []),
    "hens_device_count": (C.c_int, []),
    "hens_device_pci_bus_id": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32]),
}

_lib = None


class HipExtensionMissing(ImportError):
    pass


def load():
    """Load the shared library (no GPU needed to load and resolve symbols)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: build it with `python -m eryn_amd._build` "
            "(hipcc --offload-arch=gfx950); eryn_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    other_build = bool(os.environ.get("HENS_LIB"))   # a same-box A/B against another revision's build (tools/mkbase.sh)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
        except AttributeError:
            if other_build:           # (an older build may lack entry points added since; calling one still fails loudly)
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


_EXC = {ERR_INVALID: ValueError, ERR_HIP: RuntimeError, ERR_STATE: RuntimeError,
        ERR_TOO_FEW_WALKERS: RuntimeError, ERR_NONFINITE: ValueError, ERR_UNSUPPORTED: NotImplementedError}


def check(code, ctx=None):
    """Map a C status to the exception type the reference raises for the same condition."""
    if code == HENS_OK:
        return
    msg = load().hens_last_error(ctx)
    raise _EXC.get(code, RuntimeError)(msg.decode() if msg else f"hens error {code}")
