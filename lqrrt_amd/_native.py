"""
ctypes binding of the C ABI in include/lqrrt_hip.h (liblqrrt_hip.so, built for gfx950 by
__graft_entry__.build()).  There is deliberately no fallback: if the shared library is
missing, or no HIP device is present, every compute call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LQRRT_LIB") or os.path.join(_HERE, "liblqrrt_hip.so")     # LQRRT_LIB: an experimental build (tools/)

MAX_STATES, MAX_CONTROLS, MAX_PARAMS = 12, 6, 96

MODEL_BOAT_ADVANCED, MODEL_BOAT_INTERMEDIATE, MODEL_BOAT_NOVICE = 1, 2, 3
MODEL_CAR, MODEL_PENDULUM, MODEL_DOUBLE_INTEGRATOR, MODEL_ROS_BOAT, MODEL_PENDULUM_LQR = 4, 5, 6, 7, 8
MODEL_BOAT_NOVICE_LQR = 9
MODEL_GENERIC = 200       # no plugins compiled in: node table + nearest-neighbour stage for host callables (lqrrt_amd/callback.py)
MODEL_USER = 100          # an out-of-tree problem compiled in (csrc/models.def)

E_ARG, E_HIP, E_NODEVICE, E_CAPACITY, E_STATE = -1, -2, -3, -4, -5
STOP_ATTEMPTS, STOP_NODES, STOP_TARGET, STOP_GOAL = 1, 2, 3, 4


class SystemDesc(C.Structure):
    _fields_ = [("model", C.c_int32), ("nstates", C.c_int32), ("ncontrols", C.c_int32),
                ("n_params", C.c_int32), ("params", C.c_double * MAX_PARAMS),
                ("n_vertices", C.c_int32), ("n_obstacles", C.c_int32), ("obs_stride", C.c_int32),
                ("reserved", C.c_int32), ("vps", C.POINTER(C.c_double)), ("obs", C.POINTER(C.c_double)),
                ("ogrid", C.POINTER(C.c_int8)), ("og_rows", C.c_int32), ("og_cols", C.c_int32),
                ("og_origin", C.c_double * 2), ("og_cpm", C.c_double), ("og_threshold", C.c_double)]


class Resolution(C.Structure):
    _fields_ = [("dt", C.c_double), ("FPR", C.c_double), ("horizon_iters", C.c_int32),
                ("has_goal", C.c_int32), ("error_tol", C.c_double * MAX_STATES),
                ("goal", C.c_double * MAX_STATES), ("goal_lo", C.c_double * MAX_STATES),
                ("goal_hi", C.c_double * MAX_STATES), ("adaptive", C.c_int32), ("hspan_min", C.c_int32),
                ("horizon_iters_state", C.c_int32), ("reserved", C.c_int32)]


class SamplerDesc(C.Structure):
    _fields_ = [("centers", C.c_double * MAX_STATES), ("spans", C.c_double * MAX_STATES),
                ("goal_bias", C.c_double * MAX_STATES), ("tries_limit", C.c_int32), ("reserved", C.c_int32)]


class ExtendStats(C.Structure):
    _fields_ = [("attempts", C.c_int64), ("accepted", C.c_int64), ("candidates", C.c_int64),
                ("waves", C.c_int64), ("fix_rounds", C.c_int64), ("resteers", C.c_int64),
                ("goal_hits", C.c_int64), ("speculated", C.c_int64), ("chain_slots", C.c_int64), ("tree_size", C.c_int32),
                ("stop_reason", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/lqrrt_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_I, _I64, _D = C.c_int, C.c_int64, C.c_double
SIGNATURES = {
    "lqrrt_last_error": (C.c_char_p, []),
    "lqrrt_switches_describe": (C.c_char_p, []),
    "lqrrt_abi_version": (_I, []),
    "lqrrt_device_count": (_I, []),
    "lqrrt_engine_create": (_I, [C.POINTER(SystemDesc), _I, _I, _I, C.POINTER(_P)]),
    "lqrrt_engine_destroy": (_I, [_P]),
    "lqrrt_engine_footprint": (_I, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "lqrrt_engine_set_geometry": (_I, [_P, C.POINTER(SystemDesc), _P]),
    "lqrrt_engine_set_wave_mode": (_I, [_P, _I]),
    "lqrrt_engine_set_cu_mask": (_I, [_P, _P, _I]),
    "lqrrt_engine_set_resolution": (_I, [_P, C.POINTER(Resolution)]),
    "lqrrt_engine_set_sampler": (_I, [_P, C.POINTER(SamplerDesc)]),
    "lqrrt_engine_horizon_iters": (_I, [_P]),
    "lqrrt_engine_set_dense_S": (_I, [_P, _P]),
    "lqrrt_engine_set_mt19937": (_I, [_P, _P, _I]),
    "lqrrt_engine_get_mt19937": (_I, [_P, _P, C.POINTER(_I)]),
    "lqrrt_tree_reset": (_I, [_P, _P, _P]),
    "lqrrt_tree_size": (_I, [_P]),
    "lqrrt_tree_load": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "lqrrt_tree_append": (_I, [_P, _I, _P, _P, _I, _P, _P, _P]),
    "lqrrt_tree_truncate": (_I, [_P, _I]),
    "lqrrt_tree_set_ignored": (_I, [_P, _I, _I, _P]),
    "lqrrt_tree_get_edges": (_I, [_P, _I, _I, _P, _P]),
    "lqrrt_tree_mark": (_I, [_P]),
    "lqrrt_tree_rewind": (_I, [_P]),
    "lqrrt_tree_set_rewind_above": (_I, [_P, _I]),
    "lqrrt_tree_get_states": (_I, [_P, _I, _I, _P]),
    "lqrrt_tree_get_gains": (_I, [_P, _I, _I, _P]),
    "lqrrt_tree_get_parents": (_I, [_P, _I, _I, _P]),
    "lqrrt_tree_get_edge_lengths": (_I, [_P, _I, _I, _P]),
    "lqrrt_tree_get_edge": (_I, [_P, _I, _P, _P]),
    "lqrrt_tree_get_ignored": (_I, [_P, _I, _I, _P]),
    "lqrrt_tree_climb": (_I, [_P, _I, _P, _I]),
    "lqrrt_tree_get_edges_of": (_I, [_P, _P, _I, _P, _P, _P]),
    "lqrrt_feasible_batch": (_I, [_P, _P, _P, _I, _P, _P]),
    "lqrrt_dynamics_batch": (_I, [_P, _P, _P, _I, _P, _P]),
    "lqrrt_gain_batch": (_I, [_P, _P, _P, _I, _P, _P]),
    "lqrrt_erf_batch": (_I, [_P, _P, _P, _I, _P, _P]),
    "lqrrt_lqr_dare_batch": (_I, [_P, _P, _P, _I, _P, _P, _D, _P, _P, _P, _P, _P, _P]),
    "lqrrt_nn_argmin": (_I, [_P, _P, _I, _P, _I, _P, _P, _P]),
    "lqrrt_nn_argmin_host": (_I, [_P, _P, _P, _I, C.POINTER(C.c_int32), C.POINTER(_D), _P]),
    "lqrrt_nn_argmin_errors": (_I, [_P, _P, _P, _I, C.POINTER(C.c_int32), C.POINTER(_D), _P]),
    "lqrrt_costs_to_go": (_I, [_P, _P, _P, _P, _P]),
    "lqrrt_steer_batch": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "lqrrt_steer_force": (_I, [_P, _I, _P, _I, _D, _D, _P, _P, _P, _P]),
    "lqrrt_engine_push_samples": (_I, [_P, _P, _I]),
    "lqrrt_engine_queued_samples": (_I, [_P]),
    "lqrrt_record_layout": (_I, [_P, _P]),
    "lqrrt_wave_records": (_I, [_P, C.POINTER(_P)]),
    "lqrrt_wave_suggest": (_I, [_P, _I]),
    "lqrrt_wave_speculate": (_I, [_P, _I, _I, _I, _P]),
    "lqrrt_wave_scan_nodes": (_I, [_P, _I, _I, _I, _P, _P]),
    "lqrrt_wave_steer_candidates": (_I, [_P, _I, _I, _P, _P]),
    "lqrrt_wave_commit": (_I, [_P, _I, _I64, _I64, _I, C.POINTER(ExtendStats), _P]),
    "lqrrt_engine_extend": (_I, [_P, _I, _I64, _I64, _I, _I, _I, C.POINTER(ExtendStats), _P]),
    "lqrrt_engine_extend_multi": (_I, [_P, _I, _I, _I64, _I64, _I, _I, _I, _P, _P]),
    "lqrrt_comm_unique_id": (_I, [_P]),
    "lqrrt_comm_create": (_I, [_P, _I, _I, _I, C.POINTER(_P)]),
    "lqrrt_comm_create_loopback": (_I, [_I, _I, C.POINTER(_P)]),
    "lqrrt_comm_destroy": (_I, [_P]),
    "lqrrt_allgather_nodes": (_I, [_P, _P, _I, _P]),
    "lqrrt_engine_extend_sharded": (_I, [_P, _P, _I, _I, _I64, _I64, _I, _I, _I, C.POINTER(ExtendStats), _P]),
    "lqrrt_clock_probe": (_I, [_I, C.POINTER(_D), C.POINTER(_D), C.POINTER(_D), _P]),
    "lqrrt_plan_best": (_I, [_P, C.POINTER(C.c_int32), C.POINTER(_I64), C.POINTER(_I64)]),
    "lqrrt_engine_counters": (_I, [_P, C.POINTER(ExtendStats)]),
    "lqrrt_profile_enable": (_I, [_P, _I]),
    "lqrrt_profile_read": (_I, [_P, C.POINTER(_D), C.POINTER(_I64), C.POINTER(_D), C.POINTER(_D), C.POINTER(_I64)]),
}

_lib = None


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "lqrrt_hip error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Loads liblqrrt_hip.so (once).  Raises if it has not been built: the library is never compiled implicitly
    (least of all inside a planning call) and there is no CPU path to fall back to."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "lqrrt_amd: %s is missing -- build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64,
        # and a second copy (the one under /opt/rocm that hipcc linked against) would fight it for the
        # device.  Importing torch first makes the loader resolve our NEEDED libamdhip64.so.7 to the copy
        # torch already mapped, so kernels, torch tensors and RCCL all share one runtime.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.lqrrt_abi_version() != 1:
            raise RuntimeError("lqrrt_amd: ABI version mismatch")
        _lib = L
    return _lib


def check(rc):
    if rc < 0:
        msg = lib().lqrrt_last_error()
        code = rc
        text = msg.decode() if msg else ""
        if code == E_ARG:
            raise ValueError(text)          # the reference raises ValueError on bad arguments
        raise NativeError(code, text)
    return rc


def device_count():
    return lib().lqrrt_device_count()


def available():
    """True when the HIP library is built and a device is visible (compute calls can succeed)."""
    try:
        return os.path.exists(LIB_PATH) and device_count() >= 1
    except (OSError, RuntimeError):
        return False


def require_device():
    if device_count() < 1:
        raise NativeError(E_NODEVICE, "no HIP device visible: lqrrt_amd computes only on an MI355X (no CPU fallback)")


def ptr(a):
    """Host pointer of a C-contiguous numpy array (kept alive by the caller)."""
    return a.ctypes.data_as(C.c_void_p)


def current_stream(device=None):
    """hipStream_t of torch's current stream ON `device` (the engine's device, not torch's current one) when torch is
    importable, else the null stream."""
    try:
        import torch
        if torch.cuda.is_available():
            return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    except Exception:
        pass
    return C.c_void_p(0)


def as_f64(a, shape=None):
    out = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and out.shape != tuple(shape):
        raise ValueError("expected array of shape %r, got %r" % (tuple(shape), out.shape))
    return out
