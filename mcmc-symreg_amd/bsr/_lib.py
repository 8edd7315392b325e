"""ctypes binding of libbsr_hip.so (C ABI: include/bsr_hip.h).  No CPU fallback: if the library or a GPU is
missing, every entry point raises."""
import ctypes as C
import os

import numpy as np

MAX_K = 8
MAX_INFLIGHT = 8          # include/bsr_hip.h: BSR_MAX_INFLIGHT (batch slots of a context)
COMM_ID_BYTES = 128
F_INF, F_NAN, F_RANKDEF, F_SCALE_RETRY = 1, 2, 4, 8
DTYPE_F64, DTYPE_F32 = 0, 1

ERRORS = {-1: "BSR_E_ARG", -2: "BSR_E_HIP", -3: "BSR_E_NODEVICE", -4: "BSR_E_TOOBIG", -5: "BSR_E_TAPE",
          -6: "BSR_E_STATE", -7: "BSR_E_COMM", -8: "BSR_E_LINALG"}


class BsrError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "BSR_E_?"), code, text))
        self.code = code


class Score(C.Structure):
    _fields_ = [("loglik", C.c_double), ("sse", C.c_double), ("scale", C.c_double), ("maxabs", C.c_double),
                ("smin", C.c_double), ("smax", C.c_double), ("beta", C.c_double * MAX_K),
                ("rank", C.c_int32), ("flags", C.c_uint32)]


SCORE_DTYPE = np.dtype([("loglik", "<f8"), ("sse", "<f8"), ("scale", "<f8"), ("maxabs", "<f8"), ("smin", "<f8"),
                        ("smax", "<f8"), ("beta", "<f8", (MAX_K,)), ("rank", "<i4"), ("flags", "<u4")], align=True)
assert SCORE_DTYPE.itemsize == C.sizeof(Score)


class ChainInfo(C.Structure):
    _fields_ = [("sse_old", C.c_double), ("scale_old", C.c_double), ("maxabs", C.c_double * MAX_K),
                ("beta_old", C.c_double * MAX_K), ("colflags", C.c_uint32 * MAX_K), ("rank_old", C.c_int32),
                ("pad", C.c_int32)]


_LIB = None
# BSR_LIB_PATH: an alternative build of the same library (kernel experiments, e.g. tools/ablate_tile.sh)
LIB_PATH = os.environ.get("BSR_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbsr_hip.so")


def lib():
    """Loads the shared library once.  Raises (never falls back) when it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("libbsr_hip.so not built: run mcmc-symreg_amd/csrc/build.sh (or __graft_entry__.build())")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pd, pi, pu = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    sigs = {
        "bsr_abi_version": (C.c_int, []),
        "bsr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "bsr_ctx_create": (C.c_int, [C.POINTER(vp), C.c_int, i64, i32, vp, vp, i32, i32, i32, i32]),
        "bsr_ctx_create_tuned": (C.c_int, [C.POINTER(vp), C.c_int, i64, i32, vp, vp, i32, i32, i32, i32, i32, i32]),
        "bsr_ctx_destroy": (C.c_int, [vp]),
        "bsr_last_error": (C.c_char_p, [vp]),
        "bsr_eval_tapes": (C.c_int, [vp, vp, vp, i32, vp, vp, vp]),
        "bsr_set_current": (C.c_int, [vp, i32, i32, vp, i32]),
        "bsr_commit": (C.c_int, [vp, i32, i32, i32]),
        "bsr_refresh": (C.c_int, [vp, i32, C.POINTER(ChainInfo)]),
        "bsr_score_batch": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp]),
        "bsr_score_submit": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, pi]),
        "bsr_score_wait": (C.c_int, [vp, i32, vp]),
        "bsr_score_submit_mh": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, pi]),
        "bsr_score_wait_mh": (C.c_int, [vp, i32, vp, vp]),
        "bsr_fit_beta": (C.c_int, [vp, i32, vp, pd]),
        "bsr_get_current": (C.c_int, [vp, i32, vp]),
        "bsr_yloglike_host": (C.c_int, [C.c_int, i64, i32, vp, vp, dbl, i32, pd, pd, pd, vp, pi]),
        "bsr_set_profiling": (C.c_int, [vp, i32]),
        "bsr_last_timing": (C.c_int, [vp, vp]),
        "bsr_ctx_info": (C.c_int, [vp, vp]),
        "bsr_batch_stats": (C.c_int, [vp, C.c_int32, vp]),
        "bsr_dispatch_info": (C.c_int, [vp, vp]),
        "bsr_place_info": (C.c_int, [vp]),
        "bsr_comm_unique_id": (C.c_int, [vp]),
        "bsr_comm_init": (C.c_int, [vp, i32, i32, vp]),
        "bsr_comm_allgather": (C.c_int, [vp, vp, vp, i64]),
        "bsr_comm_destroy": (C.c_int, [vp]),
        "bsr_engine_create": (C.c_int, [C.POINTER(vp), vp, i32, i32, i64, i32, dbl, i32, i32]),
        "bsr_engine_destroy": (C.c_int, [vp]),
        "bsr_engine_last_error": (C.c_char_p, [vp]),
        "bsr_engine_set_nan_policy": (C.c_int, [vp, i32]),
        "bsr_engine_set_ops": (C.c_int, [vp, i32, vp, vp]),
        "bsr_engine_seed": (C.c_int, [vp, i32, C.c_uint32]),
        "bsr_engine_set_rng": (C.c_int, [vp, i32, vp, i32, i32, dbl]),
        "bsr_engine_get_rng": (C.c_int, [vp, i32, vp, pi, pi, pd]),
        "bsr_engine_init_chain": (C.c_int, [vp, i32]),
        "bsr_engine_run": (C.c_int, [vp, i32, i64, vp, i64, C.POINTER(i64), i32]),
        "bsr_engine_chain_result": (C.c_int, [vp, i32, vp, i32, vp, vp, vp, i32, pi, vp, pd, i32]),
        "bsr_engine_memo_stats": (C.c_int, [vp, i32, vp]),
        "bsr_rng_selftest": (C.c_int, [C.c_uint32, i32, vp, vp, vp, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.bsr_abi_version() != 1:
        raise ImportError("libbsr_hip.so ABI version mismatch")
    _LIB = L
    return L


EXPORTS = ["bsr_abi_version", "bsr_device_count", "bsr_ctx_create", "bsr_ctx_create_tuned", "bsr_ctx_destroy", "bsr_last_error",
           "bsr_eval_tapes", "bsr_set_current", "bsr_commit", "bsr_refresh", "bsr_score_batch", "bsr_score_submit",
           "bsr_score_wait", "bsr_score_submit_mh", "bsr_score_wait_mh", "bsr_fit_beta",
           "bsr_get_current", "bsr_yloglike_host", "bsr_set_profiling", "bsr_last_timing", "bsr_ctx_info", "bsr_batch_stats", "bsr_debug_tile_stamps", "bsr_debug_tile_stamp_ring", "bsr_dispatch_info", "bsr_place_info", "bsr_comm_unique_id",
           "bsr_comm_init", "bsr_comm_allgather", "bsr_comm_destroy", "bsr_engine_create", "bsr_engine_destroy",
           "bsr_engine_last_error", "bsr_engine_set_nan_policy", "bsr_engine_set_ops", "bsr_engine_seed", "bsr_engine_set_rng", "bsr_engine_get_rng",
           "bsr_engine_init_chain", "bsr_engine_run", "bsr_engine_chain_result", "bsr_engine_memo_stats", "bsr_rng_selftest"]

EVENT_DTYPE = np.dtype([("index", "<i4"), ("kind", "<i4"), ("logR", "<f8")], align=True)
MH_JUMP, MH_NO_UNIFORM = 1, 2
EV_NONE, EV_ACCEPT, EV_GATE, EV_GATE_PASSED = 0, 1, 2, 3

TRACE_DTYPE = np.dtype([("chain", "<i4"), ("count", "<i4"), ("action", "<i4"), ("change", "<i4"), ("rank", "<i4"),
                        ("accepted", "<i4"), ("n_nodes", "<i4"), ("pad", "<i4"), ("Q", "<f8"), ("Qinv", "<f8"),
                        ("new_sigma", "<f8"), ("new_sa2", "<f8"), ("new_sb2", "<f8"), ("yllstar", "<f8"),
                        ("yll", "<f8"), ("logR", "<f8"), ("u", "<f8"), ("rmse", "<f8"), ("tree_hash", "<u8")],
                       align=True)
ACTIONS = ["stay", "grow", "prune", "detransform", "transform", "ReassignOperator", "ReassignFeature"]
CHANGES = ["", "shrinkage", "expansion"]


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().bsr_last_error(ctx)
        raise BsrError(rc, msg.decode() if msg else "")


def device_count():
    n = C.c_int(0)
    rc = lib().bsr_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def ptr(a):
    """Address of a numpy array for a c_void_p argument (a plain int: half the cost of ctypes.data_as)."""
    return a.ctypes.data
