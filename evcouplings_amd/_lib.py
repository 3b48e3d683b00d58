"""
ctypes binding of libplm_hip.so (include/plm_hip.h).

There is deliberately no CPU fallback: if the shared library is missing, or no gfx950
device is visible, every call raises.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C evcouplings_amd/csrc`` so it travels with the repo.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLM_HIP_LIB selects another build of the same library (kernel A/B experiments only)
LIB_PATH = os.environ.get("PLM_HIP_LIB") or os.path.join(_HERE, "libplm_hip.so")

PLM_OK = 0
STATUS_CONVERGED, STATUS_MAXITER, STATUS_LINESEARCH, STATUS_INTERRUPTED = 0, 1, 2, 3
ABI_VERSION = 2
K_EXPAND, K_FORWARD, K_BACKWARD, K_ASSEMBLE, K_TOTAL, K_REWEIGHT, K_FIELDS, K_FORWARD_ACCURATE, K_LBFGS_VECTOR, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
S_COUNT = 7

ITER_CB = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                      C.c_double, C.c_double, C.c_void_p)
EXCHANGE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_void_p)
COLLECTIVE_CB = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64),
                            C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_void_p)
COLL_ALLTOALL, COLL_ALLREDUCE_F64, COLL_ALLREDUCE_F32, COLL_BROADCAST = 1, 2, 3, 4


class PlmProblem(C.Structure):
    _fields_ = [
        ("n_seqs", C.c_int32), ("n_sites", C.c_int32), ("n_states", C.c_int32),
        ("msa", C.c_void_p),
        ("theta_id", C.c_double), ("scale", C.c_double),
        ("lambda_h", C.c_double), ("lambda_j", C.c_double),
        ("max_iter", C.c_int32), ("epsilon", C.c_double), ("lbfgs_m", C.c_int32),
        ("n_shards", C.c_int32), ("shard", C.c_int32), ("flags", C.c_int32),
        ("lambda_group", C.c_double),
    ]


class PlmResult(C.Structure):
    _fields_ = [
        ("weights", C.c_void_p), ("fi", C.c_void_p), ("fij", C.c_void_p), ("hi", C.c_void_p),
        ("jij", C.c_void_p), ("fn", C.c_void_p), ("cn", C.c_void_p),
        ("n_eff", C.c_float), ("iters_done", C.c_int32), ("n_evals", C.c_int32),
        ("status", C.c_int32), ("fx", C.c_double),
        ("seconds_reweight", C.c_double), ("seconds_marginals", C.c_double),
        ("seconds_optimize", C.c_double), ("seconds_total", C.c_double),
        ("status_msg", C.c_char * 128),
    ]


class PlmMfResult(C.Structure):
    _fields_ = [
        ("weights", C.c_void_p), ("n_eff", C.c_float), ("fi", C.c_void_p), ("fij", C.c_void_p),
        ("hi", C.c_void_p), ("jij_full", C.c_void_p), ("jij", C.c_void_p), ("di", C.c_void_p),
    ]


# every symbol include/plm_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("plm_version", C.c_int, []),
    ("plm_device_count", C.c_int, []),
    ("plm_strerror", C.c_char_p, [C.c_int]),
    ("plm_last_error", C.c_char_p, []),
    ("plm_fit", C.c_int, [C.POINTER(PlmProblem), C.POINTER(PlmResult), C.c_int, _P, ITER_CB, _P,
                          EXCHANGE_CB, _P]),
    ("plm_fit_sharded", C.c_int, [C.POINTER(PlmProblem), C.POINTER(PlmResult), C.c_int, _P, ITER_CB, _P,
                                  COLLECTIVE_CB, _P]),
    ("plm_fit_sharded_rccl", C.c_int, [C.POINTER(PlmProblem), C.POINTER(PlmResult), C.c_int, _P, ITER_CB, _P, _P]),
    ("plm_rccl_unique_id", C.c_int, [_P]),
    ("plm_rccl_runtime_version", C.c_int, []),
    ("plm_rccl_selftest", C.c_int, [C.c_int, _P]),
    ("plm_ctx_attach_rccl", C.c_int, [_P, _P]),
    ("plm_reweight", C.c_int, [_P, C.c_int32, C.c_int32, C.c_double, _P]),
    ("plm_reweight_ex", C.c_int, [_P, C.c_int32, C.c_int32, C.c_double, C.c_int32, _P]),
    ("plm_marginals", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("plm_eval", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, _P,
                           C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    ("plm_scores", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    ("plm_scores_ex", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("plm_hamiltonians", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int, _P, _P]),
    ("plm_potentials", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int, _P, _P]),
    ("plm_meanfield", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int, _P,
                                C.POINTER(PlmMfResult)]),
    ("plm_direct_information", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int, _P, _P]),
    ("plm_alignment_stats", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int, _P]),
    ("plm_fasta_split", C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P, _P, _P, _P]),
    ("plm_encode_columns", C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, _P, _P, _P]),
    ("plm_write_raw_ec_file", C.c_int, [C.c_char_p, C.c_int32, _P, C.c_char_p, _P]),
    ("plm_ctx_create", C.c_int, [C.POINTER(PlmProblem), C.c_int, _P, C.POINTER(_P)]),
    ("plm_ctx_destroy", None, [_P]),
    ("plm_ctx_set_exchange", C.c_int, [_P, EXCHANGE_CB, _P]),
    ("plm_ctx_set_collective", C.c_int, [_P, COLLECTIVE_CB, _P]),
    ("plm_ctx_set_options", C.c_int, [_P, C.c_int32, C.c_double, C.c_int32]),
    ("plm_ctx_native_size", C.c_int64, [_P]),
    ("plm_ctx_reweight", C.c_int, [_P]),
    ("plm_ctx_set_weights", C.c_int, [_P, _P]),
    ("plm_ctx_get_weights", C.c_int, [_P, _P, _P, C.POINTER(C.c_float)]),
    ("plm_ctx_marginals", C.c_int, [_P, _P, _P]),
    ("plm_ctx_set_x", C.c_int, [_P, _P]),
    ("plm_ctx_get_x", C.c_int, [_P, _P]),
    ("plm_ctx_get_g", C.c_int, [_P, _P]),
    ("plm_ctx_eval", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("plm_ctx_optimize", C.c_int, [_P, ITER_CB, _P, C.POINTER(PlmResult)]),
    ("plm_ctx_scores", C.c_int, [_P, _P, _P]),
    ("plm_ctx_time_kernels", C.c_int, [_P, C.c_int32, _P]),
    ("plm_ctx_time_field_positions", C.c_int, [_P, C.c_int32, _P]),
    ("plm_ctx_solver_stats", C.c_int, [_P, _P]),
    ("plm_rccl_probe", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int, _P]),
    ("plm_rccl_probe_local", C.c_int, [C.c_int32, C.c_int, _P]),
    ("plm_lbfgs_coefficients", None, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_double, _P, _P,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]),
]

_lib = None


class PlmError(RuntimeError):
    """Failure reported by libplm_hip (code + the library's message)."""

    def __init__(self, code, message):
        super().__init__("libplm_hip error %d: %s" % (code, message))
        self.code = code


def load():
    """Load libplm_hip.so and bind every declared symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the PLM solver)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        if os.environ.get("PLM_HIP_LIB") and not hasattr(lib, name):
            continue              # A/B experiments against an older build: tolerate newer symbols
        fn = getattr(lib, name)   # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.plm_version() != ABI_VERSION:
        raise ImportError("libplm_hip ABI version %d, expected %d" % (lib.plm_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != PLM_OK:
        lib = load()
        msg = lib.plm_last_error() or lib.plm_strerror(rc) or b""
        raise PlmError(rc, msg.decode("utf-8", "replace"))
