"""ctypes loader for ``libgraphite_mi355x.so`` (the C-ABI of include/graphite_mi355x.h).

There is no CPU fallback: if the library is missing, or no GPU is visible when a
compute entry point is called, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GR_LIB_PATH") or os.path.join(_HERE, "libgraphite_mi355x.so")  # GR_LIB_PATH: diagnostic builds (tools/)
CSRC = os.path.join(_HERE, "csrc")

GR_OK = 0
STATUS_NAMES = {0: "GR_OK", 1: "GR_ERR_INVALID", 2: "GR_ERR_HIP", 3: "GR_ERR_NO_DEVICE",
                4: "GR_ERR_DUPLICATE_EDGE", 5: "GR_ERR_SOLVE_FAILED", 6: "GR_ERR_COMM"}

# every symbol include/graphite_mi355x.h declares
EXPORTS = [
    "gr_version", "gr_last_error_string", "gr_device_count", "gr_warm_up",
    "gr_bal_create", "gr_bal_create_shard", "gr_bal_destroy", "gr_bal_set_loss", "gr_bal_set_scale_system", "gr_bal_set_jacobian_precision",
    "gr_bal_set_params", "gr_bal_get_params", "gr_bal_linearize", "gr_bal_chi2",
    "gr_bal_backup_parameters", "gr_bal_revert_parameters", "gr_bal_apply_update",
    "gr_bal_solver_update_structure", "gr_bal_solver_update_values", "gr_bal_solver_set_damping",
    "gr_bal_solver_solve", "gr_bal_schur_update_values", "gr_bal_schur_matvec",
    "gr_bal_landmark_update", "gr_bal_schur_structure", "gr_bal_get", "gr_bal_hessian_structure", "gr_bal_export_csc",
    "gr_bal_levenberg_marquardt", "gr_bal_kernel_stats", "gr_comm_unique_id", "gr_bal_comm_init", "gr_bal_comm_ipc_mailbox", "gr_bal_comm_set_contributors", "gr_bal_comm_init_ipc", "gr_bal_set_fixed",
    "gr_dense_cholesky_solve", "gr_bal_model_evaluate", "gr_bal_tuning_default", "gr_bal_set_tuning", "gr_bal_get_tuning",
    "gr_bal_direct_solver_info", "gr_bal_lm_iteration_seconds", "gr_bal_comm_info",
    "gr_spchol_create", "gr_spchol_factor_solve", "gr_spchol_info", "gr_spchol_destroy",
    "gr_bal_create_model", "gr_bal_model_orders",  # include/graphite_mi355x_model.h: the engine on user traits
]
# include/graphite_mi355x_test.h (test / diagnostic entry points, not part of the drop-in boundary)
TEST_EXPORTS = ["gr_bal_comm_init_local", "gr_bal_diag_time", "gr_bal_comm_allreduce_host", "gr_test_lane_xor"]


class GraphiteError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status


class CommInfo(C.Structure):
    """gr_comm_info (include/graphite_mi355x.h)"""
    _fields_ = [(k, C.c_int32) for k in ("rank", "size", "transport", "rccl_ranks", "mailboxes_opened", "device", "fused_agreed", "reserved")] + \
               [("oneshot_messages", C.c_int64), ("fallback_messages", C.c_int64)]


class LMOptions(C.Structure):
    _fields_ = [("solver", C.c_int32), ("iterations", C.c_int32), ("initial_damping", C.c_double),
                ("use_identity", C.c_int32), ("pcg_max_iter", C.c_int32), ("pcg_tol", C.c_double),
                ("pcg_rejection_ratio", C.c_double), ("profile", C.c_int32), ("early_stop", C.c_int32),
                ("stop_flag", C.c_void_p)]


class LMStats(C.Structure):
    _fields_ = [("iterations_run", C.c_int32), ("accepted", C.c_int32), ("pcg_iterations", C.c_int32),
                ("ok", C.c_int32), ("setup_seconds", C.c_double), ("loop_seconds", C.c_double),
                ("solve_seconds", C.c_double), ("final_chi2", C.c_double), ("collectives", C.c_int64), ("kernel_launches", C.c_int64), ("fused_messages", C.c_int64)]


class Tuning(C.Structure):
    """gr_bal_tuning (include/graphite_mi355x.h)"""
    _fields_ = [(k, C.c_int32) for k in ("point_tiles", "g3_gather", "point_records", "pcg_lazy", "pcg_single_reduction", "sparse_cholesky",
                                         "spchol_overlap", "lm_speculate", "lm_ahead", "lm_fused", "grid_mult", "vec_per_thread", "schur_item",
                                         "verbose", "ipc_timeout_ms", "shard_fused", "shard_virtual_ranks", "chol_fuse", "chol_pin", "spchol_fuse", "spchol_slice", "schur_fused", "spchol_bwd_chain", "pcg_resident", "comm_transport")]


class DirectSolverInfo(C.Structure):
    """gr_direct_solver_info (include/graphite_mi355x.h)"""
    _fields_ = [("sparse", C.c_int32), ("tile_columns", C.c_int32), ("levels", C.c_int32), ("supernodes", C.c_int32),
                ("padded_n", C.c_int64), ("factor_tiles", C.c_int64), ("factor_bytes", C.c_int64), ("dense_bytes", C.c_int64)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("bytes_per_launch", C.c_double), ("flops_per_launch", C.c_double), ("active_launches", C.c_int64)]


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "graphite_mi355x.h"))
    stale = (not os.path.exists(LIB_PATH)
             or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs))
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-B"])
    return LIB_PATH


_lib = None


def lib():
    """Load the shared library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GraphiteError(-1, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(there is no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        _lib.gr_version.restype = C.c_char_p
        _lib.gr_last_error_string.restype = C.c_char_p
    return _lib


def check(status):
    if status != GR_OK:
        raise GraphiteError(status, lib().gr_last_error_string().decode())
