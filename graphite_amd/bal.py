"""Python harness over the C-ABI for BAL problems (tests, bench, smoke).

Mirrors the reference's call sequence for this path — Graph / Solver /
levenberg_marquardt (graph.hpp, solver/solver.hpp,
optimizer/levenberg_marquardt.hpp) — with the same method names, so the parity
tests read like the reference's own tests.  All compute happens inside
``libgraphite_mi355x.so``; numpy arrays only cross the boundary as host
pointers, torch CUDA tensors as device pointers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import LMOptions, LMStats, KernelStat, Tuning, check

SOLVER_PCG_SCHUR, SOLVER_PCG, SOLVER_PCG_IDENTITY, SOLVER_PCG_SCHUR_IMPLICIT, SOLVER_DENSE_SCHUR = 0, 1, 2, 3, 4
LOSS_DEFAULT, LOSS_HUBER = 0, 1
F32, F64 = 0, 1

_GET = dict(scales=0, b=1, Hcc=2, Hcp=3, Hll=4, S=5, b_schur=6, Hll_inv=7, residuals=8, H=9)


def _ptr(a):
    """host numpy array or torch tensor (host or cuda) -> void*"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())  # torch tensor


class BalProblem:
    """Device-resident BAL problem (gr_bal_problem handle)."""

    def __init__(self, cameras, points, obs, cam_idx, pt_idx, dtype=np.float64, device=0, stream=None, shard=False):
        self.lib = _lib.lib()
        self.dt = np.dtype(dtype)
        self.code = F32 if self.dt == np.float32 else F64
        cameras = np.ascontiguousarray(cameras, dtype=self.dt).reshape(-1, 9)
        points = np.ascontiguousarray(points, dtype=self.dt).reshape(-1, 3)
        obs = np.ascontiguousarray(obs, dtype=self.dt).reshape(-1, 2)
        ci = np.ascontiguousarray(cam_idx, dtype=np.int32)
        pi = np.ascontiguousarray(pt_idx, dtype=np.int32)
        self.Nc, self.Np, self.No = len(cameras), len(points), len(obs)
        self.n = 9 * self.Nc + 3 * self.Np
        self.h = C.c_void_p()
        create = self.lib.gr_bal_create_shard if shard else self.lib.gr_bal_create
        check(create(C.byref(self.h), C.c_int(self.code), C.c_int64(self.Nc), C.c_int64(self.Np),
                                     C.c_int64(self.No), _ptr(cameras), _ptr(points), _ptr(obs), _ptr(ci),
                                     _ptr(pi), C.c_int(device), C.c_void_p(stream or 0)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.gr_bal_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_tuning(self):
        t = Tuning()
        check(self.lib.gr_bal_get_tuning(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Tuning._fields_ if k != "reserved"}

    def set_tuning(self, **kw):
        """gr_bal_set_tuning: e.g. set_tuning(point_tiles=8, pcg_lazy=0); unnamed fields keep their value"""
        t = Tuning()
        check(self.lib.gr_bal_get_tuning(self.h, C.byref(t)))
        for k, v in kw.items():
            if k not in dict(Tuning._fields_) or k == "reserved":
                raise KeyError(k)
            setattr(t, k, int(v))
        check(self.lib.gr_bal_set_tuning(self.h, C.byref(t)))

    # --- Graph -------------------------------------------------------------------------
    def set_loss(self, kind, delta=0.0):
        check(self.lib.gr_bal_set_loss(self.h, C.c_int(kind), C.c_double(delta)))

    def scale_system(self, on):
        check(self.lib.gr_bal_set_scale_system(self.h, C.c_int(int(on))))

    def set_jacobian_precision(self, dtype):
        """np.float32 on an fp64 problem: Jacobian entries in fp32 (the reference's FP64-FP32 mode)."""
        check(self.lib.gr_bal_set_jacobian_precision(self.h, C.c_int(F64 if np.dtype(dtype) == np.float64 else F32)))

    def set_fixed(self, cam_fixed=None, pt_fixed=None):
        """VertexDescriptor::set_fixed (vertex.hpp:262): boolean masks over cameras / points (caller's order)"""
        cf = None if cam_fixed is None else np.ascontiguousarray(np.asarray(cam_fixed) != 0, dtype=np.uint8)
        pf = None if pt_fixed is None else np.ascontiguousarray(np.asarray(pt_fixed) != 0, dtype=np.uint8)
        check(self.lib.gr_bal_set_fixed(self.h, None if cf is None else cf.ctypes.data_as(C.c_void_p),
                                        None if pf is None else pf.ctypes.data_as(C.c_void_p)))

    def set_params(self, cameras, points):
        c = np.ascontiguousarray(cameras, dtype=self.dt)
        p = np.ascontiguousarray(points, dtype=self.dt)
        check(self.lib.gr_bal_set_params(self.h, _ptr(c), _ptr(p)))

    def get_params(self):
        c = np.zeros((self.Nc, 9), self.dt)
        p = np.zeros((self.Np, 3), self.dt)
        check(self.lib.gr_bal_get_params(self.h, _ptr(c), _ptr(p)))
        return c, p

    def linearize(self):
        check(self.lib.gr_bal_linearize(self.h))

    def chi2(self):
        v = C.c_double()
        check(self.lib.gr_bal_chi2(self.h, C.byref(v)))
        return v.value

    def backup_parameters(self):
        check(self.lib.gr_bal_backup_parameters(self.h))

    def revert_parameters(self):
        check(self.lib.gr_bal_revert_parameters(self.h))

    def apply_update(self, dx):
        dx = np.ascontiguousarray(dx, dtype=self.dt)
        check(self.lib.gr_bal_apply_update(self.h, _ptr(dx)))

    # --- Solver --------------------------------------------------------------------------
    def solver_update_structure(self, solver):
        check(self.lib.gr_bal_solver_update_structure(self.h, C.c_int(solver)))

    def solver_update_values(self, solver):
        check(self.lib.gr_bal_solver_update_values(self.h, C.c_int(solver)))

    def solver_set_damping(self, solver, mu, use_identity=False):
        check(self.lib.gr_bal_solver_set_damping(self.h, C.c_int(solver), C.c_double(mu), C.c_int(int(use_identity))))

    def solver_solve(self, solver, max_iter=10, tol=1.0, rej=5.0):
        dx = np.zeros(self.n, self.dt)
        it = C.c_int()
        check(self.lib.gr_bal_solver_solve(self.h, C.c_int(solver), C.c_int(max_iter), C.c_double(tol),
                                           C.c_double(rej), _ptr(dx), C.byref(it)))
        return dx, it.value

    # --- Schur ---------------------------------------------------------------------------
    def schur_update_values(self):
        check(self.lib.gr_bal_schur_update_values(self.h))

    def schur_matvec(self, x):
        x = np.ascontiguousarray(x, dtype=self.dt)
        y = np.zeros(9 * self.Nc, self.dt)
        check(self.lib.gr_bal_schur_matvec(self.h, _ptr(x), _ptr(y)))
        return y

    def landmark_update(self, xp):
        xp = np.ascontiguousarray(xp, dtype=self.dt)
        xl = np.zeros(3 * self.Np, self.dt)
        check(self.lib.gr_bal_landmark_update(self.h, _ptr(xp), _ptr(xl)))
        return xl

    def schur_structure(self):
        nb = C.c_int64()
        check(self.lib.gr_bal_schur_structure(self.h, C.byref(nb), None, None))
        colptr = np.zeros(self.Nc + 1, np.int64)
        rowidx = np.zeros(nb.value, np.int64)
        check(self.lib.gr_bal_schur_structure(self.h, C.byref(nb), _ptr(colptr), _ptr(rowidx)))
        return colptr, rowidx

    def get(self, name):
        cnt = C.c_int64()
        check(self.lib.gr_bal_get(self.h, C.c_int(_GET[name]), None, C.byref(cnt)))
        out = np.zeros(cnt.value, self.dt)
        check(self.lib.gr_bal_get(self.h, C.c_int(_GET[name]), _ptr(out), C.byref(cnt)))
        return out

    def hessian_structure(self):
        """block-CSC of the upper Hessian in the reference's layout: (colptr, rowidx, value offsets)"""
        nb = C.c_int64()
        check(self.lib.gr_bal_hessian_structure(self.h, C.byref(nb), None, None, None))
        colptr = np.zeros(self.Nc + self.Np + 1, np.int64)
        rowidx = np.zeros(nb.value, np.int64)
        offsets = np.zeros(nb.value, np.int64)
        check(self.lib.gr_bal_hessian_structure(self.h, C.byref(nb), _ptr(colptr), _ptr(rowidx), _ptr(offsets)))
        return colptr, rowidx, offsets

    def export_csc(self, which):
        """which: 'H' or 'S' -> (indptr, indices, data): scalar CSC of the upper triangle (csc_utils.hpp:74-193)"""
        w = C.c_int(0 if which == "H" else 1)
        nnz = C.c_int64()
        check(self.lib.gr_bal_export_csc(self.h, w, C.byref(nnz), None, None, None))
        dim = self.n if which == "H" else 9 * self.Nc
        p = np.zeros(dim + 1, np.int64)
        i = np.zeros(nnz.value, np.int64)
        x = np.zeros(nnz.value, self.dt)
        check(self.lib.gr_bal_export_csc(self.h, w, C.byref(nnz), _ptr(p), _ptr(i), _ptr(x)))
        return p, i, x

    # --- optimizer ------------------------------------------------------------------------
    def levenberg_marquardt(self, solver=SOLVER_PCG_SCHUR, iterations=10, initial_damping=1e-4,
                            use_identity=False, pcg_max_iter=10, pcg_tol=1.0, pcg_rej=5.0, profile=False, early_stop=False):
        opt = LMOptions(solver, iterations, initial_damping, int(use_identity), pcg_max_iter, pcg_tol, pcg_rej,
                        int(profile), int(early_stop))
        st = LMStats()
        # trace buffers: one (iterations + 1) x 2 array per call, its address taken once (numpy's .ctypes is slow)
        tr = np.full((2, iterations + 1), np.nan)
        base = tr.__array_interface__["data"][0]
        check(self.lib.gr_bal_levenberg_marquardt(self.h, C.byref(opt), C.byref(st), C.c_void_p(base), C.c_void_p(base + 8 * (iterations + 1))))
        k = st.iterations_run + 1
        stats = {f: getattr(st, f) for f, _ in LMStats._fields_}
        return tr[0, :k], tr[1, :k], stats

    def direct_solver_info(self):
        """gr_bal_direct_solver_info: what solver_update_structure(SOLVER_DENSE_SCHUR) set up (tile-sparse or dense factor)."""
        from ._lib import DirectSolverInfo
        info = DirectSolverInfo()
        check(self.lib.gr_bal_direct_solver_info(self.h, C.byref(info)))
        return {f: getattr(info, f) for f, _ in DirectSolverInfo._fields_}

    def comm_info(self):
        """gr_bal_comm_info: the communicator this problem uses after the start-up self-test (audit record of a sharded run)"""
        info = _lib.CommInfo()
        check(self.lib.gr_bal_comm_info(self.h, C.byref(info)))
        d = {k: getattr(info, k) for k, _ in _lib.CommInfo._fields_ if k != "reserved"}
        d["transport_name"] = {0: "none", 1: "rccl", 2: "ipc-mailbox", 3: "in-process test group"}.get(d["transport"], "?")
        return d

    def kernel_stats(self):
        arr = (KernelStat * 64)()
        n = C.c_int()
        check(self.lib.gr_bal_kernel_stats(self.h, arr, C.c_int(64), C.byref(n)))
        out = {}
        for i in range(min(n.value, 64)):
            k = arr[i]
            out[k.name.decode()] = dict(launches=k.launches, active_launches=k.active_launches, total_ms=k.total_ms,
                                        bytes_per_launch=k.bytes_per_launch, flops_per_launch=k.flops_per_launch)
        return out


def dense_cholesky_solve(A, b, device=0):
    """x = A^-1 b on the MFMA Cholesky of GR_SOLVER_DENSE_SCHUR (gr_dense_cholesky_solve).

    A: (n, n) SPD, numpy (host) or torch CUDA tensor, row-major, lower triangle read; b: (n,).
    Returns (x as numpy array, device seconds of the factorisation)."""
    L = _lib.lib()
    n = A.shape[0]
    if isinstance(A, np.ndarray):
        dt = A.dtype
        A = np.ascontiguousarray(A)
        lda = A.strides[0] // A.itemsize
    else:
        dt = np.float64 if A.element_size() == 8 else np.float32
        lda = A.stride(0)
    b = np.ascontiguousarray(np.asarray(b, dt))
    x = np.zeros(n, dt)
    sec = C.c_double()
    check(L.gr_dense_cholesky_solve(C.c_int(F64 if dt == np.float64 else F32), C.c_int64(n), _ptr(A), C.c_int64(lda),
                                    _ptr(b), _ptr(x), C.c_int(device), None, C.byref(sec)))
    return x, sec.value


class SparseCholesky:
    """gr_spchol: nested-dissection tile Cholesky of a block-sparse SPD matrix (structure once, values per solve).

    block_row / block_col: the UPPER blocks (row <= col, every diagonal block present) of a matrix of num_nodes x num_nodes blocks of
    block_size x block_size scalars.  solve(blocks, b): blocks [num_blocks, block_size, block_size] with blocks[q][r, c] = A[bs row + r, bs col + c]
    (numpy host arrays: the library stages them), b [num_nodes * block_size]; returns x."""

    def __init__(self, num_nodes, block_size, block_row, block_col, dtype=np.float64, device=0):
        self._L = _lib.lib()
        self.dtype = np.dtype(dtype)
        self.n, self.bs, self.nb = int(num_nodes), int(block_size), len(block_row)
        r = np.ascontiguousarray(block_row, np.int64); c = np.ascontiguousarray(block_col, np.int64)
        self._h = C.c_void_p()
        check(self._L.gr_spchol_create(C.byref(self._h), C.c_int(F64 if self.dtype == np.float64 else F32), C.c_int64(self.n), C.c_int32(self.bs),
                                       C.c_int64(self.nb), _ptr(r), _ptr(c), C.c_int(device), None))

    def info(self):
        from ._lib import DirectSolverInfo
        o = DirectSolverInfo()
        check(self._L.gr_spchol_info(self._h, C.byref(o)))
        return {k: getattr(o, k) for k, _ in o._fields_}

    def solve(self, blocks, b):
        # the ABI takes column-major blocks: element (r, c) of block q at q bs^2 + c bs + r
        bl = np.ascontiguousarray(np.asarray(blocks, self.dtype).reshape(self.nb, self.bs, self.bs).transpose(0, 2, 1))
        bb = np.ascontiguousarray(np.asarray(b, self.dtype))
        x = np.zeros(self.n * self.bs, self.dtype)
        check(self._L.gr_spchol_factor_solve(self._h, _ptr(bl), _ptr(bb), _ptr(x)))
        return x

    def close(self):
        if self._h:
            self._L.gr_spchol_destroy.restype = None
            self._L.gr_spchol_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def model_evaluate(cameras, points, obs, dtype=np.float64, device=0):
    """gr_bal_model_evaluate: residual (n, 2), Jc (n, 18) and Jp (n, 6) of the engine's camera model on n triples"""
    lib = _lib.lib()
    dt = np.dtype(dtype)
    c = np.ascontiguousarray(cameras, dtype=dt).reshape(-1, 9)
    p = np.ascontiguousarray(points, dtype=dt).reshape(-1, 3)
    o = np.ascontiguousarray(obs, dtype=dt).reshape(-1, 2)
    n = len(c)
    r, jc, jp = np.zeros((n, 2), dt), np.zeros((n, 18), dt), np.zeros((n, 6), dt)
    check(lib.gr_bal_model_evaluate(C.c_int(F32 if dt == np.float32 else F64), C.c_int64(n), _ptr(c), _ptr(p), _ptr(o), _ptr(r), _ptr(jc), _ptr(jp),
                                    C.c_int(device), C.c_void_p(0)))
    return r, jc, jp
