"""ORACLE — test infrastructure only.

ctypes binding of ``oracle/libgraphite_oracle.so`` (the CPU restatement of the
reference's hot path, see ``oracle/*.hpp`` headers for the file:line map).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; nothing under ``graphite_amd/`` does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgraphite_oracle.so")
_lib = None

SOLVER_PCG_SCHUR, SOLVER_PCG, SOLVER_PCG_IDENTITY, SOLVER_LDLT, SOLVER_LDLT_SCHUR = range(5)
LOSS_DEFAULT, LOSS_HUBER = 0, 1
MODEL_BAL, MODEL_K3, MODEL_PINHOLE = 0, 1, 2


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))]
    if (force or not os.path.exists(_LIB_PATH)
            or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", C.c_float
    if dtype == np.float64:
        return "f64", C.c_double
    raise TypeError(dtype)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _fn(name, dtype, restype=None):
    sfx, _ = _sfx(dtype)
    f = getattr(lib(), f"{name}_{sfx}")
    f.restype = restype
    return f


def bal_residual(cam, pt, obs):
    dt = cam.dtype
    out = np.zeros(2, dt)
    _fn("gro_bal_residual", dt)(_p(np.ascontiguousarray(cam)), _p(np.ascontiguousarray(pt)),
                                _p(np.ascontiguousarray(obs)), _p(out))
    return out


def bal_residual_jacobian(cam, pt, obs):
    dt = cam.dtype
    res, Jc, Jp = np.zeros(2, dt), np.zeros(18, dt), np.zeros(6, dt)
    _fn("gro_bal_residual_jacobian", dt)(_p(np.ascontiguousarray(cam)), _p(np.ascontiguousarray(pt)),
                                         _p(np.ascontiguousarray(obs)), _p(res), _p(Jc), _p(Jp))
    # column-major E x d -> (2, d)
    return res, Jc.reshape(9, 2).T.copy(), Jp.reshape(3, 2).T.copy()


def small_inverse(A):
    """A: (n, n) array -> inverse (the cublas matinvBatched role)."""
    dt = A.dtype
    n = A.shape[0]
    a = np.asfortranarray(A).ravel(order="F").copy()
    out = np.zeros_like(a)
    ok = _fn("gro_small_inverse", dt, C.c_int)(C.c_int(n), _p(a), _p(out))
    return out.reshape(n, n, order="F"), bool(ok)


class GenericOps:
    """Thin wrappers of the generic per-factor kernels (oracle/generic_ops.hpp)."""

    def __init__(self, dtype):
        self.dt = np.dtype(dtype)
        _, self.ct = _sfx(dtype)

    def _u(self, a):
        return np.ascontiguousarray(a, dtype=np.uint64)

    def chi2(self, residuals, pmat, E, loss_kind=None, loss_delta=None):
        n = len(residuals) // E
        chi2 = np.zeros(n, self.dt)
        d = np.zeros(n, self.dt)
        lk = None if loss_kind is None else np.ascontiguousarray(loss_kind, dtype=np.int32)
        ld = None if loss_delta is None else np.ascontiguousarray(loss_delta, dtype=self.dt)
        _fn("gro_chi2", self.dt)(C.c_size_t(n), C.c_int(E), _p(residuals), _p(pmat), _p(lk), _p(ld),
                                 _p(chi2), _p(d))
        return chi2, d

    def _slot(self, active_ids, E, d, jac, ids, N, I, hessian_ids, active_state):
        return (C.c_size_t(len(active_ids)), _p(self._u(active_ids)), C.c_int(E), C.c_int(d), _p(jac),
                _p(self._u(ids)), C.c_size_t(N), C.c_size_t(I), _p(self._u(hessian_ids)),
                _p(np.ascontiguousarray(active_state, dtype=np.uint8)))

    def scalar_diagonal(self, active_ids, E, d, jac, ids, N, I, hessian_ids, active_state, pmat, dchi2, diagonal):
        _fn("gro_scalar_diagonal", self.dt)(*self._slot(active_ids, E, d, jac, ids, N, I, hessian_ids, active_state),
                                            _p(pmat), _p(dchi2), _p(diagonal))

    def block_diagonal(self, active_ids, E, d, jac, ids, N, I, hessian_ids, active_state, pmat, dchi2, blocks):
        _fn("gro_block_diagonal", self.dt)(*self._slot(active_ids, E, d, jac, ids, N, I, hessian_ids, active_state),
                                           _p(pmat), _p(dchi2), _p(blocks))

    def scale_jacobians(self, active_ids, E, d, jac, ids, N, I, hessian_ids, active_state, scales):
        _fn("gro_scale_jacobians", self.dt)(*self._slot(active_ids, E, d, jac, ids, N, I, hessian_ids, active_state),
                                            _p(scales))

    def compute_b(self, active_ids, E, d, jac, ids, N, I, hessian_ids, active_state, residuals, pmat, dchi2, b):
        _fn("gro_compute_b", self.dt)(*self._slot(active_ids, E, d, jac, ids, N, I, hessian_ids, active_state),
                                      _p(residuals), _p(pmat), _p(dchi2), _p(b))

    def Jv(self, active_ids, E, d, jac, ids, N, I, hessian_ids, active_state, x, y):
        _fn("gro_Jv", self.dt)(*self._slot(active_ids, E, d, jac, ids, N, I, hessian_ids, active_state), _p(x), _p(y))

    def JtPv(self, active_ids, E, d, jac, ids, N, I, hessian_ids, active_state, pmat, dchi2, x, y):
        _fn("gro_JtPv", self.dt)(*self._slot(active_ids, E, d, jac, ids, N, I, hessian_ids, active_state),
                                 _p(pmat), _p(dchi2), _p(x), _p(y))

    def hessian_block(self, active_ids, E, slot_i, slot_j, ids, N, block_offsets, pmat, dchi2, hessian):
        """slot_* = (d, jac, I, hessian_ids, active_state)"""
        di, ji, Ii, hi, ai = slot_i
        dj, jj, Ij, hj, aj = slot_j
        _fn("gro_hessian_block", self.dt)(
            C.c_size_t(len(active_ids)), _p(self._u(active_ids)), C.c_int(E),
            C.c_int(di), _p(ji), C.c_size_t(Ii), _p(self._u(hi)), _p(np.ascontiguousarray(ai, dtype=np.uint8)),
            C.c_int(dj), _p(jj), C.c_size_t(Ij), _p(self._u(hj)), _p(np.ascontiguousarray(aj, dtype=np.uint8)),
            _p(self._u(ids)), C.c_size_t(N), _p(self._u(block_offsets)), _p(pmat), _p(dchi2), _p(hessian))

    def augment_block_diagonal(self, D, blocks, scalar_diag, mu, use_identity, active_state):
        nv = len(active_state)
        _fn("gro_augment_block_diagonal", self.dt)(C.c_size_t(nv), C.c_int(D), _p(blocks), _p(scalar_diag),
                                                   self.ct(mu), C.c_int(int(use_identity)),
                                                   _p(np.ascontiguousarray(active_state, dtype=np.uint8)))

    def apply_block_jacobi(self, D, z, r, blocks, hessian_ids, active_state):
        nv = len(active_state)
        _fn("gro_apply_block_jacobi", self.dt)(C.c_size_t(nv), C.c_int(D), _p(z), _p(r), _p(blocks),
                                               _p(self._u(hessian_ids)),
                                               _p(np.ascontiguousarray(active_state, dtype=np.uint8)))

    def apply_update(self, D, params, delta_x, scales, hessian_ids, active_state):
        nv = len(active_state)
        _fn("gro_apply_update", self.dt)(C.c_size_t(nv), C.c_int(D), _p(params), _p(delta_x), _p(scales),
                                         _p(self._u(hessian_ids)),
                                         _p(np.ascontiguousarray(active_state, dtype=np.uint8)))


_GET = dict(res=0, Jc=1, Jp=2, scales=3, b=4, Hcc=5, Hcp=6, Hll=7, S=8, b_schur=9, Hll_inv=10,
            chi2_vec=11, dchi2=12, prev_diag=13)


class BalOracle:
    """Handle on the CPU restatement of the BAL hot path (oracle/bal_pipeline.hpp)."""

    def __init__(self, cams, pts, obs, cam_idx, pt_idx, dtype=np.float64):
        self.dt = np.dtype(dtype)
        _, self.ct = _sfx(dtype)
        cams = np.ascontiguousarray(cams, dtype=self.dt).reshape(-1, 9)
        pts = np.ascontiguousarray(pts, dtype=self.dt).reshape(-1, 3)
        obs = np.ascontiguousarray(obs, dtype=self.dt).reshape(-1, 2)
        self.Nc, self.Np, self.No = len(cams), len(pts), len(obs)
        self.n = 9 * self.Nc + 3 * self.Np
        ci = np.ascontiguousarray(cam_idx, dtype=np.int32)
        pi = np.ascontiguousarray(pt_idx, dtype=np.int32)
        f = _fn("gro_bal_create", self.dt, C.c_void_p)
        self.h = C.c_void_p(f(C.c_size_t(self.Nc), C.c_size_t(self.Np), C.c_size_t(self.No), _p(cams), _p(pts),
                              _p(obs), _p(ci), _p(pi)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _fn("gro_bal_destroy", self.dt)(self.h)
                self.h = None
        except Exception:
            pass

    def _call(self, name, *args, restype=None):
        return _fn(name, self.dt, restype)(self.h, *args)

    def set_loss(self, kind, delta=0.0):
        self._call("gro_bal_set_loss", C.c_int(kind), self.ct(delta))

    def set_scale_system(self, on):
        self._call("gro_bal_set_scale_system", C.c_int(int(on)))

    def set_factor_tables(self, pmat=None, loss_kinds=None, loss_deltas=None, fdata=None):
        """Per-factor precision matrices (No, 4) row-major, loss kinds / deltas (No,), constraint data (No, 4)
        (factor.hpp:158-174, :373-412); None = identity / the loss of set_loss / none."""
        pm = None if pmat is None else np.ascontiguousarray(pmat, dtype=self.dt).reshape(-1)
        lk = None if loss_kinds is None else np.ascontiguousarray(loss_kinds, dtype=np.int32)
        ld = None if loss_kinds is None else np.ascontiguousarray(loss_deltas, dtype=self.dt)
        fd = None if fdata is None else np.ascontiguousarray(fdata, dtype=self.dt).reshape(-1)
        self._call("gro_bal_set_factor_tables", _p(pm), _p(lk), _p(ld), _p(fd))

    def set_jacobian_storage(self, mode):
        """Graph<T, S>'s Jacobian storage type S: 0 = T, 1 = bfloat16 (round to nearest even), 2 = float (types.hpp:8-43)"""
        self._call("gro_bal_set_jacobian_storage", C.c_int({"T": 0, "bf16": 1, "f32": 2}.get(mode, mode)))

    def set_model(self, kind):
        """0 the BAL camera (analytic Jacobian), 1 BAL + k3 r^6 (data[:, 0]), 2 pinhole (6, 3) -> 2 (oracle/user_models.hpp)"""
        self._call("gro_bal_set_model", C.c_int(int(kind)))

    def set_fixed(self, cam_fixed=None, pt_fixed=None):
        """VertexDescriptor::set_fixed (vertex.hpp:262): boolean masks over cameras / points"""
        cf = np.ascontiguousarray(np.zeros(self.Nc, np.uint8) if cam_fixed is None else (np.asarray(cam_fixed) != 0).astype(np.uint8))
        pf = np.ascontiguousarray(np.zeros(self.Np, np.uint8) if pt_fixed is None else (np.asarray(pt_fixed) != 0).astype(np.uint8))
        self._call("gro_bal_set_fixed", cf.ctypes.data_as(C.c_void_p), pf.ctypes.data_as(C.c_void_p))

    def set_pcg_single_reduction(self, on):
        """documented variant (not in the reference): Chronopoulos-Gear recurrence for SOLVER_PCG / SOLVER_PCG_IDENTITY"""
        self._call("gro_bal_set_pcg_single_reduction", C.c_int(int(on)))

    def set_params(self, cams, pts):
        self._call("gro_bal_set_params", _p(np.ascontiguousarray(cams, dtype=self.dt)),
                   _p(np.ascontiguousarray(pts, dtype=self.dt)))

    def get_params(self):
        cams = np.zeros((self.Nc, 9), self.dt)
        pts = np.zeros((self.Np, 3), self.dt)
        self._call("gro_bal_get_params", _p(cams), _p(pts))
        return cams, pts

    def compute_error(self):
        self._call("gro_bal_compute_error")

    def chi2(self):
        return self._call("gro_bal_chi2", restype=C.c_double)

    def linearize(self):
        self._call("gro_bal_linearize")

    def hessian_update(self):
        self._call("gro_bal_hessian_update")

    def apply_damping(self, mu, use_identity=False):
        self._call("gro_bal_apply_damping", self.ct(mu), C.c_int(int(use_identity)))

    def schur_update(self):
        self._call("gro_bal_schur_update")

    def get(self, name):
        which = C.c_int(_GET[name])
        size = self._call("gro_bal_get", which, None, restype=C.c_size_t)
        out = np.zeros(size, self.dt)
        self._call("gro_bal_get", which, _p(out), restype=C.c_size_t)
        return out

    def schur_structure(self):
        nnzb = self._call("gro_bal_nnzb_schur", restype=C.c_size_t)
        colptr = np.zeros(self.Nc + 1, np.int64)
        rowidx = np.zeros(nnzb, np.int64)
        self._call("gro_bal_schur_structure", _p(colptr), _p(rowidx))
        return colptr, rowidx

    def schur_matvec(self, x):
        x = np.ascontiguousarray(x, dtype=self.dt)
        y = np.zeros(9 * self.Nc, self.dt)
        self._call("gro_bal_schur_matvec", _p(x), _p(y))
        return y

    def landmark_update(self, xp):
        xp = np.ascontiguousarray(xp, dtype=self.dt)
        xl = np.zeros(3 * self.Np, self.dt)
        self._call("gro_bal_landmark_update", _p(xp), _p(xl))
        return xl

    def export_hessian(self):
        """Reference layout: (values, block colptr, block rowidx, value offsets)."""
        nb = C.c_size_t(0)
        nv = self._call("gro_bal_export_hessian", None, None, None, None, C.byref(nb), restype=C.c_size_t)
        values = np.zeros(nv, self.dt)
        colptr = np.zeros(self.Nc + self.Np + 1, np.int64)
        rowidx = np.zeros(nb.value, np.int64)
        offsets = np.zeros(nb.value, np.int64)
        self._call("gro_bal_export_hessian", _p(values), _p(colptr), _p(rowidx), _p(offsets), None,
                   restype=C.c_size_t)
        return values, colptr, rowidx, offsets

    def export_csc(self, which):
        """which: 'H' or 'S' -> (indptr, indices, data) scalar upper CSC."""
        w = C.c_int(0 if which == "H" else 1)
        dim = self.n if which == "H" else 9 * self.Nc
        nnz = self._call("gro_bal_export_csc", w, None, None, None, restype=C.c_size_t)
        p = np.zeros(dim + 1, np.int64)
        i = np.zeros(nnz, np.int64)
        x = np.zeros(nnz, self.dt)
        self._call("gro_bal_export_csc", w, _p(p), _p(i), _p(x), restype=C.c_size_t)
        return p, i, x

    def solver_update_values(self, kind):
        self._call("gro_bal_solver_update_values", C.c_int(kind))

    def solver_set_damping(self, kind, mu, use_identity=False):
        self._call("gro_bal_solver_set_damping", C.c_int(kind), self.ct(mu), C.c_int(int(use_identity)))

    def solver_solve(self, kind, max_iter=10, tol=1.0, rej=5.0):
        x = np.zeros(self.n, self.dt)
        it = self._call("gro_bal_solver_solve", C.c_int(kind), C.c_int(max_iter), C.c_double(tol),
                        C.c_double(rej), _p(x), restype=C.c_int)
        return x, it

    def apply_update(self, dx):
        self._call("gro_bal_apply_update", _p(np.ascontiguousarray(dx, dtype=self.dt)))

    def backup(self):
        self._call("gro_bal_backup")

    def revert(self):
        self._call("gro_bal_revert")

    def levenberg_marquardt(self, solver=SOLVER_PCG_SCHUR, iterations=10, initial_damping=1e-4,
                            use_identity=False, pcg_max_iter=10, pcg_tol=1.0, pcg_rej=5.0, early_stop=False):
        ct = np.zeros(iterations + 1, np.float64)
        lt = np.zeros(iterations + 1, np.float64)
        st = np.zeros(6, np.float64)
        run = self._call("gro_bal_lm", C.c_int(solver), C.c_int(iterations), C.c_double(initial_damping),
                         C.c_int(int(use_identity)), C.c_int(pcg_max_iter), C.c_double(pcg_tol),
                         C.c_double(pcg_rej), _p(ct), _p(lt), _p(st), C.c_int(int(early_stop)), restype=C.c_int)
        k = int(st[0]) + 1
        stats = dict(iterations_run=int(st[0]), accepted=int(st[1]), pcg_iterations=int(st[2]),
                     solve_seconds=st[3], loop_seconds=st[4], setup_seconds=st[5], ok=bool(run))
        return ct[:k], lt[:k], stats


class CpuBaseline:
    """Timed CPU comparator (oracle/cpu_baseline.hpp): the oracle's LM with OpenMP-parallel assembly and the
    single-thread simplicial LDL^T (full H or Schur), or the all-cores block-Jacobi PCG."""

    TIME_KEYS = ("linearize", "hessian", "schur", "export_csc", "ldlt_analyze", "ldlt_factor", "ldlt_solve",
                 "backsub", "pcg", "update_chi2", "loop", "setup", "ldlt_nnz", "threads")

    def __init__(self, cams, pts, obs, cam_idx, pt_idx, dtype=np.float64):
        self.dt = np.dtype(dtype)
        cams = np.ascontiguousarray(cams, dtype=self.dt).reshape(-1, 9)
        pts = np.ascontiguousarray(pts, dtype=self.dt).reshape(-1, 3)
        obs = np.ascontiguousarray(obs, dtype=self.dt).reshape(-1, 2)
        ci = np.ascontiguousarray(cam_idx, dtype=np.int32)
        pi = np.ascontiguousarray(pt_idx, dtype=np.int32)
        self._init = (cams, pts)
        f = _fn("gro_baseline_create", self.dt, C.c_void_p)
        self.h = C.c_void_p(f(C.c_size_t(len(cams)), C.c_size_t(len(pts)), C.c_size_t(len(obs)), _p(cams), _p(pts),
                              _p(obs), _p(ci), _p(pi)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _fn("gro_baseline_destroy", self.dt)(self.h)
                self.h = None
        except Exception:
            pass

    def reset(self):
        _fn("gro_bal_set_params", self.dt)(self.h, _p(self._init[0]), _p(self._init[1]))

    def levenberg_marquardt(self, solver, iterations, threads=0, ordering=1, initial_damping=1e-4, pcg_max_iter=10,
                            pcg_tol=1.0, pcg_rej=5.0):
        """threads: 0 = all host cores; ordering: 2 = AMD of the factorised matrix (oracle/amd.hpp: Eigen::SimplicialLDLT's default, the
        reference's own choice), 1 = exact minimum degree on the camera graph, 0 = reverse Cuthill-McKee."""
        ct = np.zeros(iterations + 1)
        lt = np.zeros(iterations + 1)
        st = np.zeros(6)
        tm = np.zeros(14)
        f = _fn("gro_baseline_lm", self.dt, C.c_int)
        f(self.h, C.c_int(solver), C.c_int(iterations), C.c_double(initial_damping), C.c_int(pcg_max_iter),
          C.c_double(pcg_tol), C.c_double(pcg_rej), C.c_int(threads), C.c_int(ordering), _p(ct), _p(lt), _p(st), _p(tm))
        k = int(st[0]) + 1
        stats = dict(iterations_run=int(st[0]), accepted=int(st[1]), pcg_iterations=int(st[2]), solve_seconds=st[3],
                     loop_seconds=st[4], setup_seconds=st[5])
        times = dict(zip(self.TIME_KEYS, (float(x) for x in tm)))
        return ct[:k], lt[:k], stats, times


def circle_lm(points, radius, fixed=None, factor_on=None, solver="eigen", iterations=100, initial_damping=1e-6, use_identity=False,
              pcg_max_iter=50, pcg_tol=1e-20, pcg_rej=10.0):
    """BASELINE configs[0] on the CPU (oracle/circle_fit.hpp): examples/circle.cu's unary-factor graph through
    optimizer::levenberg_marquardt with EigenLDLTSolver ("eigen") or PCGSolver + IdentityPreconditioner ("pcg").
    Returns (chi2 trace, lambda trace, final points, stats)."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2).copy()
    n = len(pts)
    fx = np.ascontiguousarray(np.zeros(n, np.uint8) if fixed is None else (np.asarray(fixed) != 0).astype(np.uint8))
    on = np.ascontiguousarray(np.ones(n, np.uint8) if factor_on is None else (np.asarray(factor_on) != 0).astype(np.uint8))
    ct = np.zeros(iterations + 1); lt = np.zeros(iterations + 1)
    stats = np.zeros(2, np.int32)
    f = lib().gro_circle_lm_f64
    f.restype = C.c_int
    it = f(C.c_size_t(n), C.c_double(radius), _p(pts), _p(fx), _p(on), C.c_int({"eigen": 0, "pcg": 1}[solver]), C.c_int(iterations),
           C.c_double(initial_damping), C.c_int(int(use_identity)), C.c_int(pcg_max_iter), C.c_double(pcg_tol), C.c_double(pcg_rej),
           _p(ct), _p(lt), _p(stats))
    return ct[:it + 1], lt[:it + 1], pts, dict(iterations_run=it, accepted=int(stats[0]), pcg_iterations=int(stats[1]))


def amd_order(n, indptr, indices):
    """oracle/amd.hpp: approximate minimum degree order (perm[new] = old) of a symmetric matrix given by the scalar CSC of its upper triangle"""
    ap = np.ascontiguousarray(indptr, dtype=np.int64); ai = np.ascontiguousarray(indices, dtype=np.int64)
    perm = np.zeros(n, np.int64)
    f = lib().gro_amd_order
    f.restype = None
    f(C.c_int64(n), _p(ap), _p(ai), _p(perm))
    return perm


def ldlt_fill(n, indptr, indices, perm=None):
    """nnz of the strictly lower part of L for the simplicial LDL^T (oracle/sparse_ldlt.hpp) under `perm` (None: natural order)"""
    ap = np.ascontiguousarray(indptr, dtype=np.int64); ai = np.ascontiguousarray(indices, dtype=np.int64)
    pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
    f = lib().gro_ldlt_fill
    f.restype = C.c_int64
    return int(f(C.c_int64(n), _p(ap), _p(ai), _p(pm)))
