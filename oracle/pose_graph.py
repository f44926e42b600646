"""ORACLE — test infrastructure only (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/).

CPU restatement (numpy, fp64) of the reference's generic pipeline on a graph of BINARY factors between vertices of ONE descriptor:
a planar pose graph, SE(2) between-factors — the SLAM back-end shape the reference's README names and BASELINE configs[0] calls a
"2D pose-graph".  Every stage follows the reference file it cites; sums over factors run in factor order (np.add.at), which is one
valid instance of the reference's unordered float atomics.

  error / Jacobians of the factor   the user's traits (tests/cpp/test_pose_graph.hip carries the same formulas):
                                    e = [R_i^T (t_j - t_i) - m_t ; th_j - th_i - m_th], blocks E x d column-major (ops/error.hpp:146-149)
  chi2, rho, rho'                   ops/chi2.hpp:10-44, loss.hpp:15-51 (Default / Huber), precision read row-major (ops/linearize.hpp:283)
  Graph::linearize                  graph.hpp:236-290: column scales 1 / (eps + sqrt(diag(J^T rho' P J))) in double, J scaled in
                                    place (ops/linearize.hpp:142-180), b = -J^T rho' P r (:240-303); fixed vertices have no column and
                                    their Jacobian blocks are skipped (ops/linearize.hpp:24)
  ordering                          graph.hpp:100-149: active vertices in vertex order, d columns each
  BlockJacobiPreconditioner         preconditioner/block_jacobi.hpp:79-186: per-vertex J^T rho' P J blocks (ops/hessian.hpp:169-240),
                                    diagonal + mu clamp(d, 1e-6, 1e32) or + mu (ops/hessian.hpp:80-112), batched inverse, z = B r
  PCGSolver::solve                  solver/pcg.hpp:61-232 (preconditioner applied to r / |r|; rejection ratio; x backup)
  levenberg_marquardt               optimizer/levenberg_marquardt.hpp:20-47,110-242

PARITY UNPINNED: the reference holds no test, example or golden vector on a pose graph.  What ties this restatement to the reference is the
formula of every line cited above (the same formulas oracle/generic_ops.hpp states for the graphs the reference's vectors DO pin), its own
consistency checks on the CPU (tests/test_oracle_pose_graph.py) and the agreement of the HIP generic kernels — pinned on the reference's
vectors — with it at 1e-12 on the same graphs (tests/test_generic_api.py::test_pose_graph_on_the_generic_kernels).
"""
import numpy as np

EPS = np.finfo(np.float64).eps


class PoseGraphOracle:
    def __init__(self, poses, fixed, edges, meas, info, huber_delta=0.0, priors=None):
        """priors = (vertex ids, measurements [n, 3], information matrices [n, 3, 3]): a SECOND factor descriptor of unary factors
        e = x_i - m (Jacobian = identity), default loss — the graph then has two factor descriptors on its one vertex descriptor"""
        self.x = np.array(poses, dtype=np.float64).reshape(-1, 3).copy()
        self.n = len(self.x)
        self.fixed = np.asarray(fixed).astype(bool)
        self.i = np.asarray(edges)[:, 0].astype(np.int64)
        self.j = np.asarray(edges)[:, 1].astype(np.int64)
        self.meas = np.asarray(meas, dtype=np.float64)
        self.P = np.asarray(info, dtype=np.float64)          # [F, 3, 3], read row-major
        self.delta = float(huber_delta)
        if priors is None:
            self.pi = np.zeros(0, np.int64); self.pm = np.zeros((0, 3)); self.pP = np.zeros((0, 3, 3))
        else:
            self.pi = np.asarray(priors[0]).astype(np.int64); self.pm = np.asarray(priors[1], dtype=np.float64).reshape(-1, 3)
            self.pP = np.asarray(priors[2], dtype=np.float64).reshape(-1, 3, 3)
        # Graph::initialize_optimization: columns for the active vertices in vertex order
        used = np.zeros(self.n, bool)
        used[self.i] = True; used[self.j] = True; used[self.pi] = True
        self.active = used & ~self.fixed
        self.col = np.full(self.n, -1, np.int64)
        self.col[self.active] = 3 * np.arange(int(self.active.sum()))
        self.dim = 3 * int(self.active.sum())
        self.scale_system = True

    # ---- the factor -----------------------------------------------------------------------------------------------------------
    def _error(self):
        a, b = self.x[self.i], self.x[self.j]
        c, s = np.cos(a[:, 2]), np.sin(a[:, 2])
        dx, dy = b[:, 0] - a[:, 0], b[:, 1] - a[:, 1]
        return np.stack([c * dx + s * dy - self.meas[:, 0], -s * dx + c * dy - self.meas[:, 1], b[:, 2] - a[:, 2] - self.meas[:, 2]], 1)

    def _jacobians(self):
        a, b = self.x[self.i], self.x[self.j]
        c, s = np.cos(a[:, 2]), np.sin(a[:, 2])
        dx, dy = b[:, 0] - a[:, 0], b[:, 1] - a[:, 1]
        F = len(self.i)
        Ji = np.zeros((F, 3, 3)); Jj = np.zeros((F, 3, 3))   # [f, row e, col d]
        Ji[:, 0, 0] = -c; Ji[:, 0, 1] = -s; Ji[:, 0, 2] = -s * dx + c * dy
        Ji[:, 1, 0] = s; Ji[:, 1, 1] = -c; Ji[:, 1, 2] = -c * dx - s * dy
        Ji[:, 2, 2] = -1.0
        Jj[:, 0, 0] = c; Jj[:, 0, 1] = s
        Jj[:, 1, 0] = -s; Jj[:, 1, 1] = c
        Jj[:, 2, 2] = 1.0
        return Ji, Jj

    def compute_error(self):
        self.r = self._error()
        self.pr = self.x[self.pi] - self.pm

    def chi2(self):
        raw = np.einsum("fi,fij,fj->f", self.r, self.P, self.r)
        if self.delta > 0:
            big = ~(raw <= self.delta ** 2)
            val = np.where(big, 2 * np.sqrt(np.where(big, raw, 1.0)) * self.delta - self.delta ** 2, raw)
            self.dchi2 = np.where(big, self.delta / np.sqrt(np.where(big, raw, 1.0)), 1.0)
        else:
            val = raw; self.dchi2 = np.ones_like(raw)
        tot = 0.0
        for v in val:       # sequential, factor order
            tot += v
        for v in np.einsum("fi,fij,fj->f", self.pr, self.pP, self.pr):   # the prior descriptor's factors, default loss
            tot += v
        return tot

    def linearize(self):
        self.compute_error()
        Ji, Jj = self._jacobians()
        Ji[~self.active[self.i]] = 0.0; Jj[~self.active[self.j]] = 0.0   # no column: block skipped
        self.chi2()
        w = self.dchi2
        PJi = np.einsum("fab,fbd->fad", self.P, Ji); PJj = np.einsum("fab,fbd->fad", self.P, Jj)
        if self.scale_system:
            diag = np.zeros(self.dim)
            for J, PJ, v in ((Ji, PJi, self.i), (Jj, PJj, self.j)):
                d = np.einsum("fad,fad->fd", J, PJ) * w[:, None]
                m = self.active[v]
                np.add.at(diag, (self.col[v[m]][:, None] + np.arange(3)[None, :]).ravel(), d[m].ravel())
            pa = self.active[self.pi]
            np.add.at(diag, (self.col[self.pi[pa]][:, None] + np.arange(3)[None, :]).ravel(), np.einsum("fdd->fd", self.pP)[pa].ravel())
            self.scales = 1.0 / (EPS + np.sqrt(diag))
        else:
            self.scales = np.ones(self.dim)
        for J, v in ((Ji, self.i), (Jj, self.j)):
            m = self.active[v]
            J[m] *= self.scales[self.col[v[m]][:, None] + np.arange(3)[None, :]][:, None, :]
        self.Ji, self.Jj = Ji, Jj
        # b = -J^T rho' P r
        Pr = np.einsum("fab,fb->fa", self.P, self.r) * w[:, None]
        self.b = np.zeros(self.dim)
        for J, v in ((Ji, self.i), (Jj, self.j)):
            g = -np.einsum("fad,fa->fd", J, Pr)
            m = self.active[v]
            np.add.at(self.b, (self.col[v[m]][:, None] + np.arange(3)[None, :]).ravel(), g[m].ravel())
        # the priors: J = diag(scales of the vertex), b -= J^T P r
        pa = self.active[self.pi]
        self.pS = np.zeros((len(self.pi), 3))
        self.pS[pa] = self.scales[self.col[self.pi[pa]][:, None] + np.arange(3)[None, :]]
        gp = -self.pS * np.einsum("fab,fb->fa", self.pP, self.pr)
        np.add.at(self.b, (self.col[self.pi[pa]][:, None] + np.arange(3)[None, :]).ravel(), gp[pa].ravel())

    # ---- BlockJacobiPreconditioner ---------------------------------------------------------------------------------------------
    def block_diagonal(self):
        nb = self.dim // 3
        B = np.zeros((nb, 3, 3))
        for J, v in ((self.Ji, self.i), (self.Jj, self.j)):
            blk = np.einsum("fad,fab,fbe->fde", J, self.P, J) * self.dchi2[:, None, None]
            m = self.active[v]
            np.add.at(B, self.col[v[m]] // 3, blk[m])
        pa = self.active[self.pi]
        np.add.at(B, self.col[self.pi[pa]] // 3, (self.pS[:, :, None] * self.pP * self.pS[:, None, :])[pa])
        self.Bdiag = B
        self.hdiag = np.einsum("kii->ki", B).reshape(-1).copy()

    def set_damping(self, mu, use_identity):
        self.mu, self.use_identity = mu, use_identity
        A = self.Bdiag.copy()
        d = np.einsum("kii->ki", A)
        nd = d + mu if use_identity else d + mu * np.clip(d, 1.0e-6, 1.0e32)
        for q in range(3):
            A[:, q, q] = nd[:, q]
        self.Binv = np.linalg.inv(A)

    def apply_precond(self, r, identity):
        if identity:
            return r.copy()
        return np.einsum("kab,kb->ka", self.Binv, r.reshape(-1, 3)).reshape(-1)

    # ---- matrix-free operator: J^T rho' P J v + mu D v (ops/product.hpp:195,405) ---------------------------------------------
    def operator(self, v, diag):
        u = np.zeros((len(self.i), 3))
        for J, vx in ((self.Ji, self.i), (self.Jj, self.j)):
            m = self.active[vx]
            vv = np.zeros((len(vx), 3))
            vv[m] = v[self.col[vx[m]][:, None] + np.arange(3)[None, :]]
            u += np.einsum("fad,fd->fa", J, vv)
        u = np.einsum("fab,fb->fa", self.P, u) * self.dchi2[:, None]
        y = np.zeros(self.dim)
        for J, vx in ((self.Ji, self.i), (self.Jj, self.j)):
            g = np.einsum("fad,fa->fd", J, u)
            m = self.active[vx]
            np.add.at(y, (self.col[vx[m]][:, None] + np.arange(3)[None, :]).ravel(), g[m].ravel())
        pa = self.active[self.pi]
        if pa.any():
            cols = self.col[self.pi[pa]][:, None] + np.arange(3)[None, :]
            up = np.einsum("fab,fb->fa", self.pP[pa], self.pS[pa] * v[cols])
            np.add.at(y, cols.ravel(), (self.pS[pa] * up).ravel())
        return y + self.mu * (1.0 if self.use_identity else diag) * v

    def solve_pcg(self, max_iter, tol, rej, identity_precond):
        x = np.zeros(self.dim); r = self.b.copy()
        diag = np.clip(self.hdiag, 1.0e-6, 1.0e32)
        z = self.apply_precond(r / np.sqrt(r @ r), identity_precond)
        p = z.copy()
        rz = r @ z
        rz0 = np.inf
        its = 0
        for k in range(max_iter):
            if rz == 0:
                break
            v2 = self.operator(p, diag)
            alpha = rz / (p @ v2)
            xb = x.copy()
            x = x + alpha * p
            r = r - alpha * v2
            z = self.apply_precond(r / np.sqrt(r @ r), identity_precond)
            rz_new = r @ z
            its = k + 1
            if abs(rz_new) > rej * rz0 or np.isnan(rz_new):
                x = xb
                break
            rz0 = min(rz0, abs(rz_new))
            beta = rz_new / rz
            rz = rz_new
            p = z + beta * p
            if abs(rz_new) < tol:
                break
        return x, its

    # ---- EigenLDLTSolver (solver/eigen.hpp:49-98): the damped system assembled sparse and solved directly (hessian.hpp:136-176 damping) ----
    def solve_direct(self):
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        rows, cols, vals = [], [], []
        ar = np.arange(3)

        def add(va, vb, blk):     # blk [F, 3, 3] at block (va, vb), both active
            m = self.active[va] & self.active[vb]
            if not m.any():
                return
            r = (self.col[va[m]][:, None, None] + ar[None, :, None]) + 0 * ar[None, None, :]
            c = (self.col[vb[m]][:, None, None] + ar[None, None, :]) + 0 * ar[None, :, None]
            rows.append(r.ravel()); cols.append(c.ravel()); vals.append(blk[m].ravel())
        W = self.P * self.dchi2[:, None, None]
        for Ja, va in ((self.Ji, self.i), (self.Jj, self.j)):
            for Jb, vb in ((self.Ji, self.i), (self.Jj, self.j)):
                add(va, vb, np.einsum("fad,fab,fbe->fde", Ja, W, Jb))
        if len(self.pi):
            add(self.pi, self.pi, self.pS[:, :, None] * self.pP * self.pS[:, None, :])
        H = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.dim, self.dim))
        d = H.diagonal()
        H = H + sp.diags(self.mu * (np.ones_like(d) if self.use_identity else np.clip(d, 1.0e-6, 1.0e32)))
        return spla.spsolve(H.tocsc(), self.b), 0

    def apply_update(self, dx):
        m = self.active
        self.x[m] += (dx * self.scales).reshape(-1, 3)

    def levenberg_marquardt(self, iterations=10, initial_damping=1e-4, use_identity=False, pcg_max_iter=10, pcg_tol=1.0, pcg_rej=5.0,
                            identity_precond=False, direct=False):
        mu, nu = initial_damping, 2.0
        self.linearize(); self.block_diagonal()
        chi2v = self.chi2()
        ct, lt = [chi2v], [mu]
        st = dict(accepted=0, pcg_iterations=0, iterations_run=0)
        for _ in range(iterations):
            self.set_damping(mu, use_identity)
            dx, its = self.solve_direct() if direct else self.solve_pcg(pcg_max_iter, pcg_tol, pcg_rej, identity_precond)
            st["pcg_iterations"] += its
            backup = self.x.copy()
            self.apply_update(dx)
            self.compute_error()
            new_chi2 = self.chi2()
            denom = float(np.sum(dx * (mu * dx + self.b))) + 1.0e-3
            rho = (chi2v - new_chi2) / denom
            if np.isfinite(new_chi2) and rho > 0:
                alpha = 1.0 - (2.0 * rho - 1.0) ** 3
                alpha = max(min(alpha, 2.0 / 3.0), 1.0 / 3.0)
                mu *= alpha; nu = 2.0
                self.linearize(); self.block_diagonal()
                st["accepted"] += 1
            else:
                self.x = backup
                self.compute_error(); self.chi2()
                mu *= nu; nu *= 2.0
                new_chi2 = chi2v
            chi2v = new_chi2
            st["iterations_run"] += 1
            ct.append(chi2v); lt.append(mu)
            if not np.isfinite(mu) or rho == 0:
                break
        return np.array(ct), np.array(lt), st
