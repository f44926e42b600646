"""oracle/pose_graph.py (numpy restatement of the reference's generic pipeline on a one-descriptor graph of binary factors) checked on the CPU:
its analytic Jacobians against central differences of its own error function, its linear algebra against dense numpy on the assembled
system, and the LM loop's behaviour.  The reference holds no golden vector for a pose graph: what pins this oracle to the reference is
that every stage is the formula of the file:line it cites, the same formulas oracle/generic_ops.hpp states for the BAL graphs (pinned to
tests/factor.cu, tests/vertex.cu there) — and these self-consistency checks."""
import numpy as np

from graphite_amd import synth
from oracle.pose_graph import PoseGraphOracle


def _graph(n=60, huber=0.0):
    p0, fx, e, m, info, _ = synth.make_pose_graph(n)
    return PoseGraphOracle(p0, fx, e, m, info, huber_delta=huber), (p0, fx, e, m, info)


def test_jacobians_are_the_derivatives_of_the_error():
    o, _ = _graph()
    Ji, Jj = o._jacobians()
    h = 1e-6
    for side, J, idx in ((0, Ji, o.i), (1, Jj, o.j)):
        for c in range(3):
            x0 = o.x.copy()
            d = np.zeros_like(o.x)
            # move every factor's side-vertex along c — one factor at a time would be the definition; vertices shared by factors make the
            # batched perturbation wrong, so perturb per factor
            num = np.zeros((len(o.i), 3))
            for f in range(len(o.i)):
                o.x = x0.copy(); o.x[idx[f], c] += h; ep = o._error()[f]
                o.x = x0.copy(); o.x[idx[f], c] -= h; em = o._error()[f]
                num[f] = (ep - em) / (2 * h)
            o.x = x0
            self_loop = o.i == o.j
            assert np.allclose(J[~self_loop, :, c], num[~self_loop], atol=1e-7)


def test_linear_system_matches_dense_numpy():
    o, _ = _graph(40)
    o.linearize(); o.block_diagonal()
    # dense J (scaled) from the oracle's blocks
    F = len(o.i)
    J = np.zeros((3 * F, o.dim))
    for f in range(F):
        for blk, v in ((o.Ji[f], o.i[f]), (o.Jj[f], o.j[f])):
            if o.active[v]:
                J[3 * f:3 * f + 3, o.col[v]:o.col[v] + 3] += blk
    W = np.zeros((3 * F, 3 * F))
    for f in range(F):
        W[3 * f:3 * f + 3, 3 * f:3 * f + 3] = o.dchi2[f] * o.P[f]
    H = J.T @ W @ J
    b = -J.T @ W @ o.r.reshape(-1)
    assert np.allclose(o.b, b, rtol=1e-12, atol=1e-12)
    v = np.random.default_rng(0).standard_normal(o.dim)
    o.set_damping(1e-3, False)
    diag = np.clip(o.hdiag, 1e-6, 1e32)
    assert np.allclose(o.operator(v, diag), H @ v + 1e-3 * diag * v, rtol=1e-11, atol=1e-11)
    assert np.allclose(np.diag(H), o.hdiag, rtol=1e-12) and np.allclose(o.hdiag, 1.0, atol=1e-9)  # scaled system: unit diagonal
    # block-Jacobi blocks are the diagonal blocks of H
    for k in range(o.dim // 3):
        assert np.allclose(o.Bdiag[k], H[3 * k:3 * k + 3, 3 * k:3 * k + 3], rtol=1e-12, atol=1e-13)
    # PCG run to convergence solves the damped system
    x, its = o.solve_pcg(500, 1e-28, 1e30, False)
    Hd = H + 1e-3 * np.diag(diag)
    assert np.allclose(Hd @ x, o.b, rtol=1e-7, atol=1e-9)


def test_lm_descends_keeps_fixed_vertices_and_huber_changes_the_trace():
    o, (p0, fx, e, m, info) = _graph(200)
    ct, lt, st = o.levenberg_marquardt(iterations=8)
    assert st["accepted"] >= 6 and ct[-1] < 0.1 * ct[0] and np.all(np.diff(ct) <= 1e-12)
    fixed = np.flatnonzero(np.asarray(fx))
    assert len(fixed) >= 1 and np.array_equal(o.x[fixed], np.asarray(p0, dtype=np.float64).reshape(-1, 3)[fixed])
    oh, _ = _graph(200, huber=0.05)
    cth, _, _ = oh.levenberg_marquardt(iterations=8)
    assert cth[0] < ct[0] and not np.allclose(cth[1:], ct[1:], rtol=1e-3)


def test_priors_as_a_second_factor_descriptor_match_dense_numpy():
    p0, fx, e, m, info, truth = synth.make_pose_graph(120)
    rng = np.random.default_rng(1)
    idx = np.arange(5, 120, 7)
    pm = np.asarray(truth)[idx] + 0.01 * rng.standard_normal((len(idx), 3))
    L = 0.3 * rng.standard_normal((len(idx), 3, 3)) + 2.0 * np.eye(3)
    P = np.einsum("fab,fcb->fac", L, L)
    P = 0.5 * (P + P.transpose(0, 2, 1))
    o = PoseGraphOracle(p0, fx, e, m, info, priors=(idx, pm, P))
    o.linearize(); o.block_diagonal(); o.set_damping(1e-3, False)
    F, n = len(o.i), o.dim
    J = np.zeros((3 * F + 3 * len(idx), n)); W = np.zeros((J.shape[0], J.shape[0]))
    r = np.concatenate([o.r.ravel(), o.pr.ravel()])
    for f in range(F):
        for blk, v in ((o.Ji[f], o.i[f]), (o.Jj[f], o.j[f])):
            if o.active[v]:
                J[3 * f:3 * f + 3, o.col[v]:o.col[v] + 3] += blk
        W[3 * f:3 * f + 3, 3 * f:3 * f + 3] = o.dchi2[f] * o.P[f]
    for q, v in enumerate(idx):
        rr = 3 * F + 3 * q
        if o.active[v]:
            J[rr:rr + 3, o.col[v]:o.col[v] + 3] = np.diag(o.scales[o.col[v]:o.col[v] + 3])
        W[rr:rr + 3, rr:rr + 3] = P[q]
    H = J.T @ W @ J
    v = rng.standard_normal(n)
    diag = np.clip(o.hdiag, 1e-6, 1e32)
    assert np.allclose(o.b, -J.T @ W @ r, rtol=1e-12, atol=1e-12)
    assert np.allclose(o.operator(v, diag), H @ v + 1e-3 * diag * v, rtol=1e-11, atol=1e-11)
    assert np.allclose(np.diag(H), o.hdiag, rtol=1e-12)
    assert abs(o.chi2() - r @ W @ r) < 1e-9 * abs(r @ W @ r)
    ct, _, st = o.levenberg_marquardt(iterations=6)
    assert st["accepted"] >= 4 and ct[-1] < ct[0]
