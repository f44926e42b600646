"""Oracle pinning, part 3: the BAL pipeline (oracle/bal_pipeline.hpp) against
  * a dense numpy restatement of tests/schur_cpu_ref.cpp:8-51 (Hpp - Hpl blkinv3(Hll) Hpl^T,
    b_S, landmark back-substitution) at the reference's own 1e-12 (tests/schur.cu:180-239),
  * the reference's relational solver assertions (tests/schur.cu:242-389),
  * the committed golden fixtures under tests/golden/ (regression pins generated from this
    oracle by tests/golden/make_golden.py — NOT reference outputs: the reference cannot be built
    here)."""
import os

import numpy as np
import pytest

from graphite_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def dense_from_csc(p, i, x, n):
    A = np.zeros((n, n))
    for col in range(n):
        for k in range(p[col], p[col + 1]):
            A[i[k], col] = x[k]
    return A + np.triu(A, 1).T


def linearized(oracle_mod, prob, dtype=np.float64, mu=None):
    o = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    o.linearize()
    o.hessian_update()
    if mu is not None:
        o.apply_damping(mu)
    return o


def test_schur_vs_dense_reference_2x3(oracle_mod):
    """tests/schur.cu:113-240 (undamped)."""
    prob = synth.schur_test_fixture()
    o = linearized(oracle_mod, prob)
    assert o.chi2() != 0.0                                   # schur.cu:144
    o.schur_update()
    n, pd = o.n, 18
    H = dense_from_csc(*o.export_csc("H"), n)
    Hpp, Hpl, Hll = H[:pd, :pd], H[:pd, pd:], H[pd:, pd:]
    Hll_inv = np.zeros_like(Hll)
    for k in range(0, 9, 3):                                 # schur_cpu_ref.cpp:23-31
        Hll_inv[k:k + 3, k:k + 3] = np.linalg.inv(Hll[k:k + 3, k:k + 3])
    S_ref = Hpp - Hpl @ Hll_inv @ Hpl.T
    S = dense_from_csc(*o.export_csc("S"), pd)
    assert np.abs(S - S_ref).max() <= 1e-12 * np.abs(S_ref).max()
    b = o.get("b")
    b_S_ref = b[:pd] - Hpl @ Hll_inv @ b[pd:]                # schur_cpu_ref.cpp:37-42
    assert np.abs(o.get("b_schur") - b_S_ref).max() < 1e-12 * max(1.0, np.abs(b_S_ref).max())
    dx_p = 0.01 * np.arange(1, pd + 1)                       # schur.cu:211-214
    xl_ref = Hll_inv @ (b[pd:] - Hpl.T @ dx_p)               # schur_cpu_ref.cpp:44-51
    assert np.abs(o.landmark_update(dx_p) - xl_ref).max() < 1e-12 * max(1.0, np.abs(xl_ref).max())
    assert np.allclose(o.schur_matvec(dx_p), S_ref @ dx_p, rtol=1e-12, atol=1e-9)


def test_hessian_layout_matches_reference_block_order(oracle_mod):
    """hessian.hpp:257-288: upper blocks, column-major sorted; cameras first, then points."""
    prob = synth.schur_test_fixture()
    o = linearized(oracle_mod, prob)
    values, colptr, rowidx, offsets = o.export_hessian()
    # 2 camera diagonal blocks, then per point: (c0,l), (c1,l), (l,l)
    assert list(colptr) == [0, 1, 2, 5, 8, 11]
    assert list(rowidx) == [0, 1, 0, 1, 2, 0, 1, 3, 0, 1, 4]
    assert list(np.diff(list(offsets) + [len(values)])) == [81, 81, 27, 27, 9, 27, 27, 9, 27, 27, 9]
    J = np.zeros((12, 27))                                   # H == J^T J from the scaled Jacobians
    Jc, Jp = o.get("Jc").reshape(6, 9, 2), o.get("Jp").reshape(6, 3, 2)
    for f in range(6):
        J[2 * f:2 * f + 2, 9 * prob.cam_idx[f]:9 * prob.cam_idx[f] + 9] = Jc[f].T
        J[2 * f:2 * f + 2, 18 + 3 * prob.pt_idx[f]:18 + 3 * prob.pt_idx[f] + 3] = Jp[f].T
    H = dense_from_csc(*o.export_csc("H"), 27)
    assert np.allclose(H, J.T @ J, rtol=1e-12, atol=1e-12)
    assert np.allclose(np.diag(H), 1.0, atol=1e-9)          # column scaling: unit diagonal (graph.hpp:262-270)
    assert np.allclose(o.get("b"), -J.T @ o.get("res"), rtol=1e-12, atol=1e-12)


def solve(oracle_mod, prob, kind, mu=1e-4, **kw):
    o = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx)
    o.linearize()
    o.solver_update_values(kind)
    o.solver_set_damping(kind, mu)
    return o.solver_solve(kind, **kw)[0]


def test_full_vs_schur_direct_solve(oracle_mod):
    """SchurTests.EigenSchur, tests/schur.cu:242-289: |dx_full - dx_schur| < 1e-8."""
    prob = synth.schur_test_fixture()
    full = solve(oracle_mod, prob, oracle_mod.SOLVER_LDLT)
    schur = solve(oracle_mod, prob, oracle_mod.SOLVER_LDLT_SCHUR)
    assert np.abs(full - schur).max() < 1e-8
    prob = synth.make_config("mini-50")
    full = solve(oracle_mod, prob, oracle_mod.SOLVER_LDLT)
    schur = solve(oracle_mod, prob, oracle_mod.SOLVER_LDLT_SCHUR)
    assert np.abs(full - schur).max() < 1e-8 * max(1.0, np.abs(full).max())


def test_pcg_schur_vs_direct(oracle_mod):
    """SchurTests.PCGSchur, tests/schur.cu:340-389: 512 it, tol 1e-14, rejection 1e6, 5e-4."""
    prob = synth.schur_test_fixture()
    direct = solve(oracle_mod, prob, oracle_mod.SOLVER_LDLT_SCHUR)
    pcg = solve(oracle_mod, prob, oracle_mod.SOLVER_PCG_SCHUR, max_iter=512, tol=1e-14, rej=1e6)
    assert np.abs(direct - pcg).max() < 5e-4


def test_matrix_free_pcg_converges_to_direct(oracle_mod):
    prob = synth.make_config("mini-50")
    direct = solve(oracle_mod, prob, oracle_mod.SOLVER_LDLT)
    pcg = solve(oracle_mod, prob, oracle_mod.SOLVER_PCG, max_iter=400, tol=1e-20, rej=1e9)
    assert np.abs(direct - pcg).max() / np.abs(direct).max() < 1e-5


@pytest.mark.parametrize("solver", [0, 1, 3, 4])
def test_lm_converges(oracle_mod, solver):
    prob = synth.make_config("mini-50")
    o = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx)
    ct, lt, st = o.levenberg_marquardt(solver=solver, iterations=10)
    assert ct[-1] < 0.05 * ct[0] and np.all(np.diff(ct) <= 0)
    assert abs(ct[-1] / prob.shape[2] - 0.33) < 0.1          # MSE -> ~2 sigma^2 (1 - dof) with sigma = 0.5 px


def test_golden_fixtures(oracle_mod):
    g = np.load(os.path.join(GOLD, "schur_2x3_f64.npz"))
    prob = synth.schur_test_fixture()
    o = linearized(oracle_mod, prob)
    o.schur_update()
    for k in ("res", "scales", "b", "Hcc", "Hcp", "Hll", "S", "b_schur"):
        assert np.allclose(o.get(k), g[k], rtol=1e-12, atol=1e-13), k
    g = np.load(os.path.join(GOLD, "mini50_lm_f64.npz"))
    prob = synth.make_config("mini-50")
    assert np.array_equal(prob.cam_idx, g["cam_idx"]) and np.allclose(prob.obs, g["obs"])  # generator is pinned too
    for solver, key in ((0, "chi2_pcg_schur"), (1, "chi2_pcg"), (4, "chi2_ldlt_schur")):
        o = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx)
        ct, _, _ = o.levenberg_marquardt(solver=solver, iterations=8)
        assert np.allclose(ct, g[key], rtol=1e-9), key


def test_oracle_single_reduction_pcg_equals_reference_recurrence(oracle_mod):
    """solve_pcg_cg (Chronopoulos-Gear, the documented variant the GPU engine runs on landmark shards) against solve_pcg
    (solver/pcg.hpp:61-232): same iterates up to rounding, same iteration counts for cap, tolerance and rejection exits."""
    from graphite_amd import synth
    prob = synth.make_config("mini-50")
    for dtype, bar in ((np.float64, 1e-11), (np.float32, 1e-3)):
        out = []
        for variant in (0, 1):
            o = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
            o.set_pcg_single_reduction(variant)
            o.linearize()
            o.solver_update_values(oracle_mod.SOLVER_PCG)
            o.solver_set_damping(oracle_mod.SOLVER_PCG, 1e-4)
            res = [o.solver_solve(oracle_mod.SOLVER_PCG, max_iter=m, tol=t, rej=r) for m, t, r in ((10, 0.0, 1e30), (25, 1e-3, 1e30), (25, 0.0, 0.5))]
            ct, _, st = o.levenberg_marquardt(solver=oracle_mod.SOLVER_PCG, iterations=6)
            out.append((res, ct, st["pcg_iterations"]))
        for (dx0, it0), (dx1, it1) in zip(out[0][0], out[1][0]):
            assert it0 == it1
            assert np.abs(dx0 - dx1).max() / np.abs(dx0).max() < bar
        assert out[0][2] == out[1][2]
        assert np.max(np.abs(out[0][1] - out[1][1]) / out[0][1]) < (1e-12 if dtype == np.float64 else 1e-4)


def test_oracle_fixed_vertices(oracle_mod):
    """VertexDescriptor::set_fixed (vertex.hpp:262-264) as the oracle restates it: fixed vertices do not move, their step
    and gradient entries are exactly zero, and the Schur direct solve still equals the full-system one (tests/schur.cu's
    relation) with vertices of both kinds fixed."""
    from graphite_amd import synth
    prob = synth.make_config("mini-50")
    cf = np.zeros(prob.shape[0], bool)
    cf[[0, 7]] = True
    pf = np.zeros(prob.shape[1], bool)
    pf[:10] = True
    finals = {}
    for solver in (oracle_mod.SOLVER_PCG, oracle_mod.SOLVER_PCG_SCHUR, oracle_mod.SOLVER_LDLT_SCHUR, oracle_mod.SOLVER_LDLT):
        o = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx)
        o.set_fixed(cf, pf)
        o.linearize()
        b = o.get("b")
        fixed_entries = np.concatenate([np.repeat(cf, 9), np.repeat(pf, 3)])
        assert np.all(b[fixed_entries] == 0) and np.abs(b[~fixed_entries]).max() > 0
        ct, _, _ = o.levenberg_marquardt(solver=solver, iterations=5)
        c, p = o.get_params()
        assert np.array_equal(c[cf], prob.cameras[cf]) and np.array_equal(p[pf], prob.points[pf])
        assert np.abs(c[~cf] - prob.cameras[~cf]).max() > 1e-3 and ct[-1] < 0.1 * ct[0]
        finals[solver] = ct[-1]
    assert abs(finals[oracle_mod.SOLVER_LDLT_SCHUR] - finals[oracle_mod.SOLVER_LDLT]) / finals[oracle_mod.SOLVER_LDLT] < 1e-9


def test_amd_ordering_of_the_cpu_baseline(oracle_mod):
    """oracle/amd.hpp — the approximate minimum degree ordering Eigen::SimplicialLDLT applies by default, i.e. the ordering of the
    reference's eigen_solver CPU path (/root/reference/src/eigen_solver.cpp:10-13).  No reference vector pins it (Eigen is absent,
    SURVEY 8c), so it is held to its defining properties: a permutation; on unstructured sparse symmetric patterns a fill far
    below the natural order's; on a BAL-shaped reduced camera system a fill within 15 % of EXACT minimum degree on the camera
    graph and of the block-level AMD; and the CPU baseline's LM trace does not depend on which ordering factorises S or H."""
    import scipy.sparse as sp
    for n, dens in ((300, 0.02), (3000, 0.0015)):
        A = sp.random(n, n, density=dens, random_state=1, format="csc")
        U = sp.triu(A + A.T + sp.eye(n), format="csc")
        perm = oracle_mod.amd_order(n, U.indptr, U.indices)
        assert sorted(perm.tolist()) == list(range(n))
        assert oracle_mod.ldlt_fill(n, U.indptr, U.indices, perm) < 0.6 * oracle_mod.ldlt_fill(n, U.indptr, U.indices)
    # arrow matrix: natural order (dense row first) fills completely, minimum degree orders the hub last: no fill at all
    n = 200
    rows = [0] * n + list(range(1, n)); cols = list(range(n)) + list(range(1, n))
    U = sp.csc_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    U.sum_duplicates()
    perm = oracle_mod.amd_order(n, U.indptr, U.indices)
    assert oracle_mod.ldlt_fill(n, U.indptr, U.indices) == n * (n - 1) // 2
    assert oracle_mod.ldlt_fill(n, U.indptr, U.indices, perm) == n - 1
    # the CPU baseline: same LM trace whichever ordering factorises (eigen-schur and eigen legs), fill reported per ordering
    prob = synth.make_problem(40, 1500, 7000, seed=11, window=10)
    base = oracle_mod.CpuBaseline(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    for solver in (oracle_mod.SOLVER_LDLT_SCHUR, oracle_mod.SOLVER_LDLT):
        out = {}
        for ordering in (0, 1, 2):
            base.reset()
            ct, _, st, tm = base.levenberg_marquardt(solver, 4, threads=2, ordering=ordering)
            out[ordering] = (ct, tm["ldlt_nnz"])
        assert np.allclose(out[2][0], out[1][0], rtol=1e-9) and np.allclose(out[2][0], out[0][0], rtol=1e-9)
        assert out[2][1] <= 1.15 * out[1][1], (out[2][1], out[1][1])   # AMD against exact minimum degree on the camera graph
