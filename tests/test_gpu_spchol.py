"""gr_spchol (include/graphite_mi355x.h): the nested-dissection tile Cholesky of GR_SOLVER_DENSE_SCHUR on ANY block-sparse SPD matrix — what
EigenLDLTSolver of the header-only layer hands the Hessian of a graph without an elimination order (a pose graph) to, the numerical role of
Eigen::SimplicialLDLT in solver/eigen.hpp:49-98.  Against scipy's sparse LU on the same matrix."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu


def _chain_graph_matrix(nodes, bs, window, extra, seed, dtype=np.float64):
    """SPD block matrix of a chain with closures inside a window (a locally connected pose graph): A = sum over edges of J^T J + I"""
    rng = np.random.default_rng(seed)
    edges = {(i, i + 1) for i in range(nodes - 1)} | {(i, i + 2) for i in range(nodes - 2)}
    for _ in range(extra):
        i = int(rng.integers(0, nodes - 2))
        j = min(nodes - 1, i + 1 + int(rng.integers(1, window)))
        edges.add((i, j))
    edges = sorted(edges)
    blocks = {(i, i): np.eye(bs) for i in range(nodes)}
    for (i, j) in edges:
        J = rng.standard_normal((bs, 2 * bs))
        H = J.T @ J
        blocks[(i, i)] = blocks[(i, i)] + H[:bs, :bs]
        blocks[(j, j)] = blocks[(j, j)] + H[bs:, bs:]
        blocks[(i, j)] = blocks.get((i, j), np.zeros((bs, bs))) + H[:bs, bs:]
    keys = sorted(blocks, key=lambda k: (k[1], k[0]))  # by block column, as the header-only layer's Hessian lists them
    row = np.array([k[0] for k in keys]); col = np.array([k[1] for k in keys])
    vals = np.stack([blocks[k] for k in keys]).astype(dtype)
    # scipy matrix (both triangles)
    R, Cc, V = [], [], []
    for (i, j), B in zip(keys, vals):
        for r in range(bs):
            for c in range(bs):
                R.append(bs * i + r); Cc.append(bs * j + c); V.append(B[r, c])
                if i != j:
                    R.append(bs * j + c); Cc.append(bs * i + r); V.append(B[r, c])
    A = sp.csc_matrix((np.array(V, np.float64), (R, Cc)), shape=(bs * nodes, bs * nodes))
    return row, col, vals, A


@pytest.mark.parametrize("nodes,bs,dtype,tol", [(3000, 3, np.float64, 1e-10), (1500, 6, np.float64, 1e-10), (900, 9, np.float64, 1e-10), (2500, 3, np.float32, 2e-4), (700, 2, np.float64, 1e-10)])
def test_block_sparse_solve_against_scipy(nodes, bs, dtype, tol):
    import graphite_amd as ga
    row, col, vals, A = _chain_graph_matrix(nodes, bs, window=40, extra=2 * nodes, seed=nodes + bs, dtype=dtype)
    b = np.random.default_rng(1).standard_normal(bs * nodes)
    ch = ga.bal.SparseCholesky(nodes, bs, row, col, dtype=dtype)
    info = ch.info()
    assert info["sparse"] == 1 and info["supernodes"] > 1 and info["levels"] < info["tile_columns"] and info["factor_bytes"] < info["dense_bytes"]
    x = ch.solve(vals, b)
    ref = spla.spsolve(A, b)
    assert np.abs(x - ref).max() <= tol * np.abs(ref).max()
    # values change, structure stays: a second factorisation on the same handle
    x2 = ch.solve(2.0 * vals, b)
    assert np.abs(x2 - 0.5 * ref).max() <= tol * np.abs(ref).max()
    ch.close()


def test_errors_are_reported():
    import graphite_amd as ga
    from graphite_amd._lib import GraphiteError
    # a complete graph does not dissect: the caller is told to use the dense solver
    n = 30
    row, col = np.triu_indices(n)
    with pytest.raises(GraphiteError, match="does not dissect"):
        ga.bal.SparseCholesky(n, 3, row, col)
    # a missing diagonal block, a lower block
    with pytest.raises(GraphiteError, match="diagonal block"):
        ga.bal.SparseCholesky(3, 3, [0, 1, 0], [0, 1, 2])
    with pytest.raises(GraphiteError, match="upper block"):
        ga.bal.SparseCholesky(3, 3, [0, 1, 2, 2], [0, 1, 2, 1])
    # an indefinite matrix: the factorisation says so
    row, col, vals, A = _chain_graph_matrix(800, 3, window=20, extra=800, seed=9)
    ch = ga.bal.SparseCholesky(800, 3, row, col)
    bad = vals.copy()
    bad[np.flatnonzero(row == col)[400]] *= -1.0
    with pytest.raises(GraphiteError, match="pivot"):
        ch.solve(bad, np.ones(2400))
    x = ch.solve(vals, np.ones(2400))  # the handle is usable afterwards
    assert np.abs(A @ x - 1.0).max() < 1e-8
    ch.close()


@pytest.mark.parametrize("nodes,extra_edges", [(600, 0), (600, 5), (900, 300)], ids=["no-edges", "five-edges", "two-components-and-singletons"])
def test_disconnected_node_graphs(nodes, extra_edges):
    """A graph of unary factors only (the circle example with more vertices) has a block-DIAGONAL Hessian: no edges at all; graphs with a few
    edges or several components must factorise as well."""
    import graphite_amd as ga
    from graphite_amd._lib import GraphiteError
    rng = np.random.default_rng(nodes + extra_edges)
    bs = 2
    blocks = {(i, i): (lambda M: M @ M.T + np.eye(bs))(rng.standard_normal((bs, bs))) for i in range(nodes)}
    for _ in range(extra_edges):
        i, j = sorted(rng.choice(nodes // 2, 2, replace=False))   # edges only among the first half: the second half stays singletons
        J = rng.standard_normal((bs, 2 * bs)); H = J.T @ J
        blocks[(i, i)] = blocks[(i, i)] + H[:bs, :bs]; blocks[(j, j)] = blocks[(j, j)] + H[bs:, bs:]
        blocks[(i, j)] = blocks.get((i, j), np.zeros((bs, bs))) + H[:bs, bs:]
    keys = sorted(blocks, key=lambda k: (k[1], k[0]))
    row = np.array([k[0] for k in keys]); col = np.array([k[1] for k in keys]); vals = np.stack([blocks[k] for k in keys])
    A = np.zeros((bs * nodes, bs * nodes))
    for (i, j), B in zip(keys, vals):
        A[bs * i:bs * i + bs, bs * j:bs * j + bs] = B
        A[bs * j:bs * j + bs, bs * i:bs * i + bs] = B.T
    b = rng.standard_normal(bs * nodes)
    try:
        ch = ga.bal.SparseCholesky(nodes, bs, row, col)
    except GraphiteError as ex:  # allowed answer: "does not dissect" (the caller then takes the dense form)
        assert "does not dissect" in str(ex)
        return
    x = ch.solve(vals, b)
    ref = np.linalg.solve(A, b)
    assert np.abs(x - ref).max() <= 1e-10 * np.abs(ref).max()
    ch.close()
