"""Oracle pinning, part 2: every literal known answer the reference's own unit tests hold for
the generic per-factor kernels (tests/factor.cu, tests/vertex.cu), replayed on the oracle's
restatement of those kernels (oracle/generic_ops.hpp).  Inputs and expected values are the
literal numbers of the cited tests; EXPECT_FLOAT_EQ = 4 ULP in fp32."""
import numpy as np
import pytest

F = np.float32


def feq(a, b):
    a, b = np.asarray(a, F), np.asarray(b, F)
    return np.all(np.abs(a - b) <= 4 * np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(F)))


@pytest.fixture
def ops(oracle_mod):
    return oracle_mod.GenericOps(F)


def unary_problem(nf, jac_row, vertex_x=7.0, obs=2.5):
    """One Vec2 vertex (id 10, hessian column 0), nf identical unary E=1 factors."""
    jac = np.tile(np.asarray(jac_row, F), nf)
    res = np.full(nf, vertex_x - obs, F) if len(jac_row) == 2 and jac_row[1] == 0 else None
    return dict(active=np.arange(nf), ids=np.zeros(nf), hid=np.array([0]), act=np.array([0], np.uint8),
                jac=jac, pmat=np.ones(nf, F), res=res)


def test_compute_error_residual():            # tests/factor.cu:139-157
    assert feq(F(7.0) - F(2.5), 4.5)


def test_scale_jacobians(ops):                # tests/factor.cu:383-423
    p = unary_problem(2, [2.0, 3.0])
    jac = p["jac"].copy()
    ops.scale_jacobians(p["active"], 1, 2, jac, p["ids"], 1, 0, p["hid"], p["act"], np.array([2.0, 3.0], F))
    assert feq(jac, [4, 9, 4, 9])


def test_compute_b_accumulates(ops, oracle_mod):          # tests/factor.cu:425-466
    p = unary_problem(2, [1.0, 0.0])
    res = np.full(2, 4.5, F)
    _, dchi2 = ops.chi2(res, p["pmat"], 1)
    b = np.array([3.0, -7.0], F)
    for _ in range(2):
        ops.compute_b(p["active"], 1, 2, p["jac"], p["ids"], 1, 0, p["hid"], p["act"], res, p["pmat"], dchi2, b)
    assert feq(b, [3.0 - 4.0 * 4.5, -7.0])


def test_compute_b_huber(ops, oracle_mod):                # tests/factor.cu:468-509
    p = unary_problem(2, [1.0, 0.0])
    res = np.full(2, 4.5, F)
    _, dchi2 = ops.chi2(res, p["pmat"], 1, [oracle_mod.LOSS_HUBER] * 2, [1.0, 1.0])
    b = np.array([5.0, -11.0], F)
    for _ in range(2):
        ops.compute_b(p["active"], 1, 2, p["jac"], p["ids"], 1, 0, p["hid"], p["act"], res, p["pmat"], dchi2, b)
    assert feq(b, [5.0 - 4.0, -11.0])


def test_block_diagonal(ops):                 # tests/factor.cu:511-555
    p = unary_problem(2, [2.0, 3.0])
    blocks = np.zeros(4, F)
    ops.block_diagonal(p["active"], 1, 2, p["jac"], p["ids"], 1, 0, p["hid"], p["act"], p["pmat"], np.ones(2, F), blocks)
    assert feq(blocks, [8, 12, 12, 18])


def test_scalar_diagonal(ops):                # tests/factor.cu:557-595
    p = unary_problem(2, [2.0, 3.0])
    diag = np.zeros(2, F)
    ops.scalar_diagonal(p["active"], 1, 2, p["jac"], p["ids"], 1, 0, p["hid"], p["act"], p["pmat"], np.ones(2, F), diag)
    assert feq(diag, [8, 18])


def test_Jv_with_fixed_inactive_and_no_factors(ops):      # tests/factor.cu:597-668
    p = unary_problem(2, [1.0, 0.0])
    x = np.array([3.0, 5.0], F)
    out = np.zeros(2, F)
    ops.Jv(p["active"], 1, 2, p["jac"], p["ids"], 1, 0, p["hid"], p["act"], x, out)
    assert feq(out, [3, 3])
    for state, init in ((1, [17, 23]), (0x80, [29, 31])):   # fixed bit, MSB "unused" bit
        out = np.array(init, F)
        ops.Jv(p["active"], 1, 2, p["jac"], p["ids"], 1, 0, p["hid"], np.array([state], np.uint8), x, out)
        assert feq(out, init)
    out = np.array([37, 41], F)                              # no active factors
    ops.Jv(np.zeros(0), 1, 2, p["jac"], p["ids"], 1, 0, p["hid"], p["act"], x, out)
    assert feq(out, [37, 41])


def test_JtPv_huber(ops, oracle_mod):                     # tests/factor.cu:670-756
    p = unary_problem(2, [1.0, 0.0])
    res = np.full(2, 4.5, F)
    _, dchi2 = ops.chi2(res, p["pmat"], 1, [oracle_mod.LOSS_HUBER] * 2, [1.0, 1.0])
    x = np.array([9.0, 9.0], F)
    out = np.zeros(2, F)
    ops.JtPv(p["active"], 1, 2, p["jac"], p["ids"], 1, 0, p["hid"], p["act"], p["pmat"], dchi2, x, out)
    assert feq(out, [4, 0])
    for state, init in ((1, [43, 47]), (0x80, [53, 59])):
        out = np.array(init, F)
        ops.JtPv(p["active"], 1, 2, p["jac"], p["ids"], 1, 0, p["hid"], np.array([state], np.uint8), p["pmat"], dchi2, x, out)
        assert feq(out, init)


def test_chi2_huber(ops, oracle_mod):                     # tests/factor.cu:758-784
    res = np.array([4.5, 0.5], F)
    chi2, _ = ops.chi2(res, np.ones(2, F), 1, [oracle_mod.LOSS_HUBER] * 2, [1.0, 1.0])
    assert feq(chi2, [8.0, 0.25]) and feq(chi2.sum(), 8.25)


def test_compute_hessian_blocks(ops):         # tests/factor.cu:854-967
    """v0 (col 0), v1 (col 2); unary J=[2,3] on v0 and on v1; binary J0=[1,2], J1=[3,4] on (v0,v1).
    Offsets [0,8,0,4,8]; 12 values [5,8,8,13 / 3,6,4,8 / 13,18,18,25]."""
    hid = np.array([0, 2])
    act = np.zeros(2, np.uint8)
    H = np.zeros(12, F)
    one = np.ones(2, F)
    # unary descriptor: factor 0 on v0 -> block (0,0) offset 0, factor 1 on v1 -> block (1,1) offset 8
    ju = np.array([2, 3, 2, 3], F)
    ids_u = np.array([0, 1])
    su = (2, ju, 0, hid, act)
    ops.hessian_block(np.arange(2), 1, su, su, ids_u, 1, np.array([0, 8]), one, one, H)
    # binary descriptor, one factor (v0, v1): pairs (0,0)->0, (0,1)->4, (1,1)->8
    j0, j1 = np.array([1, 2], F), np.array([3, 4], F)
    ids_b = np.array([0, 1])
    s0, s1 = (2, j0, 0, hid, act), (2, j1, 1, hid, act)
    for si, sj, off in ((s0, s0, 0), (s0, s1, 4), (s1, s1, 8)):
        ops.hessian_block(np.arange(1), 1, si, sj, ids_b, 2, np.array([off]), one, one, H)
    assert feq(H, [5, 8, 8, 13, 3, 6, 4, 8, 13, 18, 18, 25])


def test_hessian_block_transposed_when_columns_inverted(ops):   # ops/hessian.hpp:39-49
    hid = np.array([2, 0])   # vertex 0 sits AFTER vertex 1 in the Hessian
    act = np.zeros(2, np.uint8)
    H = np.zeros(4, F)
    one = np.ones(1, F)
    s0, s1 = (2, np.array([1, 2], F), 0, hid, act), (2, np.array([3, 4], F), 1, hid, act)
    ops.hessian_block(np.arange(1), 1, s0, s1, np.array([0, 1]), 2, np.array([0]), one, one, H)
    # stored block is (row = v1, col = v0): J1^T J0 = [[3,6],[4,8]] column-major -> [3,4,6,8]
    assert feq(H, [3, 4, 6, 8])


def test_apply_update(ops):                   # tests/vertex.cu:76-119
    params = np.array([1, 2, 10, 20], F)
    ops.apply_update(2, params, np.array([4, -2, 100, 200], F), np.array([0.5, 2, 1, 1], F), np.array([0, 2]),
                     np.array([0, 1], np.uint8))
    assert feq(params, [1 + 4 * 0.5, 2 - 2 * 2, 10, 20])


def test_augment_block_diagonal(ops):         # tests/vertex.cu:121-166
    blocks = np.full(8, -1, F)
    sd = np.array([2, 4, 8, 16], F)
    ops.augment_block_diagonal(2, blocks, sd, 0.5, False, np.array([0, 1], np.uint8))
    assert feq(blocks, [2 + 0.5 * 2, -1, -1, 4 + 0.5 * 4, -1, -1, -1, -1])


def test_apply_block_jacobi(ops):             # tests/vertex.cu:168-226
    z = np.full(4, -5, F)
    r = np.array([11, 13, 17, 19], F)
    blocks = np.array([2, 3, 5, 7, 101, 103, 107, 109], F)
    ops.apply_block_jacobi(2, z, r, blocks, np.array([0, 2]), np.array([0, 1], np.uint8))
    assert feq(z, [2 * 11 + 5 * 13, 3 * 11 + 7 * 13, -5, -5])
