"""Oracle pinning, part 1: the BAL camera model (oracle/bal_model.hpp).

The reference holds no golden vector for residual or Jacobian VALUES ("parity
unpinned"; tests/schur.cu:144 only asserts chi2 != 0), so the restatement is pinned by
(i) central finite differences of its own residual, (ii) an independent numpy projection,
(iii) the documented theta == 0 quirk of projection_jacobians.cuh:175-212."""
import numpy as np
import pytest

from graphite_amd import synth

CAM_REF = np.array([0.12, -0.08, 0.03, 0.25, -0.10, 0.20, 800.0, 0.01, -0.001])  # tests/schur.cu:52-56
PT_REF = np.array([0.1, 0.0, 2.0], dtype=np.float32).astype(np.float64)          # tests/schur.cu:63


def fd_jacobian(oracle_mod, cam, pt, obs):
    x = np.concatenate([cam, pt])
    J = np.zeros((2, 12))
    for i in range(12):
        h = 1e-6 * max(1.0, abs(x[i]))
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        J[:, i] = (oracle_mod.bal_residual(xp[:9], xp[9:], obs) - oracle_mod.bal_residual(xm[:9], xm[9:], obs)) / (2 * h)
    return J


@pytest.mark.parametrize("theta_scale", [1e-4, 0.05, 0.45, 0.55, 1.5, 3.0])
def test_jacobian_matches_finite_differences(oracle_mod, theta_scale):
    rng = np.random.default_rng(int(theta_scale * 1e4))
    for _ in range(5):
        r = rng.normal(size=3)
        r *= theta_scale / np.linalg.norm(r)
        cam = np.concatenate([r, rng.normal(0, 0.1, 2), [-5.0 + rng.normal(0, 0.1)], [rng.uniform(500, 1500)],
                              [rng.normal(0, 0.01)], [rng.normal(0, 0.001)]])
        pt = rng.uniform(-1, 1, 3)
        obs = rng.normal(0, 10, 2)
        _, Jc, Jp = oracle_mod.bal_residual_jacobian(cam, pt, obs)
        J = np.concatenate([Jc, Jp], 1)
        Jfd = fd_jacobian(oracle_mod, cam, pt, obs)
        assert np.abs(J - Jfd).max() / np.abs(J).max() < 1e-7  # rel 1e-7 fp64


def test_reference_fixture_camera(oracle_mod):
    res, Jc, Jp = oracle_mod.bal_residual_jacobian(CAM_REF, PT_REF, np.zeros(2))
    assert np.abs(np.concatenate([Jc, Jp], 1) - fd_jacobian(oracle_mod, CAM_REF, PT_REF, np.zeros(2))).max() < 1e-4
    # independent numpy projection (graphite_amd.synth.project)
    proj = synth.project(CAM_REF[None], PT_REF[None], np.array([0]), np.array([0]))[0]
    assert np.allclose(res, proj, rtol=1e-13)


def test_theta_zero_quirk(oracle_mod):
    """theta == 0: identity rotation in the residual (reprojection_error.cuh:72-77) and a ZERO
    rotation block in the Jacobian (projection_jacobians.cuh:175-212), although the true
    derivative is not zero there."""
    cam = CAM_REF.copy()
    cam[:3] = 0.0
    res, Jc, Jp = oracle_mod.bal_residual_jacobian(cam, PT_REF, np.zeros(2))
    assert np.all(Jc[:, :3] == 0.0)
    tiny = cam.copy()
    tiny[:3] = [1e-12, 0, 0]
    res2, Jc2, _ = oracle_mod.bal_residual_jacobian(tiny, PT_REF, np.zeros(2))
    assert np.allclose(res, res2, rtol=1e-9)
    assert np.abs(Jc2[:, :3]).max() > 1.0          # the limit from theta > 0 is the true derivative
    assert np.allclose(Jc[:, 3:], Jc2[:, 3:], rtol=1e-9) and np.allclose(Jp, _, rtol=1e-9)


def test_series_branch_is_continuous(oracle_mod):
    """so3 coefficients switch from series to closed form at theta^2 = 0.25."""
    base = np.array([0.3, -0.2, 0.346])
    for s in (0.5 - 1e-9, 0.5 + 1e-9):
        cam = CAM_REF.copy()
        cam[:3] = base / np.linalg.norm(base) * s
        _, Jc, _ = oracle_mod.bal_residual_jacobian(cam, PT_REF, np.zeros(2))
        if s < 0.5:
            lo = Jc
    assert np.allclose(lo, Jc, rtol=1e-7)


def test_float32_model_close_to_float64(oracle_mod):
    r64 = oracle_mod.bal_residual_jacobian(CAM_REF, PT_REF, np.zeros(2))
    r32 = oracle_mod.bal_residual_jacobian(CAM_REF.astype(np.float32), PT_REF.astype(np.float32), np.zeros(2, np.float32))
    for a, b in zip(r64, r32):
        assert np.abs(a - b).max() / np.abs(a).max() < 2e-5


def test_small_inverse(oracle_mod):
    rng = np.random.default_rng(3)
    for n in (3, 9):
        A = rng.normal(size=(n, n))
        A = A @ A.T + n * np.eye(n)
        Ainv, ok = oracle_mod.small_inverse(A)
        assert ok and np.allclose(Ainv @ A, np.eye(n), atol=1e-12)
