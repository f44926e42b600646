"""Synthetic BAL-shaped bundle-adjustment problems and BAL text I/O.

The reference ships no BAL data (git-ignored, /root/reference/.gitignore:2) and
there is no network, so every benchmark/parity input is generated here with the
shapes of BASELINE.json's configs.  File format follows the
reader in /root/reference/examples/bal.cu:70,96-141:

    <num_cameras> <num_points> <num_observations>
    <camera_idx> <point_idx> <x> <y>          (one per observation)
    <9 numbers per camera, one per line>       r(3) t(3) f k1 k2
    <3 numbers per point, one per line>
"""
from __future__ import annotations

import dataclasses

import numpy as np

# name -> (Nc, Np, No, seed, window)   (public BAL shapes as recalled in SURVEY §8)
CONFIGS = {
    "mini-6": (6, 40, 150, 11, 6),
    "mini-50": (50, 2000, 9000, 7, 16),
    "ladybug-49": (49, 7776, 31843, 1, 32),
    "ladybug-1723": (1723, 156502, 678718, 2, 32),
    "venice-1778": (1778, 993923, 5001946, 3, 1778),
    "final-13682": (13682, 4456117, 28987644, 4, 13682),
}


@dataclasses.dataclass
class BalProblem:
    cameras: np.ndarray   # (Nc, 9) float64  initial guess
    points: np.ndarray    # (Np, 3)
    obs: np.ndarray       # (No, 2)
    cam_idx: np.ndarray   # (No,) int32
    pt_idx: np.ndarray    # (No,) int32
    name: str = "synthetic"

    @property
    def shape(self):
        return len(self.cameras), len(self.points), len(self.obs)


def _rodrigues(rvec):
    theta = np.linalg.norm(rvec, axis=1)
    safe = np.where(theta > 0, theta, 1.0)
    a = rvec / safe[:, None]
    c, s = np.cos(theta), np.sin(theta)
    K = np.zeros((len(rvec), 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -a[:, 2], a[:, 1]
    K[:, 1, 0], K[:, 1, 2] = a[:, 2], -a[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -a[:, 1], a[:, 0]
    eye = np.eye(3)[None]
    R = c[:, None, None] * eye + s[:, None, None] * K + (1 - c)[:, None, None] * (a[:, :, None] * a[:, None, :])
    R[theta == 0] = np.eye(3)
    return R


def project(cameras, points, cam_idx, pt_idx):
    """BAL projection (float64 numpy), used to synthesise observations."""
    cam = cameras[cam_idx]
    X = points[pt_idx]
    R = _rodrigues(cameras[:, :3])[cam_idx]
    P = np.einsum("nij,nj->ni", R, X) + cam[:, 3:6]
    p = -P[:, :2] / P[:, 2:3]
    r2 = (p * p).sum(1)
    d = 1 + cam[:, 7] * r2 + cam[:, 8] * r2 * r2
    return cam[:, 6:7] * d[:, None] * p


def _degrees(rng, Np, No, Nc):
    """k_l = 2 + Geometric, clipped to Nc, adjusted so that sum == No exactly."""
    assert No >= 2 * Np, "need at least two observations per point"
    assert No <= Np * Nc, "more observations than (camera, point) pairs"
    mean_extra = No / Np - 2.0
    if mean_extra <= 0:
        k = np.full(Np, 2, np.int64)
    else:
        p = 1.0 / (1.0 + mean_extra)
        k = 2 + rng.geometric(p, Np).astype(np.int64) - 1
    k = np.minimum(k, Nc)
    diff = int(No - k.sum())
    # spread the remainder one observation at a time over random points
    while diff != 0:
        m = min(abs(diff), Np)
        sel = rng.permutation(Np)[:m]
        if diff > 0:
            ok = sel[k[sel] < Nc]
            k[ok] += 1
            diff -= len(ok)
        else:
            ok = sel[k[sel] > 2]
            k[ok] -= 1
            diff += len(ok)
    return k


def make_problem(Nc, Np, No, seed=0, window=32, noise_px=0.5, name="synthetic") -> BalProblem:
    """Deterministic BAL-shaped problem: truth -> noisy observations -> perturbed initial guess."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pts_true = rng.uniform(-1.0, 1.0, (Np, 3))
    cams_true = np.zeros((Nc, 9))
    cams_true[:, 0:3] = rng.normal(0, 0.05, (Nc, 3))
    cams_true[:, 3:5] = rng.normal(0, 0.1, (Nc, 2))
    cams_true[:, 5] = -5.0 + rng.normal(0, 0.1, Nc)
    cams_true[:, 6] = rng.uniform(500, 1500, Nc)
    cams_true[:, 7] = rng.normal(0, 0.01, Nc)
    cams_true[:, 8] = rng.normal(0, 0.001, Nc)

    k = _degrees(rng, Np, No, Nc)
    ptr = np.concatenate([[0], np.cumsum(k)])
    pt_idx = np.repeat(np.arange(Np, dtype=np.int64), k)
    # k distinct cameras inside a window of W cameras around a home camera:
    # sorted uniforms u_(j) -> floor(u_(j) * (W-k+1)) + j is strictly increasing in [0, W)
    W = np.maximum(np.minimum(window, Nc), k)
    home = rng.integers(0, Nc, Np)
    u = rng.random(No)
    order = np.lexsort((u, pt_idx))
    u = u[order]
    j = np.arange(No) - ptr[pt_idx]
    Wl = W[pt_idx]
    off = np.floor(u * (Wl - k[pt_idx] + 1)).astype(np.int64) + j
    cam_idx = (home[pt_idx] - Wl // 2 + off) % Nc
    # Wl <= Nc guarantees distinct cameras per point after the modulo

    obs = project(cams_true, pts_true, cam_idx, pt_idx) + rng.normal(0, noise_px, (No, 2))

    cams = cams_true.copy()
    cams[:, 0:6] += rng.normal(0, 0.001, (Nc, 6))
    cams[:, 6] *= 1 + rng.normal(0, 0.001, Nc)
    pts = pts_true + rng.normal(0, 0.01, (Np, 3))

    # shuffle the observation order like a real BAL file (grouped by nothing in particular)
    perm = rng.permutation(No)
    return BalProblem(cams, pts, obs[perm], cam_idx[perm].astype(np.int32), pt_idx[perm].astype(np.int32), name)


def make_config(name: str) -> BalProblem:
    Nc, Np, No, seed, window = CONFIGS[name]
    return make_problem(Nc, Np, No, seed=seed, window=window, name=name)


def make_circle(n=100, radius=4.0):
    """Start points of BASELINE configs[0] (the reference's examples/circle.cu:87-104 problem: n 2-d points near the circle
    |p| = radius, one unary factor each), as a FIXED function of the index instead of circle.cu's std::random_device: one point
    per quadrant in turn, within 0.3 rad of the diagonal, |p| in [radius - 0.45, radius + 0.45].  (Marquardt's diag(H) damping
    divides the step by 4 x^2 and 4 y^2: a start near an axis — the reference's random starts hit one now and then — sends that
    point off tangentially and the shared accept / reject decision stalls the whole graph; near the diagonals every run converges
    and its trace can be compared digit by digit.)  tests/cpp/test_generic_radius.hip carries the same formula as its default start;
    tests/test_generic_api.py hands these very bits to it and to oracle.circle_lm through a file."""
    i = np.arange(n, dtype=np.float64)
    ang = 0.25 * np.pi + 0.5 * np.pi * i + 0.3 * np.sin(2.1 * i + 0.4)
    rad = radius + 0.45 * np.sin(1.3 * i + 0.2)
    return np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)


def write_bal(path, prob: BalProblem):
    Nc, Np, No = prob.shape
    with open(path, "w") as f:
        f.write(f"{Nc} {Np} {No}\n")
        for c, p, (x, y) in zip(prob.cam_idx, prob.pt_idx, prob.obs):
            f.write(f"{int(c)} {int(p)} {x:.17g} {y:.17g}\n")
        for v in prob.cameras.ravel():
            f.write(f"{v:.17g}\n")
        for v in prob.points.ravel():
            f.write(f"{v:.17g}\n")


def read_bal(path) -> BalProblem:
    with open(path) as f:
        tok = f.read().split()
    Nc, Np, No = int(tok[0]), int(tok[1]), int(tok[2])
    o = np.array(tok[3:3 + 4 * No], dtype=np.float64).reshape(No, 4)
    base = 3 + 4 * No
    cams = np.array(tok[base:base + 9 * Nc], dtype=np.float64).reshape(Nc, 9)
    pts = np.array(tok[base + 9 * Nc:base + 9 * Nc + 3 * Np], dtype=np.float64).reshape(Np, 3)
    return BalProblem(cams, pts, o[:, 2:4].copy(), o[:, 0].astype(np.int32), o[:, 1].astype(np.int32), path)


def schur_test_fixture(dtype=np.float64) -> BalProblem:
    """The 2-camera / 3-point / 6-observation fixture of
    /root/reference/tests/schur.cu:35-79 (inputs only).  Point coordinates are
    float literals there (0.1f ...), so they are rounded through float32 first."""
    cams = np.array([[0.12, -0.08, 0.03, 0.25, -0.10, 0.20, 800.0, 0.01, -0.001],
                     [-0.09, 0.06, -0.04, -0.30, 0.14, -0.22, 820.0, -0.012, 0.0009]], dtype=np.float64)
    pts = np.array([[0.1, 0.0, 2.0], [-0.1, 0.05, 2.2], [0.0, -0.05, 1.8]], dtype=np.float32).astype(np.float64)
    if np.dtype(dtype) == np.float32:
        cams = cams.astype(np.float32).astype(np.float64)
    obs = np.zeros((6, 2))
    cam_idx = np.array([0, 1, 0, 1, 0, 1], np.int32)
    pt_idx = np.array([0, 0, 1, 1, 2, 2], np.int32)
    return BalProblem(cams, pts, obs, cam_idx, pt_idx, "schur-2x3")


def make_pose_graph(n_poses=2000, factors_per_pose=5, seed=7, sigma_t=0.02, sigma_th=0.005, huber=False):
    """A planar pose graph (SE(2) between-factors: the SLAM back-end workload the reference's README names, and what BASELINE
    configs[0] calls a "2D pose-graph"): a wandering trajectory, for every pose i the odometry factor (i, i + 1), short-range
    factors (i, i + 2), (i, i + 3) and loop closures to earlier poses that lie nearby, `factors_per_pose` per pose on average.
    Returns (poses0 [n, 3] initial guess from integrated noisy odometry, fixed [n] (pose 0), edges [F, 2] int32, meas [F, 3],
    info [F, 3, 3] per-factor information matrices (symmetric positive definite, not diagonal), truth [n, 3])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = n_poses
    # ground truth: constant forward speed, slowly varying turn rate (loops back on itself)
    th = np.cumsum(0.12 * np.sin(0.013 * np.arange(n)) + 0.05 + rng.normal(0, 0.01, n))
    step = 0.5
    xy = np.cumsum(np.stack([step * np.cos(th), step * np.sin(th)], 1), 0)
    truth = np.concatenate([xy, th[:, None]], 1)

    def rel(a, b):  # measurement of b in a's frame (the factor's error at the truth is zero)
        c, s = np.cos(a[:, 2]), np.sin(a[:, 2])
        dx, dy = b[:, 0] - a[:, 0], b[:, 1] - a[:, 1]
        return np.stack([c * dx + s * dy, -s * dx + c * dy, b[:, 2] - a[:, 2]], 1)

    ii = [np.arange(n - 1)]; jj = [np.arange(1, n)]
    for d in (2, 3):
        ii.append(np.arange(n - d)); jj.append(np.arange(d, n))
    target = int(factors_per_pose * n)
    have = sum(len(a) for a in ii)
    if target > have:  # loop closures: a random earlier pose within a window of 40 steps in space order
        order = np.argsort(xy[:, 0] + 1e-3 * xy[:, 1])
        k = target - have
        a = rng.integers(0, n, k)
        pos = np.empty(n, np.int64); pos[order] = np.arange(n)
        b = order[np.clip(pos[a] + rng.integers(-20, 21, k), 0, n - 1)]
        keep = np.abs(a - b) > 3
        ii.append(np.minimum(a, b)[keep]); jj.append(np.maximum(a, b)[keep])
    i = np.concatenate(ii).astype(np.int32); j = np.concatenate(jj).astype(np.int32)
    F = len(i)
    meas = rel(truth[i], truth[j]) + rng.normal(0, 1, (F, 3)) * np.array([sigma_t, sigma_t, sigma_th])
    # information: R diag(1 / sigma^2) R^T with a small random rotation of the (x, y) axes and an x-theta coupling: symmetric, SPD, full
    w = np.array([1 / sigma_t ** 2, 1 / sigma_t ** 2, 1 / sigma_th ** 2]) * 1e-3  # (scaled: chi2 of order F)
    ang = rng.uniform(-0.3, 0.3, F)
    info = np.zeros((F, 3, 3))
    c, s = np.cos(ang), np.sin(ang)
    sx, sy = w[0] * rng.uniform(0.5, 1.5, F), w[1] * rng.uniform(0.5, 1.5, F)
    info[:, 0, 0] = c * c * sx + s * s * sy; info[:, 1, 1] = s * s * sx + c * c * sy
    info[:, 0, 1] = info[:, 1, 0] = c * s * (sx - sy)
    info[:, 2, 2] = w[2] * rng.uniform(0.5, 1.5, F)
    cpl = 0.1 * np.sqrt(info[:, 0, 0] * info[:, 2, 2]) * rng.uniform(-1, 1, F)
    info[:, 0, 2] = info[:, 2, 0] = cpl
    # initial guess: the odometry chain integrated with its noise (drifts away from the truth)
    poses0 = np.zeros((n, 3)); poses0[0] = truth[0]
    odo = meas[: n - 1]
    for k_ in range(n - 1):
        c0, s0 = np.cos(poses0[k_, 2]), np.sin(poses0[k_, 2])
        poses0[k_ + 1] = [poses0[k_, 0] + c0 * odo[k_, 0] - s0 * odo[k_, 1], poses0[k_, 1] + s0 * odo[k_, 0] + c0 * odo[k_, 1], poses0[k_, 2] + odo[k_, 2]]
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    return poses0, fixed, np.stack([i, j], 1), meas, info, truth


def write_pose_graph(path, poses, fixed, edges, meas, info, huber_delta=0.0, priors=None):
    """text file read by tests/cpp/test_pose_graph.hip: 'N F delta', N lines 'x y theta fixed', F lines 'i j mx my mth p00 .. p22' (row-major);
    priors = (vertex ids, measurements [n, 3], information matrices [n, 3, 3]): then 'PRIORS n' and n lines 'i mx my mth p00 .. p22' — unary
    factors x_i - m of a second factor descriptor"""
    with open(path, "w") as f:
        f.write(f"{len(poses)} {len(edges)} {float(huber_delta)!r}\n")
        for p, fx in zip(poses, fixed):
            f.write(f"{float(p[0])!r} {float(p[1])!r} {float(p[2])!r} {int(fx)}\n")
        for e, m, P in zip(edges, meas, info):
            f.write(f"{int(e[0])} {int(e[1])} " + " ".join(repr(float(v)) for v in m) + " " + " ".join(repr(float(v)) for v in P.ravel()) + "\n")
        if priors is not None:
            f.write(f"PRIORS {len(priors[0])}\n")
            for i, m, P in zip(priors[0], priors[1], priors[2]):
                f.write(f"{int(i)} " + " ".join(repr(float(v)) for v in m) + " " + " ".join(repr(float(v)) for v in np.asarray(P).ravel()) + "\n")
