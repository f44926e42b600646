"""Landmark-sharded (multi-GPU) algorithm on a 1-GPU box.

Several shards of one problem run concurrently in ONE process (one host thread per shard, the
in-process test communicator of csrc/comm.hpp) and must reproduce the unsharded solve; RCCL
itself is exercised with a 1-rank communicator.  The 8-GPU run is the driver's."""
import ctypes as C
import threading

import numpy as np
import pytest

import graphite_amd as ga
from graphite_amd import dist as gdist, synth, _lib

pytestmark = pytest.mark.gpu


def run_sharded(prob, world, dtype, iterations, solver):
    shards = [gdist.partition_by_landmark(prob, r, world) for r in range(world)]
    engines = [ga.BalProblem(s.cameras, s.points, s.obs, s.cam_idx, s.pt_idx, dtype=dtype, shard=True) for s in shards]
    gdist.init_local_group(engines)
    out = [None] * world
    err = []

    def work(r):
        try:
            out[r] = engines[r].levenberg_marquardt(solver=solver, iterations=iterations)
        except Exception as e:  # pragma: no cover
            err.append(e)

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not err, err
    assert all(o is not None for o in out), "a shard thread hung"
    cams = [e.get_params()[0] for e in engines]
    pts = gdist.assemble_points(shards, [e.get_params()[1] for e in engines])
    [e.close() for e in engines]
    return out, cams, pts


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("solver", [ga.SOLVER_PCG, ga.SOLVER_PCG_IDENTITY, ga.SOLVER_PCG_SCHUR_IMPLICIT])
def test_sharded_lm_matches_single(world, solver):
    prob = synth.make_config("mini-50")
    single = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ct, lt, st = single.levenberg_marquardt(solver=solver, iterations=8)
    c1, p1 = single.get_params()
    single.close()
    out, cams, pts = run_sharded(prob, world, np.float64, 8, solver)
    for r in range(world):
        ct_r, lt_r, st_r = out[r]
        assert np.allclose(ct_r, ct, rtol=1e-9), (r, ct_r, ct)       # every rank sees the global chi2
        assert np.allclose(lt_r, lt, rtol=1e-6)
        assert st_r["pcg_iterations"] == st["pcg_iterations"]
        assert np.array_equal(cams[r], cams[0])                        # replicated cameras stay bit-identical
        # collectives (gr_lm_stats.collectives): per inner iteration ONE grouped all-reduce, per LM iteration at most the
        # grouped Hcc/bc/chi2 of the linearisation, the trial chi2, and the one that closes / detects the end of the solve
        # (every PCG flavour: ONE per inner iteration — the matrix-free solvers run the single-reduction recurrence on shards,
        # camera rows and all dot products of an iteration in one message; the implicit Schur solver its one camera vector)
        lm_its = len(ct_r) - 1
        assert st_r["pcg_iterations"] <= st_r["collectives"] <= st_r["pcg_iterations"] + 4 * (lm_its + 1), st_r
    assert np.allclose(cams[0], c1, rtol=1e-7, atol=1e-10)
    assert np.allclose(pts, p1, rtol=1e-7, atol=1e-10)


def test_sharded_ladybug49_fp32():
    prob = synth.make_config("ladybug-49")
    single = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float32)
    ct, _, _ = single.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=5)
    single.close()
    out, cams, _ = run_sharded(prob, 4, np.float32, 5, ga.SOLVER_PCG)
    assert np.allclose(out[0][0], ct, rtol=2e-3)
    assert all(np.array_equal(c, cams[0]) for c in cams)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("solver", [ga.SOLVER_PCG_SCHUR, ga.SOLVER_DENSE_SCHUR])
def test_sharded_explicit_schur_matches_single(world, solver):
    """S and b_S are reduced over each shard's points and all-reduced; the replicated reduced solve
    applies rank 0's camera step everywhere."""
    prob = synth.make_config("mini-50")
    single = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ct, lt, st = single.levenberg_marquardt(solver=solver, iterations=8)
    c1, p1 = single.get_params()
    single.close()
    out, cams, pts = run_sharded(prob, world, np.float64, 8, solver)
    for r in range(world):
        ct_r, lt_r, st_r = out[r]
        assert np.allclose(ct_r, ct, rtol=1e-9), (r, ct_r, ct)
        assert np.allclose(lt_r, lt, rtol=1e-6)
        assert np.array_equal(cams[r], cams[0])
    assert out[0][2]["pcg_iterations"] == st["pcg_iterations"]
    assert np.allclose(cams[0], c1, rtol=1e-7, atol=1e-10)
    assert np.allclose(pts, p1, rtol=1e-7, atol=1e-10)


def _schur_blocks(e, mu):
    e.solver_update_structure(ga.SOLVER_PCG_SCHUR)
    e.linearize()
    e.solver_update_values(ga.SOLVER_PCG_SCHUR)
    e.solver_set_damping(ga.SOLVER_PCG_SCHUR, mu)
    e.schur_update_values()
    cp, ri = e.schur_structure()
    return cp, ri, e.get("S"), e.get("b_schur")


def test_sharded_schur_complement_is_global():
    """gr_bal_schur_update_values on shards: every rank ends with the S, b_S and block structure of the
    unsharded problem (a shard alone only sees the co-observations of its own points)."""
    prob = synth.make_problem(9, 300, 2400, seed=21)
    single = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    colptr, rowidx, S1, b1 = _schur_blocks(single, 1e-3)
    single.close()
    world = 3
    shards = [gdist.partition_by_landmark(prob, r, world) for r in range(world)]
    engines = [ga.BalProblem(s.cameras, s.points, s.obs, s.cam_idx, s.pt_idx, dtype=np.float64, shard=True) for s in shards]
    gdist.init_local_group(engines)
    res = [None] * world
    err = []

    def work(r):
        try:
            res[r] = _schur_blocks(engines[r], 1e-3)
        except Exception as ex:  # pragma: no cover
            err.append(ex)

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not err, err
    for r in range(world):
        cp, ri, S, b = res[r]
        assert np.array_equal(cp, colptr) and np.array_equal(ri, rowidx)
        assert np.allclose(S, S1, rtol=1e-10, atol=1e-12 * np.abs(S1).max())
        assert np.allclose(b, b1, rtol=1e-10, atol=1e-12 * np.abs(b1).max())
    [e.close() for e in engines]


def test_rccl_single_rank_communicator():
    """RCCL code path (dlopen, ncclCommInitRank, grouped all-reduces) with world_size = 1."""
    lib = _lib.lib()
    prob = synth.make_config("mini-50")
    ref = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    ct, _, _ = ref.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=6)
    ref.close()
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64, shard=True)
    uid = (C.c_char * 128)()
    _lib.check(lib.gr_comm_unique_id(uid))
    _lib.check(lib.gr_bal_comm_init(gpu.h, uid, C.c_int(0), C.c_int(1)))
    ct1, _, _ = gpu.levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=6)
    gpu.close()
    assert np.allclose(ct1, ct, rtol=1e-10)


# boundary shapes of the kernels' tilings, cut into 2 and 3 landmark shards (a shard may leave cameras without
# any observation, end inside a wave, or hold a single point): (Nc, Np, No, window, seed)
SHARD_SHAPES = [(3, 30, 65, 3, 3), (5, 60, 257, 5, 7), (29, 300, 1025, 8, 9), (70, 40, 2000, 70, 10), (64, 5000, 12000, 4, 13)]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("shape", SHARD_SHAPES, ids=lambda s: "x".join(map(str, s[:3])))
def test_sharded_boundary_shapes_all_solvers(shape, world):
    Nc, Np, No, window, seed = shape
    prob = synth.make_problem(Nc, Np, No, seed=seed, window=window)
    for solver in (ga.SOLVER_PCG, ga.SOLVER_PCG_SCHUR_IMPLICIT, ga.SOLVER_PCG_SCHUR, ga.SOLVER_DENSE_SCHUR):
        single = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
        ct, lt, st = single.levenberg_marquardt(solver=solver, iterations=6)
        c1, p1 = single.get_params()
        single.close()
        out, cams, pts = run_sharded(prob, world, np.float64, 6, solver)
        for r in range(world):
            assert len(out[r][0]) == len(ct), (solver, r)
            assert np.allclose(out[r][0], ct, rtol=1e-8), (solver, r, out[r][0], ct)
            assert np.array_equal(cams[r], cams[0])
        assert np.allclose(cams[0], c1, rtol=1e-6, atol=1e-9) and np.allclose(pts, p1, rtol=1e-6, atol=1e-9), solver


@pytest.mark.parametrize("solver", [ga.SOLVER_PCG_SCHUR_IMPLICIT, ga.SOLVER_PCG])
def test_sharded_many_cameras_replicas_stay_identical(solver):
    """4 000 cameras = 143 blocks of the 252-scalar camera kernels, i.e. more blocks than the 64 atomic slots of
    the matrix-free dots: the implicit-Schur scalars are per-block partials summed in a fixed order, so the
    replicated camera vectors, the loop decisions (one collective per inner iteration hangs if ranks disagree)
    and the iteration counts are identical on every rank."""
    prob = synth.make_problem(4000, 30000, 140000, seed=31, window=64)
    out, cams, _ = run_sharded(prob, 3, np.float64, 4, solver)
    for r in range(3):
        assert np.array_equal(cams[r], cams[0])
        assert out[r][2]["pcg_iterations"] == out[0][2]["pcg_iterations"]
        assert np.array_equal(out[r][0], out[0][0]) and np.array_equal(out[r][1], out[0][1])
        assert out[r][2]["collectives"] == out[0][2]["collectives"] > 0


def test_sharded_implicit_schur_solve_is_reproducible():
    """same solve twice on the same engine: identical bits (no atomics on the Schur-PCG path)."""
    prob = synth.make_problem(4000, 30000, 140000, seed=31, window=64)
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    g.solver_update_structure(ga.SOLVER_PCG_SCHUR_IMPLICIT)
    g.linearize()
    g.solver_update_values(ga.SOLVER_PCG_SCHUR_IMPLICIT)
    g.solver_set_damping(ga.SOLVER_PCG_SCHUR_IMPLICIT, 1e-4)
    a, ia = g.solver_solve(ga.SOLVER_PCG_SCHUR_IMPLICIT, max_iter=8, tol=0.0, rej=1e9)
    b, ib = g.solver_solve(ga.SOLVER_PCG_SCHUR_IMPLICIT, max_iter=8, tol=0.0, rej=1e9)
    assert ia == ib and np.array_equal(a, b)
    g.close()
