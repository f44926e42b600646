"""GPU parity: every stage of the HIP hot path (through the C-ABI) against the CPU oracle
on the same seeded inputs.  fp64 bar: 1e-9 relative on assembled quantities (the
reference holds itself to 1e-12 on the 2x3 fixture, tests/schur.cu:180-239, asserted
separately below); fp32 bar: stated per test."""
import numpy as np
import pytest

import graphite_amd as ga
from graphite_amd import synth

pytestmark = pytest.mark.gpu

CASES = [("mini-6", np.float64), ("mini-50", np.float64), ("mini-50", np.float32), ("ladybug-49", np.float32),
         ("ladybug-49", np.float64)]


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def make_pair(oracle_mod, name, dtype):
    prob = synth.schur_test_fixture(dtype) if name == "schur-2x3" else synth.make_config(name)
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    return prob, gpu, ref


def tol_for(dtype, t64, t32):
    return t64 if np.dtype(dtype) == np.float64 else t32


@pytest.mark.parametrize("name,dtype", CASES)
def test_linearize_and_hessian(oracle_mod, name, dtype):
    prob, gpu, ref = make_pair(oracle_mod, name, dtype)
    gpu.solver_update_structure(ga.SOLVER_PCG_SCHUR)
    gpu.linearize()
    ref.linearize()
    ref.hessian_update()
    chi2_g, chi2_r = gpu.chi2(), ref.chi2()
    assert abs(chi2_g - chi2_r) / chi2_r < tol_for(dtype, 1e-12, 1e-5)
    assert relerr(gpu.get("residuals"), ref.get("res")) < tol_for(dtype, 1e-12, 1e-4)
    assert relerr(gpu.get("scales"), ref.get("scales")) < tol_for(dtype, 1e-11, 1e-4)
    # b: fp32 sums of thousands of terms in a different order
    # fp32 (measured against the fp64 oracle, tools/fp32_stage_probe.py): b 1.3e-5, Hcc 2e-7, Hll 4e-7, Hcp 5e-7 — bars 10x above
    assert relerr(gpu.get("b"), ref.get("b")) < tol_for(dtype, 1e-10, 2e-4)
    assert relerr(gpu.get("Hcc"), ref.get("Hcc")) < tol_for(dtype, 1e-10, 2e-5)
    assert relerr(gpu.get("Hll"), ref.get("Hll")) < tol_for(dtype, 1e-10, 1e-5)
    assert relerr(gpu.get("Hcp"), ref.get("Hcp")) < tol_for(dtype, 1e-10, 1e-5)
    gpu.close()


@pytest.mark.parametrize("name,dtype", [("schur-2x3", np.float64)] + CASES)
def test_schur_complement(oracle_mod, name, dtype):
    """S, b_S, Hll^-1, landmark back-substitution and S*x (tests/schur.cu:113-240)."""
    prob, gpu, ref = make_pair(oracle_mod, name, dtype)
    mu = 0.0 if name == "schur-2x3" else 1e-4  # the reference's Schur test runs undamped
    gpu.solver_update_structure(ga.SOLVER_PCG_SCHUR)
    gpu.linearize()
    gpu.solver_update_values(ga.SOLVER_PCG_SCHUR)
    gpu.solver_set_damping(ga.SOLVER_PCG_SCHUR, mu)
    gpu.schur_update_values()
    ref.linearize()
    ref.hessian_update()
    ref.apply_damping(mu)
    ref.schur_update()
    cp_g, ri_g = gpu.schur_structure()
    cp_r, ri_r = ref.schur_structure()
    assert np.array_equal(cp_g, cp_r) and np.array_equal(ri_g, ri_r)  # index work: bit-exact
    t = tol_for(dtype, 1e-12 if name == "schur-2x3" else 1e-9, 5e-3)
    # fp32 (round 5: the per-point chain scale -> block -> inverse -> M' runs in double from the stored sums): S and b_S at 1e-4
    # (measured 1.9e-5 / 4.2e-5 against the fp64 oracle, where the fp32 oracle itself sits at 1.1e-5 / 4.1e-5).  Hll^-1 is the
    # inverse of 3 x 3 blocks with condition numbers of 1e3-1e4: the fp32 ORACLE is 8.3e-4 from the fp64 one, the engine 6.6e-4
    # (tools/fp32_stage_probe.py), so two fp32 computations cannot be asked to agree below ~2e-3
    t_s = tol_for(dtype, 1e-12 if name == "schur-2x3" else 1e-9, 1e-4)
    t_inv = tol_for(dtype, 1e-12 if name == "schur-2x3" else 1e-9, 2e-3)
    assert relerr(gpu.get("Hll_inv"), ref.get("Hll_inv")) < t_inv
    assert relerr(gpu.get("S"), ref.get("S")) < t_s
    assert relerr(gpu.get("b_schur"), ref.get("b_schur")) < t_s
    xp = (0.01 * np.arange(1, 9 * gpu.Nc + 1)).astype(dtype)  # tests/schur.cu:211-214
    assert relerr(gpu.landmark_update(xp), ref.landmark_update(xp)) < t
    assert relerr(gpu.schur_matvec(xp), ref.schur_matvec(xp)) < t
    gpu.close()


@pytest.mark.parametrize("solver", ["pcg_schur", "pcg_schur_implicit", "pcg", "pcg_identity"])
@pytest.mark.parametrize("name,dtype", [("mini-50", np.float64), ("mini-50", np.float32), ("ladybug-49", np.float64)])
def test_solver_solve(oracle_mod, name, dtype, solver):
    prob, gpu, ref = make_pair(oracle_mod, name, dtype)
    gs = dict(pcg_schur=ga.SOLVER_PCG_SCHUR, pcg=ga.SOLVER_PCG, pcg_identity=ga.SOLVER_PCG_IDENTITY,
              pcg_schur_implicit=ga.SOLVER_PCG_SCHUR_IMPLICIT)[solver]
    os_ = dict(pcg_schur=oracle_mod.SOLVER_PCG_SCHUR, pcg=oracle_mod.SOLVER_PCG,
               pcg_identity=oracle_mod.SOLVER_PCG_IDENTITY, pcg_schur_implicit=oracle_mod.SOLVER_PCG_SCHUR)[solver]
    gpu.solver_update_structure(gs)
    gpu.linearize()
    gpu.solver_update_values(gs)
    gpu.solver_set_damping(gs, 1e-4)
    ref.linearize()
    ref.solver_update_values(os_)
    ref.solver_set_damping(os_, 1e-4)
    for max_iter, tol in ((4, 0.0), (25, 1e-12)):
        dx_g, it_g = gpu.solver_solve(gs, max_iter=max_iter, tol=tol, rej=1e6)
        dx_r, it_r = ref.solver_solve(os_, max_iter=max_iter, tol=tol, rej=1e6)
        assert it_g == it_r
        # PCG amplifies rounding differences with the iteration count
        # measured on MI355X: fp64 <= 5e-11 (25 identity-preconditioned iterations), fp32 <= 9e-4
        # fp32: the bar is the conditioning of the block-Jacobi / Schur blocks, not the assembly — tools/fp32_stage_probe.py and
        # tools/fp32_dx_probe.py against the fp64 oracle on this problem: every assembled quantity (Hcc, Hll, Hcp, b, scales) is
        # as close to fp64 as the fp32 oracle's or closer (Hcc 2e-7 vs 8e-7), yet after ONE inner iteration the engine's step
        # sits 8e-4 from the fp64 step and the fp32 oracle's 3e-4 (Hll^-1 alone: 1.2e-3 vs 8e-4): two fp32 runs of the same
        # algorithm cannot agree better than either agrees with fp64.  1e-3 for the short solve (was 2e-3), 2e-3 for 25 iterations.
        # round 5 (per-point block algebra and sums in double): 7e-4 for the short solve (measured <= 5.4e-4), 2e-3 for 25 iterations
        assert relerr(dx_g, dx_r) < tol_for(dtype, 1e-9, 7e-4 if max_iter <= 4 else 2e-3)
    gpu.close()


@pytest.mark.parametrize("solver", ["pcg_schur", "pcg"])
def test_fp32_step_is_as_close_to_fp64_as_the_fp32_oracle(oracle_mod, solver):
    """The fp32 step pinned from both sides: two fp32 runs of one algorithm that sum in different orders
    cannot agree better than either agrees with fp64 (block conditioning: Hll^-1 alone is 1e-3 from its fp64 value), so the
    engine's fp32 step is held (a) to the FP64 oracle at 5e-4 and (b) to the fp32 ORACLE's own distance from fp64: at most
    2 x that distance for the Schur PCG (measured 3.7e-4 against the fp32 oracle's 2.4e-4 = 1.5 x; round 4: 3.6 x) and 3.5 x for the
    matrix-free PCG (3.0e-4 against 1.05e-4 = 2.8 x; round 4: 5.6 x).  Found with tools/fp32_first_iteration_probe.py: all of the
    loss sat in the POINT part of the step — the 3 x 3 point blocks (condition 1e3-1e4) were scaled, damped and inverted from
    entries that had been rounded to fp32 three times; the chain now runs in double from the stored sums (scale_hat, k_finalize_bj,
    k_point_prepare) and the per-point sums of the update kernel are taken in double.  After ONE iteration the engine is 1.4 x
    the fp32 oracle's distance; what is left after four is the fp32 operator itself (J recomputed per product)."""
    prob = synth.make_config("mini-50")
    gs = dict(pcg_schur=ga.SOLVER_PCG_SCHUR, pcg=ga.SOLVER_PCG)[solver]
    os_ = dict(pcg_schur=oracle_mod.SOLVER_PCG_SCHUR, pcg=oracle_mod.SOLVER_PCG)[solver]
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float32)
    gpu.solver_update_structure(gs)
    gpu.linearize()
    gpu.solver_update_values(gs)
    gpu.solver_set_damping(gs, 1e-4)
    dx_g, it_g = gpu.solver_solve(gs, max_iter=4, tol=0.0, rej=1e6)
    dx = {}
    for dt in (np.float32, np.float64):
        ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
        ref.linearize()
        ref.solver_update_values(os_)
        ref.solver_set_damping(os_, 1e-4)
        dx[dt], it_r = ref.solver_solve(os_, max_iter=4, tol=0.0, rej=1e6)
        assert it_r == it_g
    e_gpu, e_ref = relerr(dx_g, dx[np.float64]), relerr(dx[np.float32], dx[np.float64])
    assert e_gpu < 5e-4, (e_gpu, e_ref)
    assert e_gpu < (2.0 if solver == "pcg_schur" else 3.5) * e_ref + 1e-5, (e_gpu, e_ref)
    gpu.close()


@pytest.mark.parametrize("name", ["mini-50", "ladybug-49"])
def test_fp32_pcg_step_distance_per_inner_iteration(oracle_mod, name):
    """The same comparison in the 2-norm and for every inner-iteration count 1 .. 6 (tools/fp32_iter_probe.py prints the table): the
    single-sample maximum-norm ratio above is set by the worst entry of one step; over the whole step the engine's fp32 matrix-free PCG
    is 0.98-1.59 x the fp32 oracle's distance from fp64 on mini-50 and 0.65-0.94 x on Ladybug-49 (closer than the restatement: tree sums
    instead of sequential ones) — held at 2 x for every count, the bar VERDICT r4 asked for."""
    prob = synth.make_config(name)
    def dist(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float32)
    gpu.solver_update_structure(ga.SOLVER_PCG)
    refs = {}
    for dt in (np.float32, np.float64):
        refs[dt] = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
    worst = 0.0
    for k in range(1, 7):
        gpu.linearize()
        gpu.solver_update_values(ga.SOLVER_PCG)
        gpu.solver_set_damping(ga.SOLVER_PCG, 1e-4)
        dx_g, _ = gpu.solver_solve(ga.SOLVER_PCG, max_iter=k, tol=0.0, rej=1e6)
        dx = {}
        for dt, ref in refs.items():
            ref.linearize()
            ref.solver_update_values(oracle_mod.SOLVER_PCG)
            ref.solver_set_damping(oracle_mod.SOLVER_PCG, 1e-4)
            dx[dt], _ = ref.solver_solve(oracle_mod.SOLVER_PCG, max_iter=k, tol=0.0, rej=1e6)
        e_gpu, e_ref = dist(dx_g, dx[np.float64]), dist(dx[np.float32], dx[np.float64])
        worst = max(worst, e_gpu / e_ref)
        assert e_gpu < 2.0 * e_ref, (k, e_gpu, e_ref)
        assert e_gpu < 2e-4, (k, e_gpu)
    gpu.close()


def test_pcg_schur_matches_direct_solve(oracle_mod):
    """tests/schur.cu:340-389: PCG-Schur (512 it, tol 1e-14, rejection 1e6) vs the direct Schur
    LDLT solve, mu = 1e-4, |delta| < 5e-4 on the 2x3 fixture."""
    prob, gpu, ref = make_pair(oracle_mod, "schur-2x3", np.float64)
    gpu.solver_update_structure(ga.SOLVER_PCG_SCHUR)
    gpu.linearize()
    gpu.solver_update_values(ga.SOLVER_PCG_SCHUR)
    gpu.solver_set_damping(ga.SOLVER_PCG_SCHUR, 1e-4)
    dx_g, _ = gpu.solver_solve(ga.SOLVER_PCG_SCHUR, max_iter=512, tol=1e-14, rej=1e6)
    ref.linearize()
    ref.solver_update_values(oracle_mod.SOLVER_LDLT_SCHUR)
    ref.solver_set_damping(oracle_mod.SOLVER_LDLT_SCHUR, 1e-4)
    dx_r, _ = ref.solver_solve(oracle_mod.SOLVER_LDLT_SCHUR)
    assert np.abs(dx_g - dx_r).max() < 5e-4
    gpu.close()


@pytest.mark.parametrize("solver", ["pcg_schur", "pcg_schur_implicit", "pcg"])
@pytest.mark.parametrize("name,dtype,rtol", [("mini-50", np.float64, 1e-9), ("ladybug-49", np.float64, 1e-8),
                                             ("ladybug-49", np.float32, 1e-5)])
def test_levenberg_marquardt_trace(oracle_mod, name, dtype, rtol, solver):
    """Whole LM loop: chi2 and lambda traces against the oracle (north star: 1e-6 relative in fp64)."""
    prob, gpu, ref = make_pair(oracle_mod, name, dtype)
    gs = dict(pcg_schur=ga.SOLVER_PCG_SCHUR, pcg=ga.SOLVER_PCG, pcg_schur_implicit=ga.SOLVER_PCG_SCHUR_IMPLICIT)[solver]
    os_ = dict(pcg_schur=oracle_mod.SOLVER_PCG_SCHUR, pcg=oracle_mod.SOLVER_PCG,
               pcg_schur_implicit=oracle_mod.SOLVER_PCG_SCHUR)[solver]
    ct_g, lt_g, st = gpu.levenberg_marquardt(solver=gs, iterations=8)
    ct_r, lt_r, st_r = ref.levenberg_marquardt(solver=os_, iterations=8)
    if np.dtype(dtype) == np.float32:
        # a converged fp32 run stops on rho == 0 (levenberg_marquardt.hpp:229) at a rounding-dependent trip,
        # in the oracle as on the GPU: compare the common prefix
        m = min(len(ct_g), len(ct_r))
        assert m >= 4
        ct_g, lt_g, ct_r, lt_r = ct_g[:m], lt_g[:m], ct_r[:m], lt_r[:m]
    assert len(ct_g) == len(ct_r)
    assert np.abs(ct_g - ct_r).max() / ct_r.max() < rtol
    assert np.allclose(ct_g, ct_r, rtol=max(rtol, 1e-6))
    if np.dtype(dtype) == np.float64:
        assert np.allclose(lt_g, lt_r, rtol=1e-3)
    else:
        # fp32: once chi2 has converged to ~1e-6 relative the accept/reject decision is rounding
        # noise (in the oracle as well); compare lambda only while chi2 still moves by > 1e-4
        moving = np.abs(np.diff(ct_r)) / ct_r[:-1] > 1e-4
        k = int(np.argmin(moving)) if not moving.all() else len(moving)
        assert np.allclose(lt_g[:k + 1], lt_r[:k + 1], rtol=1e-3)
    assert ct_g[-1] < 0.1 * ct_g[0]
    if np.dtype(dtype) == np.float64:
        assert st["pcg_iterations"] == st_r["pcg_iterations"]
    if np.dtype(dtype) == np.float64:  # fp32 parameters random-walk at the 1e-3 level once converged
        cg, pg = gpu.get_params()
        cr, pr = ref.get_params()
        assert relerr(cg, cr) < max(rtol, 1e-6) and relerr(pg, pr) < max(rtol, 1e-6)
    gpu.close()


def test_huber_loss(oracle_mod):
    prob, gpu, ref = make_pair(oracle_mod, "mini-50", np.float64)
    gpu.set_loss(ga.LOSS_HUBER, 1.0)
    ref.set_loss(oracle_mod.LOSS_HUBER, 1.0)
    ct_g, _, _ = gpu.levenberg_marquardt(solver=ga.SOLVER_PCG_SCHUR, iterations=5)
    ct_r, _, _ = ref.levenberg_marquardt(solver=oracle_mod.SOLVER_PCG_SCHUR, iterations=5)
    assert np.allclose(ct_g, ct_r, rtol=1e-8)
    gpu.close()


def test_apply_update_backup_revert(oracle_mod):
    """tests/vertex.cu:76-119 (v + delta*scale) and :299-341 (backup/restore)."""
    prob, gpu, ref = make_pair(oracle_mod, "mini-6", np.float64)
    gpu.linearize()
    s = gpu.get("scales")
    dx = np.linspace(-1e-3, 1e-3, gpu.n)
    gpu.backup_parameters()
    gpu.apply_update(dx)
    c, p = gpu.get_params()
    assert np.allclose(c.ravel(), prob.cameras.ravel() + dx[:9 * gpu.Nc] * s[:9 * gpu.Nc], rtol=0, atol=1e-15)
    assert np.allclose(p.ravel(), prob.points.ravel() + dx[9 * gpu.Nc:] * s[9 * gpu.Nc:], rtol=0, atol=1e-15)
    gpu.revert_parameters()
    c, p = gpu.get_params()
    assert np.array_equal(c, prob.cameras) and np.array_equal(p, prob.points)
    gpu.close()


def test_duplicate_edge_is_rejected():
    prob = synth.make_config("mini-6")
    ci = prob.cam_idx.copy()
    pi = prob.pt_idx.copy()
    ci[1], pi[1] = ci[0], pi[0]
    with pytest.raises(ga._lib.GraphiteError) as e:
        ga.BalProblem(prob.cameras, prob.points, prob.obs, ci, pi)
    assert e.value.status == 4


def test_against_committed_golden_fixtures():
    """HIP path vs tests/golden/*.npz (oracle regression pins, see tests/golden/make_golden.py)."""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gold, "schur_2x3_f64.npz"))
    gpu = ga.BalProblem(g["cameras"], g["points"], g["obs"], g["cam_idx"], g["pt_idx"], dtype=np.float64)
    gpu.solver_update_structure(ga.SOLVER_PCG_SCHUR)
    gpu.linearize()
    gpu.solver_update_values(ga.SOLVER_PCG_SCHUR)
    gpu.solver_set_damping(ga.SOLVER_PCG_SCHUR, 0.0)
    gpu.schur_update_values()
    for name, key in (("residuals", "res"), ("scales", "scales"), ("b", "b"), ("Hcc", "Hcc"), ("Hcp", "Hcp"),
                      ("Hll", "Hll"), ("S", "S"), ("b_schur", "b_schur")):
        assert relerr(gpu.get(name), g[key]) < 1e-12, name
    gpu.close()
    g = np.load(os.path.join(gold, "mini50_lm_f64.npz"))
    for solver, key in ((ga.SOLVER_PCG_SCHUR, "chi2_pcg_schur"), (ga.SOLVER_PCG, "chi2_pcg")):
        gpu = ga.BalProblem(g["cameras"], g["points"], g["obs"], g["cam_idx"], g["pt_idx"], dtype=np.float64)
        ct, _, _ = gpu.levenberg_marquardt(solver=solver, iterations=8)
        assert np.allclose(ct, g[key], rtol=1e-8), key
        gpu.close()


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("name,dtype,rtol", [("mini-50", np.float64, 1e-9), ("ladybug-49", np.float64, 1e-8)])
def test_levenberg_marquardt_fused_iteration(oracle_mod, name, dtype, rtol, fused):
    """gr_bal_tuning.lm_fused (default 1): iteration head in one launch with the accept decision taken on the device, trial step
    applied by the direction launch that ends the PCG loop; 0: the host-decided loop.  Same chi2 / lambda trace as the oracle,
    set through the tuning struct (not the environment)."""
    prob, gpu, ref = make_pair(oracle_mod, name, dtype)
    gpu.set_tuning(lm_fused=fused, pcg_lazy=0)
    assert gpu.get_tuning()["lm_fused"] == fused
    for solver, osolver in ((ga.SOLVER_PCG, oracle_mod.SOLVER_PCG), (ga.SOLVER_PCG_IDENTITY, oracle_mod.SOLVER_PCG_IDENTITY)):
        gpu.set_params(prob.cameras, prob.points)
        ref.set_params(prob.cameras, prob.points)
        ct_g, lt_g, st = gpu.levenberg_marquardt(solver=solver, iterations=10)
        ct_r, lt_r, st_r = ref.levenberg_marquardt(solver=osolver, iterations=10)
        assert len(ct_g) == len(ct_r)
        assert np.allclose(ct_g, ct_r, rtol=rtol)
        assert np.allclose(lt_g, lt_r, rtol=1e-6)
        assert st["pcg_iterations"] == st_r["pcg_iterations"]
        assert st["iterations_run"] == st_r["iterations_run"] and st["accepted"] == st_r["accepted"]
        # the state left behind is the accepted point and ITS linearisation (no half-applied trial step, no pending finalisation)
        cg, pg = gpu.get_params()
        cr, pr = ref.get_params()
        assert np.allclose(cg, cr, rtol=1e-7, atol=1e-10) and np.allclose(pg, pr, rtol=1e-6, atol=1e-9)
        chi2_now = gpu.chi2()
        assert abs(chi2_now - ct_g[-1]) / ct_g[-1] < 1e-12
    gpu.close()


@pytest.mark.parametrize("solver", ["pcg", "pcg_schur", "pcg_schur_implicit", "dense_schur"])
def test_levenberg_marquardt2_early_termination(oracle_mod, solver):
    """gr_lm_options.early_stop = optimizer::levenberg_marquardt2 (levenberg_marquardt.hpp:255-418): the loop
    leaves at the same iteration as the oracle's, with the same trace up to there."""
    prob, gpu, ref = make_pair(oracle_mod, "mini-50", np.float64)
    gs = dict(pcg=ga.SOLVER_PCG, pcg_schur=ga.SOLVER_PCG_SCHUR, pcg_schur_implicit=ga.SOLVER_PCG_SCHUR_IMPLICIT,
              dense_schur=ga.SOLVER_DENSE_SCHUR)[solver]
    os_ = dict(pcg=oracle_mod.SOLVER_PCG, pcg_schur=oracle_mod.SOLVER_PCG_SCHUR, pcg_schur_implicit=oracle_mod.SOLVER_PCG_SCHUR,
               dense_schur=oracle_mod.SOLVER_LDLT_SCHUR)[solver]
    ct_g, lt_g, st = gpu.levenberg_marquardt(solver=gs, iterations=40, early_stop=True)
    ct_r, lt_r, st_r = ref.levenberg_marquardt(solver=os_, iterations=40, early_stop=True)
    assert st["iterations_run"] == st_r["iterations_run"] < 40
    assert np.allclose(ct_g, ct_r, rtol=1e-6) and np.allclose(lt_g, lt_r, rtol=1e-3)
    gpu.set_params(prob.cameras, prob.points)
    ct_full, _, st_full = gpu.levenberg_marquardt(solver=gs, iterations=40)
    assert st_full["iterations_run"] > st["iterations_run"]
    assert np.allclose(ct_full[:len(ct_g)], ct_g, rtol=1e-9)
    gpu.close()


def test_dpp_lane_exchanges_equal_shfl_xor():
    """common.hpp lane_xor<1|2|4|8> (DPP quad_perm / row_ror moves) and the wave sums built on them against __shfl_xor, every lane"""
    import ctypes as C
    from graphite_amd import _lib
    bad = C.c_int(-1)
    _lib.check(_lib.lib().gr_test_lane_xor(C.c_int(0), C.byref(bad)))
    assert bad.value == 0
