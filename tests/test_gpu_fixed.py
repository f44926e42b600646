"""Fixed vertices on the BAL engine (VertexDescriptor::set_fixed, vertex.hpp:262-264; the reference's kernels skip the
Jacobian blocks of a fixed vertex, ops/linearize.hpp:24, ops/hessian.hpp:95) against the oracle's restatement:
LM traces, inner iteration counts, the step at fixed vertices exactly zero, their parameters bit-unchanged — for every
form of the matrix-free PCG the engine has (direction kernel, lazy direction, single-reduction recurrence, point-tiled
order), for the Schur solvers, and on landmark shards."""
import threading

import numpy as np
import pytest

import graphite_amd as ga
from graphite_amd import dist as gdist, synth, _lib

pytestmark = pytest.mark.gpu

FORMS = {"default": {}, "direction_kernel": {"GR_PCG_LAZY": "0"}, "lazy": {"GR_PCG_LAZY": "1"},
         "single_reduction": {"GR_PCG_CG": "1"}, "tiled8": {"GR_PTILES": "8"}}


def masks(prob):
    Nc, Np, _ = prob.shape
    cf = np.zeros(Nc, bool)
    cf[[0, 7 % Nc, Nc - 1]] = True
    pf = np.zeros(Np, bool)
    pf[:10] = True
    pf[Np // 2] = True
    pf[Np - 1] = True
    return cf, pf


@pytest.mark.parametrize("form", list(FORMS))
@pytest.mark.parametrize("solver", ["pcg", "pcg_identity", "pcg_schur", "pcg_schur_implicit", "dense_schur"])
@pytest.mark.parametrize("name,dtype", [("mini-50", np.float64), ("ladybug-49", np.float32)])
def test_lm_with_fixed_vertices_matches_oracle(oracle_mod, monkeypatch, name, dtype, solver, form):
    for k in ("GR_PCG_LAZY", "GR_PCG_CG", "GR_PTILES"):
        monkeypatch.delenv(k, raising=False)
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    if solver in ("pcg_schur", "pcg_schur_implicit", "dense_schur") and form not in ("default", "tiled8"):
        pytest.skip("PCG forms concern the matrix-free solvers")
    gs = dict(pcg=ga.SOLVER_PCG, pcg_identity=ga.SOLVER_PCG_IDENTITY, pcg_schur=ga.SOLVER_PCG_SCHUR, pcg_schur_implicit=ga.SOLVER_PCG_SCHUR_IMPLICIT,
              dense_schur=ga.SOLVER_DENSE_SCHUR)[solver]
    os_ = dict(pcg=oracle_mod.SOLVER_PCG, pcg_identity=oracle_mod.SOLVER_PCG_IDENTITY, pcg_schur=oracle_mod.SOLVER_PCG_SCHUR,
               pcg_schur_implicit=oracle_mod.SOLVER_PCG_SCHUR, dense_schur=oracle_mod.SOLVER_LDLT_SCHUR)[solver]
    prob = synth.make_config(name)
    cf, pf = masks(prob)
    gpu = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    gpu.set_fixed(cf, pf)
    ref.set_fixed(cf, pf)
    # one solve: the step is exactly zero at the fixed vertices
    gpu.solver_update_structure(gs)
    gpu.linearize()
    gpu.solver_update_values(gs)
    gpu.solver_set_damping(gs, 1e-4)
    ref.linearize()
    ref.solver_update_values(os_)
    ref.solver_set_damping(os_, 1e-4)
    dx_g, it_g = gpu.solver_solve(gs, max_iter=6, tol=0.0, rej=1e6)
    dx_r, it_r = ref.solver_solve(os_, max_iter=6, tol=0.0, rej=1e6)
    Nc = prob.shape[0]
    fixed_entries = np.concatenate([np.repeat(cf, 9), np.repeat(pf, 3)])
    assert it_g == it_r
    assert np.all(dx_g[fixed_entries] == 0) and np.all(dx_r[fixed_entries] == 0)
    f64 = np.dtype(dtype) == np.float64
    assert np.abs(dx_g - dx_r).max() / np.abs(dx_r).max() < (1e-9 if f64 else 2e-3)
    b_g, b_r = gpu.get("b"), ref.get("b")
    assert np.all(b_g[fixed_entries] == 0)
    assert np.abs(b_g - b_r).max() / np.abs(b_r).max() < (1e-10 if f64 else 2e-3)
    # the LM loop
    ct, lt, st = gpu.levenberg_marquardt(solver=gs, iterations=6)
    ct_r, lt_r, st_r = ref.levenberg_marquardt(solver=os_, iterations=6)
    c, p = gpu.get_params()
    gpu.close()
    assert np.array_equal(c[cf], prob.cameras[cf].astype(dtype)) and np.array_equal(p[pf], prob.points[pf].astype(dtype))
    assert np.abs(c[~cf] - prob.cameras[~cf]).max() > 1e-3      # the others did move
    if f64:
        assert st["pcg_iterations"] == st_r["pcg_iterations"] and st["accepted"] == st_r["accepted"]
        assert np.max(np.abs(ct - ct_r) / ct_r) < 1e-8
    else:
        assert np.max(np.abs(ct[:4] - ct_r[:4]) / ct_r[:4]) < 1e-5


def test_fixed_vertices_on_landmark_shards(oracle_mod):
    prob = synth.make_config("mini-50")
    cf, pf = masks(prob)
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx)
    ref.set_fixed(cf, pf)
    ct_r, _, st_r = ref.levenberg_marquardt(solver=oracle_mod.SOLVER_PCG, iterations=6)
    world = 3
    shards = [gdist.partition_by_landmark(prob, r, world) for r in range(world)]
    engines = [ga.BalProblem(s.cameras, s.points, s.obs, s.cam_idx, s.pt_idx, dtype=np.float64, shard=True) for s in shards]
    for e, s in zip(engines, shards):
        e.set_fixed(cf, pf[s.point_ids])
    gdist.init_local_group(engines)
    out, err = [None] * world, []

    def work(r):
        try:
            out[r] = engines[r].levenberg_marquardt(solver=ga.SOLVER_PCG, iterations=6)
        except Exception as e:  # pragma: no cover
            err.append(e)

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not err, err
    cams = [e.get_params()[0] for e in engines]
    pts = gdist.assemble_points(shards, [e.get_params()[1] for e in engines])
    [e.close() for e in engines]
    for r in range(world):
        assert np.allclose(out[r][0], ct_r, rtol=1e-8)
        assert np.array_equal(cams[r], cams[0])
    assert np.array_equal(cams[0][cf], prob.cameras[cf]) and np.array_equal(pts[pf], prob.points[pf])
