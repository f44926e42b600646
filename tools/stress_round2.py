"""Randomised cross-check of the round-2 paths against the oracle (run on the GPU box): random problem shapes, random fixed
masks, every solver, the PCG forms, tiled / gather orders, landmark shards with the single-reduction recurrence."""
import os, sys, threading, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga, oracle
from graphite_amd import synth, dist as gdist

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
GS = {"pcg": (ga.SOLVER_PCG, oracle.SOLVER_PCG), "pcg_identity": (ga.SOLVER_PCG_IDENTITY, oracle.SOLVER_PCG_IDENTITY),
      "pcg_schur": (ga.SOLVER_PCG_SCHUR, oracle.SOLVER_PCG_SCHUR), "implicit": (ga.SOLVER_PCG_SCHUR_IMPLICIT, oracle.SOLVER_PCG_SCHUR),
      "dense_schur": (ga.SOLVER_DENSE_SCHUR, oracle.SOLVER_LDLT_SCHUR)}
ENVS = [{}, {"GR_PTILES": "8"}, {"GR_PTILES": "16", "GR_G3_GATHER": "0"}, {"GR_PCG_LAZY": "0"}, {"GR_PCG_LAZY": "1"}, {"GR_PCG_CG": "1"},
        {"GR_G3_GATHER": "1"}, {"GR_PTILES": "8", "GR_PCG_CG": "1"}, {"GR_PTILES": "24", "GR_PCG_LAZY": "1"}]
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    Nc = int(rng.integers(4, 120)); Np = int(rng.integers(60, 4000)); deg = int(rng.integers(2, 6))
    No = min(Nc * Np, Np * deg + int(rng.integers(0, Np)))
    prob = synth.make_problem(Nc, Np, No, seed=int(rng.integers(1 << 30)), window=int(rng.integers(2, max(3, Nc))))
    Nc, Np, No = prob.shape
    use_fixed = rng.random() < 0.6
    cf = (rng.random(Nc) < 0.1) if use_fixed else None
    pf = (rng.random(Np) < 0.1) if use_fixed else None
    for k in ("GR_PTILES", "GR_G3_GATHER", "GR_PCG_LAZY", "GR_PCG_CG"):
        os.environ.pop(k, None)
    env = ENVS[int(rng.integers(len(ENVS)))]
    os.environ.update(env)
    name = list(GS)[int(rng.integers(len(GS)))]
    gs, osv = GS[name]
    its = 5
    dt = np.float32 if rng.random() < 0.3 else np.float64
    ref = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
    if use_fixed: ref.set_fixed(cf, pf)
    ref.set_pcg_single_reduction(1 if env.get("GR_PCG_CG") == "1" else 0)
    ct_r, _, st_r = ref.levenberg_marquardt(solver=osv, iterations=its)
    world = int(rng.integers(1, 4))
    if world == 1:
        e = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
        if use_fixed: e.set_fixed(cf, pf)
        ct, _, st = e.levenberg_marquardt(solver=gs, iterations=its)
        c, p = e.get_params(); e.close()
    else:
        try:
            shards = [gdist.partition_by_landmark(prob, r, world) for r in range(world)]
        except ValueError:
            continue
        es = [ga.BalProblem(s.cameras, s.points, s.obs, s.cam_idx, s.pt_idx, dtype=dt, shard=True) for s in shards]
        if use_fixed:
            for e, s in zip(es, shards): e.set_fixed(cf, pf[s.point_ids])
        gdist.init_local_group(es)
        out = [None] * world; errs = []
        def work(r):
            try: out[r] = es[r].levenberg_marquardt(solver=gs, iterations=its)
            except Exception as ex: errs.append(ex)
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        [t.start() for t in th]; [t.join(timeout=120) for t in th]
        if errs or any(o is None for o in out):
            print("TRIAL", trial, "sharded failure", errs); bad += 1; [e.close() for e in es]; continue
        ct, _, st = out[0]
        c = es[0].get_params()[0]; p = np.concatenate([e.get_params()[1] for e in es]); [e.close() for e in es]
    m = min(len(ct), len(ct_r)) if dt == np.float64 else min(len(ct), len(ct_r), 3)   # fp32: compared while both traces still move
    rel = float(np.max(np.abs(ct[:m] - ct_r[:m]) / np.abs(ct_r[:m])))
    ok = (rel < 1e-7 and len(ct) == len(ct_r)) if dt == np.float64 else rel < 5e-3
    if use_fixed:
        ok = ok and np.array_equal(c[cf], prob.cameras[cf].astype(dt)) and np.array_equal(p[pf], prob.points[pf].astype(dt))
    print(f"trial {trial}: Nc {Nc} Np {Np} No {No} solver {name} env {env} world {world} fixed {use_fixed} {np.dtype(dt).name}: rel {rel:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("failures:", bad)
sys.exit(1 if bad else 0)
