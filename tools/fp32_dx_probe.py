# delta_x of a 4-iteration solve in fp32: engine vs fp32 oracle vs fp64 oracle ("truth"), mini-50
import sys, numpy as np
sys.path.insert(0, ".")
import graphite_amd as ga, oracle
from graphite_amd import synth
prob = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "mini-50")
def relerr(a, b): return float(np.abs(np.asarray(a, float) - np.asarray(b, float)).max() / np.abs(b).max())
for sname, gs, os_ in (("pcg", ga.SOLVER_PCG, oracle.SOLVER_PCG), ("pcg_schur", ga.SOLVER_PCG_SCHUR, oracle.SOLVER_PCG_SCHUR), ("implicit", ga.SOLVER_PCG_SCHUR_IMPLICIT, oracle.SOLVER_PCG_SCHUR)):
    out = {}
    for tag, dt in (("g32", np.float32),):
        g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
        g.solver_update_structure(gs); g.linearize(); g.solver_update_values(gs); g.solver_set_damping(gs, 1e-4)
        out[tag] = {m: g.solver_solve(gs, max_iter=m, tol=0.0, rej=1e6)[0] for m in (1, 4)}
        out[tag + "b"] = g.get("b"); g.close()
    for tag, dt in (("o32", np.float32), ("o64", np.float64)):
        r = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
        r.linearize(); r.solver_update_values(os_); r.solver_set_damping(os_, 1e-4)
        out[tag] = {m: r.solver_solve(os_, max_iter=m, tol=0.0, rej=1e6)[0] for m in (1, 4)}
        out[tag + "b"] = r.get("b")
    for m in (1, 4):
        print("%-10s %d it: gpu32 vs o32 %.2e | gpu32 vs o64 %.2e | o32 vs o64 %.2e" % (sname, m, relerr(out["g32"][m], out["o32"][m]), relerr(out["g32"][m], out["o64"][m]), relerr(out["o32"][m], out["o64"][m])))
    print("%-10s b: gpu32 vs o32 %.2e | gpu32 vs o64 %.2e | o32 vs o64 %.2e" % (sname, relerr(out["g32b"], out["o32b"]), relerr(out["g32b"], out["o64b"]), relerr(out["o32b"], out["o64b"])))
