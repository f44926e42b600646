# How far is the fp32 engine from the fp32 oracle, and how far is each from the fp64 oracle ("truth")?  Ladybug-49, every solver.
import sys, numpy as np
sys.path.insert(0, ".")
import graphite_amd as ga, oracle
from graphite_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "ladybug-49"
prob = synth.make_config(name)
its = 6
def rel(a, b, k=None):
    a, b = np.asarray(a, float), np.asarray(b, float); m = min(len(a), len(b)) if k is None else k
    return float(np.max(np.abs(a[:m] - b[:m]) / np.abs(b[:m])))
for sname, gs, os_ in (("pcg", ga.SOLVER_PCG, oracle.SOLVER_PCG), ("pcg_schur", ga.SOLVER_PCG_SCHUR, oracle.SOLVER_PCG_SCHUR),
                       ("implicit", ga.SOLVER_PCG_SCHUR_IMPLICIT, oracle.SOLVER_PCG_SCHUR), ("dense_schur", ga.SOLVER_DENSE_SCHUR, oracle.SOLVER_LDLT_SCHUR)):
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float32)
    ct, lt, st = g.levenberg_marquardt(solver=gs, iterations=its); g.close()
    r32 = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float32)
    c32, l32, s32 = r32.levenberg_marquardt(solver=os_, iterations=its)
    r64 = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float64)
    c64, l64, s64 = r64.levenberg_marquardt(solver=os_, iterations=its)
    print("%-12s gpu32 vs oracle32: first4 %.2e all %.2e | gpu32 vs oracle64: %.2e | oracle32 vs oracle64: %.2e | pcg its gpu %d o32 %d o64 %d acc %d %d %d" % (
        sname, rel(ct, c32, 4), rel(ct, c32), rel(ct, c64), rel(c32, c64), st["pcg_iterations"], s32["pcg_iterations"], s64["pcg_iterations"], st["accepted"], s32["accepted"], s64["accepted"]))
    print("             chi2 gpu32 ", " ".join("%.6g" % x for x in ct))
    print("             chi2 ora32 ", " ".join("%.6g" % x for x in c32))
    print("             chi2 ora64 ", " ".join("%.6g" % x for x in c64))
