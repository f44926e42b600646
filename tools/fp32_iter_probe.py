# fp32 matrix-free PCG step against the fp64 oracle's, per inner-iteration count and vertex type, next to the fp32 oracle's own distance
# (VERDICT r4 next 3).   python tools/fp32_iter_probe.py [config] [max inner iterations]
import sys, numpy as np
sys.path.insert(0, ".")
import graphite_amd as ga, oracle
from graphite_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "mini-50"
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 6
prob = synth.make_config(name)
Nc, Np, No = prob.shape
mu = 1e-4
def relerr(a, b): return float(np.linalg.norm(np.asarray(a, float) - np.asarray(b, float)) / np.linalg.norm(np.asarray(b, float)))
def gpu(dt, k, solver=ga.SOLVER_PCG):
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
    g.solver_update_structure(solver); g.linearize(); g.solver_update_values(solver); g.solver_set_damping(solver, mu)
    return np.asarray(g.solver_solve(solver, max_iter=k, tol=0.0, rej=1e6)[0], float)
def ora(dt, k, solver=oracle.SOLVER_PCG):
    r = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
    r.linearize(); r.solver_update_values(solver); r.solver_set_damping(solver, mu)
    return np.asarray(r.solver_solve(solver, max_iter=k, tol=0.0, rej=1e6)[0], float)
print("%s: %d cameras, %d points, %d observations; relative 2-norm distances from the fp64 oracle's step" % (name, Nc, Np, No))
print("%3s | %-32s | %-32s | %-20s" % ("k", "engine fp32 (cam / pt / all)", "oracle fp32 (cam / pt / all)", "engine fp64 (all)"))
sc, sp = slice(0, 9 * Nc), slice(9 * Nc, None)
for k in range(1, kmax + 1):
    o64 = ora(np.float64, k); o32 = ora(np.float32, k); g32 = gpu(np.float32, k); g64 = gpu(np.float64, k)
    e = [relerr(g32[s], o64[s]) for s in (sc, sp, slice(None))]
    o = [relerr(o32[s], o64[s]) for s in (sc, sp, slice(None))]
    print("%3d | %.2e / %.2e / %.2e   | %.2e / %.2e / %.2e   | %.2e   ratio all %.2f" % (k, e[0], e[1], e[2], o[0], o[1], o[2], relerr(g64, o64), e[2] / max(o[2], 1e-30)))
