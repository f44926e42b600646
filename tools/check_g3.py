"""PCG solve with g3 in pm order vs observation order (GR_G3_GATHER) on one workload: max |dx| difference."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga
from graphite_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "venice-1778"
dt = np.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else np.float32
prob = synth.make_config(name)
out = []
for g in ("0", "1"):
    os.environ["GR_G3_GATHER"] = g
    e = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
    e.solver_update_structure(ga.SOLVER_PCG)
    e.linearize()
    e.solver_update_values(ga.SOLVER_PCG)
    e.solver_set_damping(ga.SOLVER_PCG, 1e-4)
    for it in (1, 2, 5):
        dx, n = e.solver_solve(ga.SOLVER_PCG, max_iter=it, tol=0.0, rej=1e30)
        out.append((g, it, n, dx.copy()))
    e.close()
h = len(out) // 2
Nc = prob.shape[0]
for (g0, it, n0, d0), (g1, _, n1, d1) in zip(out[:h], out[h:]):
    diff = np.abs(d0 - d1)
    i = int(np.argmax(diff))
    print(f"iters {it}: n {n0} vs {n1}; max|dx| {np.abs(d0).max():.4g}; max diff {diff.max():.4g} at {i} ({'camera' if i < 9 * Nc else 'point %d' % ((i - 9 * Nc) // 3)}); "
          f"#entries off by >1e-3 rel: {(diff > 1e-3 * np.abs(d0).max()).sum()}")
    bad = np.nonzero(diff[9 * Nc:] > 1e-3 * np.abs(d0).max())[0] // 3
    if bad.size:
        print("   bad points: first", bad[:10], "last", bad[-10:], "count", np.unique(bad).size)
