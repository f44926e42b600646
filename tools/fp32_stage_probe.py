# per-stage accuracy in fp32 against the fp64 oracle: engine vs fp32 oracle (mini-50 / ladybug-49)
import sys, numpy as np
sys.path.insert(0, ".")
import graphite_amd as ga, oracle
from graphite_amd import synth
prob = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "mini-50")
def relerr(a, b): return float(np.abs(np.asarray(a, float) - np.asarray(b, float)).max() / np.abs(np.asarray(b, float)).max())
def blockerr(a, b, bs):  # worst block-relative error
    a = np.asarray(a, float).reshape(-1, bs); b = np.asarray(b, float).reshape(-1, bs)
    return float((np.abs(a - b).max(1) / np.abs(b).max(1)).max())
g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float32)
g.solver_update_structure(ga.SOLVER_PCG_SCHUR); g.linearize(); g.solver_update_values(ga.SOLVER_PCG_SCHUR)
early = {k: g.get(k) for k in ("Hcc", "Hll", "Hcp", "b", "scales")}
g.solver_set_damping(ga.SOLVER_PCG_SCHUR, 1e-4); g.schur_update_values()
refs = {}
for tag, dt in (("o32", np.float32), ("o64", np.float64)):
    r = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
    r.linearize(); r.hessian_update()
    r.early = {k: r.get(k) for k in ("Hcc", "Hll", "Hcp", "b", "scales")}
    r.apply_damping(1e-4); r.schur_update()
    refs[tag] = r
for name, bs in (("residuals", 2), ("scales", 1), ("b", 1), ("Hcc", 81), ("Hll", 9), ("Hcp", 27), ("S", 81), ("b_schur", 9), ("Hll_inv", 9)):
    on = {"residuals": "res"}.get(name, name)
    a = early[name] if name in early else g.get(name)
    o32 = refs["o32"].early[on] if on in early else refs["o32"].get(on)
    o64 = refs["o64"].early[on] if on in early else refs["o64"].get(on)
    print("%-10s max-rel: gpu32 vs o64 %.2e | o32 vs o64 %.2e || worst block: gpu32 %.2e | o32 %.2e" % (name, relerr(a, o64), relerr(o32, o64), blockerr(a, o64, bs), blockerr(o32, o64, bs)))
