"""A/B of the direct Schur solve: nested-dissection level-scheduled tile Cholesky (GR_SPARSE_CHOL=1) vs the dense tile Cholesky (=0)."""
import sys, os, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga
from graphite_amd import synth
import torch
name = sys.argv[1] if len(sys.argv) > 1 else 'ladybug-1723'
dt = np.float64 if (len(sys.argv) < 3 or sys.argv[2] == 'f64') else np.float32
prob = synth.make_config(name)
os.environ["GR_VERBOSE"] = "1"
res = {}
for mode in ("1", "0"):
    os.environ["GR_SPARSE_CHOL"] = mode
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
    g.solver_update_structure(ga.SOLVER_DENSE_SCHUR); g.linearize(); g.solver_update_values(ga.SOLVER_DENSE_SCHUR); g.solver_set_damping(ga.SOLVER_DENSE_SCHUR, 1e-4)
    dx, _ = g.solver_solve(ga.SOLVER_DENSE_SCHUR)
    res[mode] = dx
    kw = dict(solver=ga.SOLVER_DENSE_SCHUR, iterations=6)
    g.set_params(prob.cameras, prob.points); g.levenberg_marquardt(**kw)
    rates = []
    for _ in range(3):
        g.set_params(prob.cameras, prob.points); torch.cuda.synchronize(); t0 = time.perf_counter()
        ct, lt, st = g.levenberg_marquardt(**kw); torch.cuda.synchronize(); rates.append(st["iterations_run"] / (time.perf_counter() - t0))
    g.set_params(prob.cameras, prob.points)
    ct, lt, st = g.levenberg_marquardt(profile=True, **kw)
    ks = g.kernel_stats()
    print(f"{name} GR_SPARSE_CHOL={mode}: {sorted(rates)[1]:.1f} LM it/s, chi2 {ct[-1]:.10g}", flush=True)
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"])[:8]:
        print(f"    {k:22s} {v['launches']:5d} launches {v['total_ms']:9.3f} ms total, {v['total_ms']*1e3/max(v['launches'],1):8.1f} us avg, {v['flops_per_launch']/max(v['total_ms']/max(v['launches'],1)*1e-3,1e-12)/1e12:6.2f} TFLOP/s")
    g.close()
d = np.abs(res["1"] - res["0"]).max() / np.abs(res["0"]).max()
print("delta_x sparse vs dense: rel", d)
