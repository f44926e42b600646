"""A/B of the matrix-free operator with K sub-tiles per wave (GR_OP_K) and of whole LM runs."""
import sys, os, ctypes as C, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga
from graphite_amd import synth
import torch
name = sys.argv[1] if len(sys.argv) > 1 else 'ladybug-1723'
dt = np.float64 if (len(sys.argv) < 3 or sys.argv[2] == 'f64') else np.float32
ks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3").split(",")]
prob = synth.make_config(name)
for k in ks:
    os.environ["GR_OP_K"] = str(k)
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
    f = g.lib.gr_bal_diag_time; f.restype = C.c_double
    t_op = [f(g.h, C.c_int(0), C.c_int(0), C.c_int(50)) for _ in range(3)]
    kw = dict(solver=ga.SOLVER_PCG, iterations=20, pcg_tol=0.0)
    g.set_params(prob.cameras, prob.points); g.levenberg_marquardt(**kw)
    rates = []
    for _ in range(5):
        g.set_params(prob.cameras, prob.points); torch.cuda.synchronize(); t0 = time.perf_counter()
        ct, lt, st = g.levenberg_marquardt(**kw); torch.cuda.synchronize(); rates.append(st["iterations_run"] / (time.perf_counter() - t0))
    print(f"{name} K={k}: operator {min(t_op):.2f} us (3 x 50 launches: {[round(x,2) for x in t_op]}); LM tol=0: {sorted(rates)[2]:.0f} it/s, chi2 {ct[-1]:.9g}, pcg {st['pcg_iterations']}", flush=True)
    g.close()
