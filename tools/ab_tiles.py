"""A/B of the point-tiled observation order (GR_PTILES=K) on the per-observation kernels and on whole LM runs."""
import sys, os, ctypes as C, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga
from graphite_amd import synth
import torch
name = sys.argv[1] if len(sys.argv) > 1 else 'venice-1778'
dt = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == 'f64') else np.float32
ks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,8,16,32,64").split(",")]
prob = synth.make_config(name)
ref = None
for K in ks:
    os.environ["GR_PTILES"] = str(K)
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
    f = g.lib.gr_bal_diag_time; f.restype = C.c_double
    g.solver_update_structure(ga.SOLVER_PCG)
    t = {nm: min(f(g.h, C.c_int(w), C.c_int(0), C.c_int(20)) for _ in range(2)) for w, nm in ((0, "operator"), (1, "k_linearize"), (7, "linearisation"), (2, "chi2"))}
    kw = dict(solver=ga.SOLVER_PCG, iterations=8)
    g.set_params(prob.cameras, prob.points); g.levenberg_marquardt(**kw)
    rates = []
    for _ in range(3):
        g.set_params(prob.cameras, prob.points); torch.cuda.synchronize(); t0 = time.perf_counter()
        ct, lt, st = g.levenberg_marquardt(**kw); torch.cuda.synchronize(); rates.append(st["iterations_run"] / (time.perf_counter() - t0))
    if ref is None: ref = ct
    print(f"{name} K={K}: " + ", ".join(f"{k} {v:.1f} us" for k, v in t.items()) + f"; LM {sorted(rates)[1]:.0f} it/s, chi2 {ct[-1]:.9g} (trace rel diff vs K={ks[0]}: {np.max(np.abs(ct - ref) / ref):.2e}), pcg {st['pcg_iterations']}", flush=True)
    g.close()
