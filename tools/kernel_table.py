"""Per-kernel HIP-event table of one LM run (profile mode): launches, active launches, mean us, share."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga
from graphite_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "ladybug-1723"
dt = np.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else np.float32
solver = {"pcg": ga.SOLVER_PCG, "pcg-schur": ga.SOLVER_PCG_SCHUR, "implicit": ga.SOLVER_PCG_SCHUR_IMPLICIT}[sys.argv[3] if len(sys.argv) > 3 else "pcg"]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
prob = synth.make_config(name)
g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
g.levenberg_marquardt(solver=solver, iterations=5)
ct, lt, st = g.levenberg_marquardt(solver=solver, iterations=iters, profile=True)
ks = [dict(name=n, **v) for n, v in g.kernel_stats().items()]
tot = sum(k["total_ms"] for k in ks)
print(f"{name} {np.dtype(dt).name}: {st['iterations_run']} LM iterations, {st['pcg_iterations']} PCG iterations, loop {st['loop_seconds']*1e3:.2f} ms (profiled), kernels {tot:.2f} ms")
for k in sorted(ks, key=lambda k: -k["total_ms"]):
    act = max(k["active_launches"], 1)
    print(f"  {k['name']:20s} launches {k['launches']:5d} active {k['active_launches']:5d}  mean {1e3*k['total_ms']/max(k['launches'],1):7.2f} us  per LM it {1e3*k['total_ms']/st['iterations_run']:7.2f} us  GB/s {k['bytes_per_launch']*act/max(k['total_ms'],1e-9)/1e6:8.1f}")
