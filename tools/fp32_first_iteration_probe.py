# Where the fp32 step loses digits: the first PCG iteration is dx = alpha * Minv b / |b| — the block-Jacobi
# inverses applied to the gradient.  Compares, per vertex type, the engine's fp32 step with the fp64 oracle's and with the fp32
# oracle's, and with the direction obtained by inverting the ENGINE's own fp32 blocks in float64 (inputs vs inverse).
import sys, numpy as np
sys.path.insert(0, ".")
import graphite_amd as ga, oracle
from graphite_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "mini-50"
prob = synth.make_config(name)
Nc, Np, No = prob.shape
def relerr(a, b): return float(np.abs(np.asarray(a, float) - np.asarray(b, float)).max() / np.abs(np.asarray(b, float)).max())
mu = 1e-4
g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=np.float32)
g.solver_update_structure(ga.SOLVER_PCG); g.linearize(); g.solver_update_values(ga.SOLVER_PCG); g.solver_set_damping(ga.SOLVER_PCG, mu)
dxg, _ = g.solver_solve(ga.SOLVER_PCG, max_iter=1, tol=0.0, rej=1e6)
Hcc = np.asarray(g.get("Hcc"), float).reshape(Nc, 9, 9); Hll = np.asarray(g.get("Hll"), float).reshape(Np, 3, 3); b = np.asarray(g.get("b"), float)
dx = {}
for tag, dt in (("o32", np.float32), ("o64", np.float64)):
    r = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dt)
    r.linearize(); r.solver_update_values(oracle.SOLVER_PCG); r.solver_set_damping(oracle.SOLVER_PCG, mu)
    dx[tag], _ = r.solver_solve(oracle.SOLVER_PCG, max_iter=1, tol=0.0, rej=1e6)
    if tag == "o64":
        Hcc64 = np.asarray(r.get("Hcc"), float).reshape(Nc, 9, 9); Hll64 = np.asarray(r.get("Hll"), float).reshape(Np, 3, 3); b64 = np.asarray(r.get("b"), float)
def damp(H):
    H = H.copy()
    n = H.shape[1]
    d = np.einsum("kii->ki", H)
    H[:, np.arange(n), np.arange(n)] = d + mu * np.clip(d, 1e-6, 1e32)
    return H
def direction(Hc, Hl, bb):
    zc = np.linalg.solve(damp(Hc), bb[:9 * Nc].reshape(Nc, 9, 1)).reshape(-1)
    zl = np.linalg.solve(damp(Hl), bb[9 * Nc:].reshape(Np, 3, 1)).reshape(-1)
    return np.concatenate([zc, zl])
z_own = direction(Hcc, Hll, b)       # the engine's fp32 blocks and gradient, inverted in float64
z_64 = direction(Hcc64, Hll64, b64)
for part, sl in (("cameras", slice(0, 9 * Nc)), ("points", slice(9 * Nc, None)), ("all", slice(None))):
    print("%-8s step: gpu32 vs o64 %.2e | o32 vs o64 %.2e | gpu32 vs o32 %.2e" % (part, relerr(dxg[sl], dx["o64"][sl]), relerr(dx["o32"][sl], dx["o64"][sl]), relerr(dxg[sl], dx["o32"][sl])))
    # directions, scale removed by a least-squares fit of the scalar
    def fit(a, ref): a = np.asarray(a, float); return a * (a @ ref) / (a @ a)
    ref = z_64[sl]
    print("%-8s direction (scalar fitted): gpu32 step %.2e | engine's fp32 blocks inverted in fp64 %.2e | o32 step %.2e" %
          (part, relerr(fit(dxg[sl], ref), ref), relerr(fit(z_own[sl], ref), ref), relerr(fit(dx["o32"][sl], ref), ref)))
print("alpha-like scale: |dx| gpu32 %.9g o32 %.9g o64 %.9g" % (np.linalg.norm(dxg), np.linalg.norm(dx["o32"]), np.linalg.norm(dx["o64"])))
