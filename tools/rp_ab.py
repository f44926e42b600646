"""A/B of the resident PCG launch (kernels_rp.hpp, GR_PCG_RESIDENT) against the operator / update / direction launches:
chi2 / damping traces, inner-iteration counts, final vertices, and the time per LM iteration.
usage: python tools/rp_ab.py [config ...] [--dtype f64|f32] [--iters N] [--repeat R]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga  # noqa: E402
from graphite_amd import synth  # noqa: E402


def run(prob, dtype, mode, iters, repeat, solver):
    os.environ["GR_PCG_RESIDENT"] = mode
    g = ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx, dtype=dtype)
    best, out = None, None
    for r in range(repeat):
        g.set_params(prob.cameras, prob.points)
        t0 = time.perf_counter()
        ct, lt, st = g.levenberg_marquardt(solver=solver, iterations=iters)
        dt = time.perf_counter() - t0
        loop = dt - st["setup_seconds"]
        best = loop if best is None else min(best, loop)
        out = (ct.copy(), lt.copy(), dict(st))
    c, p = g.get_params()
    g.set_params(prob.cameras, prob.points)
    ct, lt, st = g.levenberg_marquardt(solver=solver, iterations=iters, profile=True)
    ks = g.kernel_stats()
    g.close()
    return out + (c, p, best, ks)


def main():
    args = [a for a in sys.argv[1:]]
    dtype = np.float64
    iters, repeat = 20, 5
    names = []
    i = 0
    while i < len(args):
        if args[i] == "--dtype": dtype = np.float32 if args[i + 1] == "f32" else np.float64; i += 2
        elif args[i] == "--iters": iters = int(args[i + 1]); i += 2
        elif args[i] == "--repeat": repeat = int(args[i + 1]); i += 2
        else: names.append(args[i]); i += 1
    for name in names or ["mini-50", "ladybug-49", "ladybug-1723"]:
        prob = synth.make_config(name)
        for solver, sname in ((ga.SOLVER_PCG, "pcg"),):
            a = run(prob, dtype, "0", iters, repeat, solver)
            b = run(prob, dtype, "1", iters, repeat, solver)
            n = min(len(a[0]), len(b[0]))
            rel = np.max(np.abs(a[0][:n] - b[0][:n]) / np.abs(a[0][:n]))
            print(f"{name} {np.dtype(dtype).name} {sname}: launches {a[5] / iters * 1e6:.1f} us/LM-it, resident {b[5] / iters * 1e6:.1f} us/LM-it "
                  f"({a[5] / b[5]:.2f} x); chi2 trace rel diff {rel:.2e}; len {len(a[0])}/{len(b[0])}; inner iterations {a[2]['pcg_iterations']}/{b[2]['pcg_iterations']}; "
                  f"accepted {a[2]['accepted']}/{b[2]['accepted']}; launches {a[2]['kernel_launches']}/{b[2]['kernel_launches']}; "
                  f"max |dcam| {np.max(np.abs(a[3] - b[3])):.2e} max |dpt| {np.max(np.abs(a[4] - b[4])):.2e}; final chi2 {a[0][-1]:.9g} / {b[0][-1]:.9g}")
            for tag, ks in (("launches", a[6]), ("resident", b[6])):
                for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"]):
                    if v["launches"]:
                        print(f"    {tag:9s} {k:22s} n={v['launches']:4d} active={v['active_launches']:4d} {v['total_ms'] * 1e3 / max(1, v['launches']):8.2f} us/launch "
                              f"{v['bytes_per_launch'] / 1e6:8.2f} MB/launch")


if __name__ == "__main__":
    main()
