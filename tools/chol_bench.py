import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, time, sys
import graphite_amd as ga
sizes = [int(a) for a in sys.argv[1:]] or [2048, 8192, 15507]
for dt in (torch.float64, torch.float32):
    for n in sizes:
        G = torch.randn(n, n, device="cuda", dtype=dt)
        A = G @ G.T / n + torch.eye(n, device="cuda", dtype=dt)
        del G
        b = np.random.default_rng(0).standard_normal(n)
        x, sec = ga.dense_cholesky_solve(A, b)
        x, sec = ga.dense_cholesky_solve(A, b)
        r = (A.double() @ torch.tensor(x, device="cuda", dtype=torch.float64) - torch.tensor(b, device="cuda")).abs().max().item()
        nt = (n + 127)//128
        fl = n**3/3
        print(f"{dt} n={n} factor {sec*1e3:.2f} ms  {fl/sec/1e12:.2f} TFLOP/s  resid {r:.2e}", flush=True)
        del A
