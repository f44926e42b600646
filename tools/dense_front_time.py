"""Device time of the MFMA dense Cholesky (DenseChol, gr_dense_cholesky_solve) on fronts of the size of Ladybug-1723's merged top separators
(second form: 'the top separators merged into one dense front factorised by DenseChol'): n = 14 tiles = 1 792 and around."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphite_amd as ga
rng = np.random.default_rng(3)
for n in (128, 256, 512, 1024, 1792, 2048, 3584):
    M = rng.standard_normal((n, n)); A = M @ M.T + n * np.eye(n); b = rng.standard_normal(n)
    best = 1e9
    for _ in range(4):
        x, sec = ga.bal.dense_cholesky_solve(A, b)
        best = min(best, sec)
    err = np.abs(A @ x - b).max()
    print(f"n {n:5d} ({n // 128:2d} tiles): factorisation {1e6 * best:8.1f} us = {1e6 * best / (n / 128):6.1f} us per tile column, {n ** 3 / 3 / best / 1e12:6.2f} TFLOP/s, residual {err:.1e}")
