#!/usr/bin/env python3
"""Generates the committed fixtures of tests/golden/ from the CPU oracle.

These are REGRESSION PINS of the oracle (inputs + its outputs), not outputs of the
reference: /root/reference cannot be compiled or imported in this container (no CUDA,
Eigen, Boost).  The oracle itself is pinned against the reference's own
literal known answers in tests/test_oracle_known_answers.py and its 1e-12 Schur relation
in tests/test_oracle_bal.py.    Usage:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from graphite_amd import synth  # noqa: E402

prob = synth.schur_test_fixture()
o = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx)
o.linearize(); o.hessian_update(); o.schur_update()
np.savez(os.path.join(HERE, "schur_2x3_f64.npz"), cameras=prob.cameras, points=prob.points, obs=prob.obs,
         cam_idx=prob.cam_idx, pt_idx=prob.pt_idx, **{k: o.get(k) for k in ("res", "scales", "b", "Hcc", "Hcp", "Hll", "S", "b_schur")})

prob = synth.make_config("mini-50")
out = dict(cam_idx=prob.cam_idx, pt_idx=prob.pt_idx, obs=prob.obs, cameras=prob.cameras, points=prob.points)
for solver, key in ((0, "chi2_pcg_schur"), (1, "chi2_pcg"), (4, "chi2_ldlt_schur")):
    o = oracle.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx)
    out[key], _, _ = o.levenberg_marquardt(solver=solver, iterations=8)
np.savez_compressed(os.path.join(HERE, "mini50_lm_f64.npz"), **out)
print("wrote", os.listdir(HERE))
