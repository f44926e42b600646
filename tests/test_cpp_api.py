"""The host-side C++ mirror (include/graphite_mi355x.hpp) and the bal.cpp driver compile with plain
g++ against the C-ABI library (CPU check) and behave like the reference's solver tests on a GPU."""
import os
import subprocess

import numpy as np
import pytest

from graphite_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")


def compile_cpp(src, out):
    _lib.build()
    os.makedirs(BUILD, exist_ok=True)
    lib_dir = os.path.join(ROOT, "graphite_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-L", lib_dir,
           "-lgraphite_mi355x", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", out]
    subprocess.check_call(cmd)
    return out


def test_cpp_mirror_and_example_compile():
    compile_cpp(os.path.join(ROOT, "tests", "cpp", "test_cpp_api.cpp"), os.path.join(BUILD, "test_cpp_api"))
    compile_cpp(os.path.join(ROOT, "examples", "bal.cpp"), os.path.join(BUILD, "bal"))


@pytest.mark.gpu
def test_cpp_mirror_runs():
    exe = compile_cpp(os.path.join(ROOT, "tests", "cpp", "test_cpp_api.cpp"), os.path.join(BUILD, "test_cpp_api"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "OK (0 failures)" in r.stdout


ORACLE_SOLVER = {"pcg": "SOLVER_PCG", "pcg-schur": "SOLVER_PCG_SCHUR", "pcg-schur-implicit": "SOLVER_PCG_SCHUR",
                 "eigen-schur": "SOLVER_LDLT_SCHUR", "cudss-schur": "SOLVER_LDLT_SCHUR", "eigen": "SOLVER_LDLT", "cudss": "SOLVER_LDLT"}


@pytest.mark.gpu
@pytest.mark.parametrize("solver,precision", [("pcg", "FP64-FP64"), ("pcg-schur", "FP32-FP32"), ("pcg-schur-implicit", "FP64-FP64"),
                                              ("eigen-schur", "FP64-FP64"), ("cudss-schur", "FP32-FP32"), ("pcg", "FP64-FP32"),
                                              ("eigen", "FP64-FP64"), ("cudss", "FP64-FP64")])
def test_bal_driver_on_a_bal_file(oracle_mod, tmp_path, solver, precision):
    exe = compile_cpp(os.path.join(ROOT, "examples", "bal.cpp"), os.path.join(BUILD, "bal"))
    prob = synth.make_config("mini-50")
    f = tmp_path / "problem-50-2000-pre.txt"
    synth.write_bal(f, prob)
    r = subprocess.run([exe, str(f), "--iterations", "10", "--solver", solver, "--precision", precision, "--verbose",
                        "--hybrid_memory", "0"],
                       capture_output=True, text=True, timeout=120)
    print(r.stdout[-1500:], r.stderr)
    assert r.returncode == 0
    mse = float([ln for ln in r.stdout.splitlines() if ln.startswith("MSE:")][0].split()[1])
    assert 0.2 < mse < 0.5          # ~ (0.5 px)^2 * 2 * (1 - dof/2No)
    assert "Iteration" in r.stdout and "Lambda" in r.stdout
    # the printed MSE against the oracle's LM with the solver the option names (examples/bal.cu:254-273); the full-system
    # direct solvers against the oracle's full-H LDL^T: same step as the Schur direct solve up to rounding
    ref = oracle_mod.BalOracle(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx,
                               dtype=np.float32 if precision == "FP32-FP32" else np.float64)
    ct, _, _ = ref.levenberg_marquardt(solver=getattr(oracle_mod, ORACLE_SOLVER[solver]), iterations=10)
    want = ct[-1] / prob.shape[2]
    assert abs(mse - want) / want < (2e-3 if precision == "FP32-FP32" else 1e-6 if precision == "FP64-FP64" else 1e-4)
