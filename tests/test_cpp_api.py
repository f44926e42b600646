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


@pytest.mark.gpu
@pytest.mark.parametrize("solver,precision", [("pcg", "FP64-FP64"), ("pcg-schur", "FP32-FP32"), ("pcg-schur-implicit", "FP64-FP64"),
                                              ("eigen-schur", "FP64-FP64"), ("cudss-schur", "FP32-FP32"), ("pcg", "FP64-FP32")])
def test_bal_driver_on_a_bal_file(tmp_path, solver, precision):
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
