"""Host-side logic that needs no GPU: the Krylov-Schur eigen-solver over an in-memory MultiVec
(C++ unit test), compiled and run here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_krylov_schur_cpp_unit(tmp_path):
    exe = str(tmp_path / "test_krylov_schur")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "test_krylov_schur.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "PASSED" in out.stdout, out.stdout + out.stderr


def test_host_thread_budget():
    from maxwell_amd import _env
    n = _env.host_threads()
    assert 1 <= n <= (os.cpu_count() or 1)
