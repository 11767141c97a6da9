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


def test_bench_launcher_stops_when_a_rank_dies():
    """`python bench.py --gpus 2` without a launcher starts its own ranks; a rank that dies (here: no GPU at all) must end
    the whole run with a non-zero status instead of leaving the others in a rendezvous."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cells", "8", "--eig-n", "0",
                        "--steps", "1", "--warmup", "0", "--cpu-seconds", "0"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode != 0
    assert "rank exit codes" in r.stderr
    assert time.time() - t0 < 240
