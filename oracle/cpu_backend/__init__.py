"""CPU backend of the device C ABI (include/mxg.h): TEST / BASELINE INFRASTRUCTURE (see mxg_cpu.cpp).

build() compiles oracle/_cpu/libmxg.so (mxg_cpu.cpp, g++ -fopenmp) and oracle/_cpu/libmxhost.so (the product's
host sources, unchanged, linked against THAT libmxg.so).  eigen(n, ...) runs the product's own MxSolver stack over it in
a child process (two libraries with one SONAME cannot share a process with the HIP build) and returns its JSON.
Only tests/ and bench.py's cpu_baseline leg use this module."""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(ROOT, "oracle", "_cpu")
HOST_SRC = os.path.join(ROOT, "maxwell_amd", "csrc", "host")
HOST_SOURCES = ["mx_geometry.cpp", "mx_yee.cpp", "mxh_sim.cpp", "mxh_mg.cpp", "mxh_solver.cpp", "mxh_problem.cpp", "mxh_io.cpp"]


def _digest():
    h = hashlib.sha256()
    files = [os.path.join(HERE, "mxg_cpu.cpp"), os.path.join(ROOT, "maxwell_amd", "csrc", "device", "mxg_krylov.h")]
    for d in (HOST_SRC, os.path.join(ROOT, "include"), os.path.join(ROOT, "include", "mx")):
        files += [os.path.join(d, f) for f in sorted(os.listdir(d)) if os.path.isfile(os.path.join(d, f))]
    for f in files:
        h.update(f.encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    stamp = os.path.join(OUT, "stamp")
    libg, libh = os.path.join(OUT, "libmxg.so"), os.path.join(OUT, "libmxhost.so")
    dig = _digest()
    if not force and os.path.exists(libg) and os.path.exists(libh) and os.path.exists(stamp) and open(stamp).read() == dig:
        return OUT
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "mx"),
           "-I" + os.path.join(ROOT, "maxwell_amd", "csrc", "device"), "-I" + HOST_SRC]
    common = ["g++", "-O3", "-march=native", "-fopenmp", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]
    subprocess.run(common + inc + [os.path.join(HERE, "mxg_cpu.cpp"), "-o", libg], check=True, stdout=sys.stderr)
    subprocess.run(["g++", "-O2", "-fopenmp", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"] + inc +
                   [os.path.join(HOST_SRC, f) for f in HOST_SOURCES] + ["-o", libh, "-L" + OUT, "-lmxg", "-ldl", "-Wl,-rpath,$ORIGIN"],
                   check=True, stdout=sys.stderr)
    open(stamp, "w").write(dig)
    return OUT


def eigen(n, nev=5, tol=1e-8, lin_tol=1e-9, threads=None, timeout=1200, eig_type="krylov-schur"):
    """The product's eigen-solve of the pillbox at n^3 (same parameters as bench.py's GPU `eigen`) on the host cores."""
    build()
    env = dict(os.environ)
    if threads:
        env["OMP_NUM_THREADS"] = str(threads)
    p = subprocess.run([sys.executable, os.path.join(HERE, "eigen_cpu.py"), str(n), str(nev), repr(tol), repr(lin_tol), eig_type],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError("CPU backend eigen-solve failed: " + p.stderr.decode()[-2000:])
    return json.loads(p.stdout.decode().strip().splitlines()[-1])
