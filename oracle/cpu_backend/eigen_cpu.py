"""Child process of oracle.cpu_backend.eigen: the maxwell_amd Python mirror over the CPU libraries of oracle/_cpu/."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
n, nev, tol, lin_tol = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
eig_type = sys.argv[5] if len(sys.argv) > 5 else "krylov-schur"
json_fd = os.dup(1)
os.dup2(2, 1)                                    # the solver's tables go to stderr, the JSON line to stdout
import numpy as np
from maxwell_amd import _lib
_lib._LIBDIR = os.path.join(ROOT, "oracle", "_cpu")   # both libraries from the CPU build
import maxwell_amd as mx

ctx = mx.Ctx(0)
assert "CPU backend" in mx.CsrMatrix.from_scipy(ctx, __import__("scipy.sparse").sparse.identity(4, format="csr")).kernel_name()
t0 = time.perf_counter()
sim = mx.Sim.pillbox(n)
t1 = time.perf_counter()
params = mx.SolverParams(nev=nev, basis=30, tol=tol, shift=0.05 * (2.0 * np.pi) ** 2, lin_type="cg", lin_tol=lin_tol, prec_type="amg", eig_type=eig_type)
solver = mx.Solver(ctx, sim, params)
t2 = time.perf_counter()
nconv = solver.solve()
st = solver.stats()
res = [solver.result(i) for i in range(nconv)]
out = {"cells": n, "converged": nconv, "assembly_seconds": round(t1 - t0, 2), "solver_setup_seconds": round(t2 - t1, 2),
       "eigensolve_seconds": round(st["eigensolve_seconds"], 3), "eigenpairs_per_s": round(nconv / st["eigensolve_seconds"], 4),
       "applies": st["applies"], "vec_lin_iters": st["vec_iters"], "sca_lin_iters": st["sca_iters"],
       "k2": [r["k2"].real for r in res], "max_rel_residual": max(r["residual"] for r in res) if res else None,
       "threads": int(os.environ.get("OMP_NUM_THREADS", "0")) or (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())}
os.write(json_fd, (json.dumps(out) + "\n").encode())
