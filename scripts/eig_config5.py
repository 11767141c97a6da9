"""Config 5 (eps = 10 sphere in a PEC sphere, octant box; example/run.py:13-28) eigen-solve at --cells^3:
non-Hermitian Krylov-Schur, GMRES + multigrid inner solves.  Prints iterations and timings."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=128)
    ap.add_argument("--block", type=int, default=1)
    ap.add_argument("--nev", type=int, default=4)
    ap.add_argument("--lin-basis", type=int, default=40)
    ap.add_argument("--lin-type", default="gmres")
    a = ap.parse_args()
    import maxwell_amd as mx
    ctx = mx.Ctx(0)
    kw = dict(origin=(0.0,) * 3, size=(0.5,) * 3, lower=("pmc", "pmc", "pec"), upper=("pec",) * 3,
              pec_shape="sphere(0.49;0,0,0)", dielectrics=[(10.0, "sphere(0.37;0,0,0)")])
    t0 = time.perf_counter()
    sim = mx.Sim(a.cells, **kw)
    t1 = time.perf_counter()
    p = mx.SolverParams(nev=a.nev, basis=30, tol=1e-8, shift=0.5, lin_type=a.lin_type, lin_basis=a.lin_basis, lin_tol=1e-9,
                        prec_type="amg", block_size=a.block)
    s = mx.Solver(ctx, sim, p)
    t2 = time.perf_counter()
    n = s.solve()
    st = s.stats()
    print(json.dumps(dict(cells=a.cells, block=a.block, lin_type=a.lin_type, converged=n, assembly=t1 - t0, setup=t2 - t1, **st,
                          k=[abs(s.result(i)["k2"]) ** 0.5 for i in range(n)])))


if __name__ == "__main__":
    main()
