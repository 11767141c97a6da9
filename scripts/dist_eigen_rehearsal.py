"""Decomposed eigen-solve on ONE GPU: R rank threads of an in-process world (mxg_local_world), each owning an
n_local^3-cell MxGridDomain box of the weak-scaling pillbox workload of bench.py, against the single-rank solve of
the same global grid.  Shows what the rank-local multigrid blocks cost in inner iterations; timings are NOT
multi-GPU timings (the ranks share one device and the collectives are staged through the host).

  python scripts/dist_eigen_rehearsal.py --ranks 2 --cells 64
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--ranks", type=int, default=2)
    p.add_argument("--cells", type=int, default=64)
    p.add_argument("--out", default=None)
    args = p.parse_args()
    import __graft_entry__ as entry
    entry.build()
    import maxwell_amd as mx
    from maxwell_amd import distributed as mxd
    from maxwell_amd.api import PILLBOX_SHAPE

    R = args.ranks
    pg, _, _ = mxd.partition([args.cells] * 3, R, 0)
    n = [args.cells * q for q in pg]
    shift = 0.05 * (2.0 * np.pi) ** 2
    kw = dict(nev=5, basis=30, tol=1e-8, shift=shift, lin_type="cg", lin_tol=1e-9, prec_type="amg")
    box = dict(origin=(-0.5,) * 3, size=(1.0,) * 3)

    def rank(ctx):
        t0 = time.perf_counter()
        sim = mxd.BoxSim(n, ctx.rank, R, pec_shape=PILLBOX_SHAPE, **box)
        t1 = time.perf_counter()
        solver = mx.Solver(ctx, sim, mx.SolverParams(**kw))
        t2 = time.perf_counter()
        nconv = solver.solve()
        st = solver.stats()
        res = [solver.result(i) for i in range(nconv)]
        solver.close()
        return dict(assembly=t1 - t0, setup=t2 - t1, stats=st, k2=[r["k2"].real for r in res],
                    residual=max(r["residual"] for r in res))

    world = mx.LocalWorld(R)
    out = world.run(rank)
    world.close()
    ctx = mx.Ctx(0)
    t0 = time.perf_counter()
    sim = mx.Sim(n, pec_shape=PILLBOX_SHAPE, **box)
    solver = mx.Solver(ctx, sim, mx.SolverParams(**kw))
    t1 = time.perf_counter()
    nconv = solver.solve()
    st1 = solver.stats()
    k1 = [solver.result(i)["k2"].real for i in range(nconv)]
    rep = {"grid": n, "ranks": R, "proc_grid": pg,
           "decomposed": {"k2": out[0]["k2"], "applies": out[0]["stats"]["applies"], "vec_lin_iters": out[0]["stats"]["vec_iters"],
                          "sca_lin_iters": out[0]["stats"]["sca_iters"], "eigensolve_seconds_shared_device": out[0]["stats"]["eigensolve_seconds"],
                          "assembly_seconds": max(o["assembly"] for o in out), "setup_seconds": max(o["setup"] for o in out),
                          "max_rel_residual": out[0]["residual"]},
           "single_rank": {"k2": k1, "applies": st1["applies"], "vec_lin_iters": st1["vec_iters"], "sca_lin_iters": st1["sca_iters"],
                           "eigensolve_seconds": st1["eigensolve_seconds"], "setup_seconds": t1 - t0},
           "max_rel_eigenvalue_difference": float(np.max(np.abs(np.sort(out[0]["k2"]) - np.sort(k1)) / np.abs(np.sort(k1))))}
    s = json.dumps(rep, indent=1)
    print(s)
    if args.out:
        with open(args.out, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
