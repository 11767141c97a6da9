"""Convergence / timing probe of the MG-preconditioned CG against Jacobi CG."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = mx.Ctx(0)
sim = mx.Sim.pillbox(n)
shift = 0.05 * (2 * np.pi) ** 2
mask = sim.used_mask()
rng = np.random.default_rng(0)
for op in ("shifted", "scaLapl"):
    for (omega, cs, sw) in ([(0.7, 1.9, 2), (0.7, 1.5, 2), (0.8, 1.9, 1), (0.6, 1.9, 3), (0.7, 1.0, 2)] if n <= 64 else [(0.7, 1.9, 2), (0.8, 1.9, 1)]):
        t0 = time.time()
        mg = mx.Multigrid(ctx, sim, op, shift, sweeps=sw, omega=omega, coarse_scale=cs)
        ts = time.time() - t0
        A = mg.fine_matrix()
        nrow = A.nrows
        if op == "shifted":
            bh = sim.op("mRhs") @ rng.uniform(-1, 1, nrow)
        else:
            bh = sim.op("divB") @ (sim.op("mRhs") @ rng.uniform(-1, 1, sim.op("mRhs").shape[0]))
        b = mx.MultiVec.from_numpy(ctx, bh)
        out = []
        for prec, m in (("jacobi", None), ("mg", mg)):
            x = mx.MultiVec(ctx, nrow, 1)
            ctx.sync(); t0 = time.time()
            its, conv, res = A.solve(b, x, method="cg", tol=1e-9, precond=prec, mg=m)
            ctx.sync(); out.append((prec, its, conv, "%.2e" % res, "%.3fs" % (time.time() - t0)))
        print(n, op, "omega", omega, "cscale", cs, "sweeps", sw, "levels", mg.levels, "setup %.2fs" % ts, out, flush=True)
