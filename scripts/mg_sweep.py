"""Parameter sweep of the multigrid-preconditioned CG on the 256^3 shifted vector Laplacian (time to 1e-9)."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = mx.Ctx(0)
sim = mx.Sim.pillbox(n)
shift = 0.05 * (2 * np.pi) ** 2
rng = np.random.default_rng(0)
bh = sim.op("mRhs") @ rng.uniform(-1, 1, sim.op("mRhs").shape[0])
combos = [(0.8, 1.9, 2), (0.85, 1.9, 2), (0.9, 1.9, 2), (0.95, 1.9, 2), (0.85, 1.7, 2), (0.85, 2.0, 2), (0.85, 1.9, 3), (0.9, 1.8, 3)]
for omega, cs, sw in combos:
    mg = mx.Multigrid(ctx, sim, "shifted", shift, sweeps=sw, omega=omega, coarse_scale=cs)
    A = mg.fine_matrix()
    b = mx.MultiVec.from_numpy(ctx, bh)
    x = mx.MultiVec(ctx, A.nrows, 1)
    its, conv, res = A.solve(b, x, method="cg", tol=1e-9, precond="mg", mg=mg)   # warm
    x.fill(0.0)
    ctx.sync(); t0 = time.time()
    its, conv, res = A.solve(b, x, method="cg", tol=1e-9, precond="mg", mg=mg)
    ctx.sync(); dt = time.time() - t0
    print("omega %.2f cscale %.2f sweeps %d: iters %d conv %s res %.2e time %.3f s  (%.2f ms/iter)" % (omega, cs, sw, its, conv, res, dt, 1e3 * dt / max(its, 1)), flush=True)
    del mg, A, b, x
