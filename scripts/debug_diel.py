import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse.linalg as spla
import maxwell_amd as mx
ctx = mx.Ctx(0)
p = mx.Problem(path="tests/golden/dsphmsph.mx"); sim = p.sim()
A = sim.to_device(ctx, "vecLapl", shift=0.5)
As = sim.op("vecLapl", shift=0.5).tocsc()
mask = sim.used_mask()
rng = np.random.default_rng(0)
b = rng.uniform(-1, 1, A.nrows) * mask
keep = np.nonzero(mask)[0]
xs = np.zeros(A.nrows); xs[keep] = spla.spsolve(As[keep][:, keep], b[keep])
for method, prec in (("gmres", "mg"),):
    B = mx.MultiVec.from_numpy(ctx, b.reshape(-1, 1))
    X = mx.MultiVec(ctx, A.nrows, 1); X.fill(0.0)
    mg = mx.Multigrid(ctx, sim, "shifted", shift=0.5) if prec == "mg" else None
    it, conv, res = A.solve(B, X, method=method, tol=1e-10, restart=40, max_iters=4000, precond=prec, mg=mg)
    x = X.to_numpy()[:, 0]
    print(method, prec, "iters", it, "conv", conv, "res", res, "err vs spsolve", np.abs(x - xs).max() / np.abs(xs).max(), flush=True)
s = p.solver(ctx, verbose=1)
print("nconv", s.solve())
for i in range(s.nconv): print(s.result(i))
v = s.eigvec(0, A.nrows)
M = sim.op("mRhs").diagonal()
N = (mask == 0) & (M > 0)
print("eigvec norm on decoupled rows", np.linalg.norm(v[N]), "on used rows", np.linalg.norm(v[mask == 1]), "on M==0 rows", np.linalg.norm(v[M == 0]))
L = sim.op("vecLapl")
print("||L v||", np.linalg.norm(L @ v), "||M v||", np.linalg.norm(M * v))
big = np.argsort(-np.abs(v))[:8]
print("largest entries rows", big, v[big], "mask", mask[big], "M", M[big], "L diag", L.diagonal()[big])
