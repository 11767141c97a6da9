"""Diagnostic build of the persistent compressed-stream kernel: where one workgroup's iteration goes (s_memtime sums)."""
import os, sys
os.environ["MXG_CSP_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = mx.Ctx(0)
sim = mx.Sim.pillbox(n)
A = sim.to_device(ctx, "vecLapl", shift=0.05 * (2 * np.pi) ** 2)
X = mx.MultiVec(ctx, A.ncols, 1); Y = mx.MultiVec(ctx, A.nrows, 1); X.random(1)
A.compress()
for wgs in (1, 3):
    os.environ["MXG_CSP_WGS"] = str(wgs)
    print("wgs/CU", wgs, file=sys.stderr, flush=True)
    for _ in range(3):
        A.apply(X, Y)
    ctx.sync()
