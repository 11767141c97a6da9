"""Persistent compressed-stream kernel: workgroups per CU x ring depth (environment knobs of launch_csp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = mx.Ctx(0)
sim = mx.Sim.pillbox(n)
A = sim.to_device(ctx, "vecLapl", shift=0.05 * (2 * np.pi) ** 2)
X = mx.MultiVec(ctx, A.ncols, 1); Y = mx.MultiVec(ctx, A.nrows, 1); X.random(1)
nb, nf, nbytes = A.compress()
own = nbytes + 16 * A.nrows
def timed():
    for _ in range(5): A.apply(X, Y)
    ctx.sync(); ctx.timer_start()
    for _ in range(50): A.apply(X, Y)
    return ctx.timer_stop() / 50
for wgs in (1, 2, 3, 4, 5):
    os.environ["MXG_CSP_WGS"] = str(wgs)
    ms = timed()
    print("wgs/CU %d: %.4f ms  %.0f GB/s own bytes (%.3f of 8 TB/s)" % (wgs, ms, own / ms / 1e6, own / ms / 8e9), flush=True)
