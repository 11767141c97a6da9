"""Roofline check of the multivector block kernels (K2-K7) at the pillbox-256^3 vector length."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20608881
ctx = mx.Ctx(0)
reps = 30
out = []

def timed(name, fn, nbytes):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): fn()
    ms = ctx.timer_stop() / reps
    out.append(dict(kernel=name, ms=round(ms, 4), alg_GB=round(nbytes / 1e9, 3), GBps=round(nbytes / ms / 1e6, 1),
                    frac_of_8TBps=round(nbytes / ms / 1e6 / 8000, 3)))
    print(out[-1], flush=True)

for m, k in ((30, 1), (30, 4), (100, 1), (10, 1), (30, 8)):
    V = mx.MultiVec(ctx, n, m); V.random(1)
    W = mx.MultiVec(ctx, n, k); W.random(2)
    Z = mx.MultiVec(ctx, n, k); Z.random(3)
    B = np.random.default_rng(0).uniform(-1, 1, (m, k))
    if m == 30 and k == 1:
        timed("dot k=1", lambda: W.dot(Z), 16 * n)
        timed("norm2 k=1", lambda: W.norm2(), 8 * n)
        timed("axpby k=1", lambda: W.update(0.5, Z, 2.0), 24 * n)
        timed("add_mv k=1", lambda: W.add_mv(0.5, Z, 2.0, W), 24 * n)
        timed("scale k=1", lambda: W.scale(1.0000001), 16 * n)
    timed("trans_mv m=%d k=%d" % (m, k), lambda: W.trans_mv(1.0, V), 8 * n * (m + k))
    timed("times_mat m=%d k=%d (beta=1)" % (m, k), lambda: W.times_mat_add_mv(1e-9, V, B, 1.0), 8 * n * (m + 2 * k))
    del V, W, Z
outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(outdir, exist_ok=True)
json.dump(out, open(os.path.join(outdir, "bench_mv.json"), "w"), indent=1)
