"""Times of the product kernels on the pillbox operator at n^3: CSR kernel, compressed stream, tile form (k = 1 and a
block of vectors), plus the stream ceiling of the device; every form checked bit-identical to the CSR kernel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = 50
ctx = mx.Ctx(0)
sim = mx.Sim.pillbox(n)
shift = 0.05 * (2 * np.pi) ** 2
out = {"cells": n}


def timed(A, X, Y, reps=reps):
    for _ in range(5): A.apply(X, Y)
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): A.apply(X, Y)
    return ctx.timer_stop() / reps


for name in ("vecLapl",):
    A = sim.to_device(ctx, name, shift=shift if name == "vecLapl" else 0.0)
    X = mx.MultiVec(ctx, A.ncols, 1); Y = mx.MultiVec(ctx, A.nrows, 1); X.random(1)
    alg = A.apply_bytes(1)
    r = {"rows": A.nrows, "nnz": A.nnz, "csr_ms": timed(A, X, Y)}
    r["csr_frac"] = alg / r["csr_ms"] / 8e9
    ref = Y.to_numpy()
    r["stream_ceiling_ms"] = ctx.stream_ceiling_ms(alg - A.nrows * 8, A.nrows * 8)
    for key, pat in (("cs", "0"), ("tile", "1")):
        os.environ["MXG_PAT"] = pat
        B = sim.to_device(ctx, name, shift=shift if name == "vecLapl" else 0.0)
        nb, nf, nbytes = B.compress()
        os.environ.pop("MXG_PAT")
        Yb = mx.MultiVec(ctx, B.nrows, 1)
        r[key + "_ms"] = timed(B, X, Yb)
        r[key + "_same"] = bool(np.array_equal(ref, Yb.to_numpy()))
        r[key + "_bytes"] = nbytes
        r[key + "_form"] = B.product_form()
        if key == "tile":
            moved = nbytes + 8 * (B.nrows + B.ncols)
            r["tile_hbm_frac"] = moved / r["tile_ms"] / 8e9
            for k in (2, 4, 8):
                Xk = mx.MultiVec(ctx, B.ncols, k); Yk = mx.MultiVec(ctx, B.nrows, k); Xk.random(3)
                ms = timed(B, Xk, Yk, 20)
                A.apply(Xk, Yk); want = Yk.to_numpy(); Yk.fill(float('nan')); B.apply(Xk, Yk)
                r["tile_k%d" % k] = {"ms": ms, "eff_frac_csr_formula": B.apply_bytes(k) / ms / 8e9,
                                     "same": bool(np.array_equal(want, Yk.to_numpy())), "csr_ms": timed(A, Xk, Yk, 20)}
                Xk.close(); Yk.close()
        B.close(); Yb.close()
    out[name] = r
    print(name, json.dumps(r), flush=True)
    A.close(); X.close(); Y.close()
print(json.dumps(out))
