"""A/B of the two row-gather kernels on blocks of vectors (pillbox shifted vector Laplacian)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = mx.Ctx(0)
sim = mx.Sim.pillbox(N)
A = sim.to_device(ctx, "vecLapl", shift=0.05 * (2 * np.pi) ** 2)
for k in (1, 2, 4, 8):
    X = mx.MultiVec(ctx, A.ncols, k); X.random(1); Y = mx.MultiVec(ctx, A.nrows, k)
    nb = A.apply_bytes(k)
    for code, name in ((0x2750000, "persistent rgp"), (0x750000, "one block per workgroup")):
        A.set_kernel(code)
        for _ in range(4): A.apply(X, Y)
        ctx.sync(); ctx.timer_start()
        for _ in range(30): A.apply(X, Y)
        ms = ctx.timer_stop() / 30
        print("k=%d %-26s %.4f ms  %.3f of 8 TB/s" % (k, name, ms, nb / ms / 8e9), flush=True)
    del X, Y
