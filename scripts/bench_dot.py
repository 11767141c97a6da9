"""dot / norm2 latency at the pillbox-256^3 vector length (host wall clock per call, reps calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maxwell_amd as mx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20608881
ctx = mx.Ctx(0)
W = mx.MultiVec(ctx, n, 1); W.random(2)
Z = mx.MultiVec(ctx, n, 1); Z.random(3)
for name, fn, nbytes in (("dot", lambda: W.dot(Z), 16 * n), ("norm2", lambda: W.norm2(), 8 * n)):
    for _ in range(5): fn()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(200): fn()
    us = (time.perf_counter() - t0) / 200 * 1e6
    print(name, "us/call %.1f" % us, "GB/s %.0f" % (nbytes / us / 1e3), "frac %.3f" % (nbytes / us / 1e3 / 8000), flush=True)
