"""A/B the SpMV kernel variants in one process (interleaved, HIP-event timed)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
codes = [int(c, 0) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 2, 4, 6, 1, 0x10, 0x20, 0x30]
reps = 100
ctx = mx.Ctx(0)
sim = mx.Sim.pillbox(n)
A = sim.to_device(ctx, "vecLapl", shift=0.05 * (2 * np.pi) ** 2)
X = mx.MultiVec(ctx, A.ncols, 1); Y = mx.MultiVec(ctx, A.nrows, 1); X.random(1)
alg = A.apply_bytes(1)
names = {0x750000: "row-gather (r01 default)", 0x2750000: "persistent row-gather, LDS-DMA nt", 0x6750000: "persistent row-gather, LDS-DMA", 0x4000: "xwin 256x8", 0x8000: "xwin 256x4", 0: "stream nt+xcd 256x8", 2: "stream plain-loads", 4: "stream no-xcd", 6: "stream plain,no-xcd", 1: "vector8",
         0x10: "stream 512x8", 0x20: "stream 256x4", 0x30: "stream 256x16"}
res = {}
for rnd in range(3):
    for code in codes:
        A.set_kernel(code)
        for _ in range(5): A.apply(X, Y)
        ctx.sync(); ctx.timer_start()
        for _ in range(reps): A.apply(X, Y)
        ms = ctx.timer_stop() / reps
        res.setdefault(code, []).append(ms)
for code in codes:
    ms = min(res[code])
    print("%-24s code %#4x  %.4f ms  %.1f GB/s  (%.1f%% of 8 TB/s)  runs %s" % (
        names.get(code, "?"), code, ms, alg / ms / 1e6, alg / ms / 1e6 / 80.0, ["%.4f" % v for v in res[code]]))
