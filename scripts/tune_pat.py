"""Pattern-dictionary product (MXG_PAT=1): one thread per row (MXG_PATW=0) against the wave-persistent kernel with
cached per-lane patterns (default), workgroups per CU; each against the CSR kernel's result (bit-identical) and time."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "worker":
    import numpy as np
    import maxwell_amd as mx
    n = int(sys.argv[2]); op = sys.argv[3]
    ctx = mx.Ctx(0)
    sim = mx.Sim.pillbox(n)
    A = sim.to_device(ctx, op, shift=0.05 * (2 * np.pi) ** 2) if op == "vecLapl" else sim.to_device(ctx, op)
    X = mx.MultiVec(ctx, A.ncols, 1); Y = mx.MultiVec(ctx, A.nrows, 1); X.random(1)
    A.apply(X, Y); ref = Y.to_numpy().copy()
    t0 = time.time(); A.compress(); tc = time.time() - t0
    Y.fill(0.0); A.apply(X, Y)
    same = bool(np.array_equal(Y.to_numpy(), ref))
    for _ in range(5): A.apply(X, Y)
    ctx.sync(); ctx.timer_start()
    for _ in range(50): A.apply(X, Y)
    ms = ctx.timer_stop() / 50
    print("RESULT %s n=%d rows=%d %s: %.4f ms, compress %.1f s, bit-identical to the CSR kernel: %s" %
          (op, n, A.nrows, " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("MXG_")), ms, tc, same), flush=True)
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "256"
    for op in ("vecLapl", "scaLapl"):
        for env in ({}, {"MXG_TILE_SPREAD": "1"}):
            r = subprocess.run([sys.executable, __file__, "worker", n, op], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            out = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            print(out[0] if out else "FAILED rc=%d %s" % (r.returncode, (r.stdout + r.stderr)[-600:]), flush=True)
            for l in r.stderr.splitlines():
                if l.startswith("[mxg_pat]"): print("   " + l, flush=True)
