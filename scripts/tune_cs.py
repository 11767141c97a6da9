"""A/B: plain CSR row-gather kernel vs the compressed-stream kernel (mxg_csr_compress) at the headline size."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = mx.Ctx(0)
sim = mx.Sim.pillbox(n)
out = []
only = sys.argv[2] if len(sys.argv) > 2 else None
for name, shift in (("vecLapl", 0.05 * (2 * np.pi) ** 2), ("scaLapl", None)):
    if only and name != only:
        continue
    A = sim.to_device(ctx, name, shift=shift)
    X = mx.MultiVec(ctx, A.ncols, 1); Y = mx.MultiVec(ctx, A.nrows, 1); Z = mx.MultiVec(ctx, A.nrows, 1); X.random(1)
    alg = A.apply_bytes(1)
    def timed():
        for _ in range(5): A.apply(X, Y)
        ctx.sync(); ctx.timer_start()
        for _ in range(100): A.apply(X, Y)
        return ctx.timer_stop() / 100
    ms0 = timed()
    A.apply(X, Z)
    t0 = time.time(); nb, nf, nbytes = A.compress(); tc = time.time() - t0       # default: pattern dictionary
    ms1 = timed()
    same = bool(np.array_equal(Y.to_numpy(), Z.to_numpy()))
    other = {}
    for mode, (pat, csp) in (("cs_one_record_per_wg", ("0", "0")), ("csp_persistent_loader", ("0", "1"))):
        os.environ["MXG_PAT"] = pat; os.environ["MXG_CSP"] = csp
        A.compress(); other[mode + "_ms"] = round(timed(), 4)
        same = same and bool(np.array_equal(Y.to_numpy(), Z.to_numpy()))
    os.environ.pop("MXG_PAT"); os.environ.pop("MXG_CSP")
    pat_bytes = 2 * A.nrows + 16 * A.nrows                      # row ids + x + y (dictionary and individual rows are noise)
    rec = dict(op=name, rows=A.nrows, nnz=A.nnz, csr_ms=round(ms0, 4), pat_ms=round(ms1, 4), speedup=round(ms0 / ms1, 3),
               pat_own_bytes=pat_bytes, pat_own_GBps=round(pat_bytes / ms1 / 1e6, 1), pat_frac_of_8TBps=round(pat_bytes / ms1 / 8e9, 3),
               **other,
               csr_GBps=round(alg / ms0 / 1e6, 1), pat_effective_GBps=round(alg / ms1 / 1e6, 1), blocks=nb, fallback_blocks=nf,
               packed_stream_bytes=nbytes, csr_matrix_bytes=A.nnz * 12, compress_seconds=round(tc, 2), bit_identical=same)
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/tune_cs.json", "w"), indent=1)
