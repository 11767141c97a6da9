"""A/B: plain CSR row-gather kernel vs the compressed-stream kernel (mxg_csr_compress) at the headline size."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = mx.Ctx(0)
sim = mx.Sim.pillbox(n)
out = []
for name, shift in (("vecLapl", 0.05 * (2 * np.pi) ** 2), ("scaLapl", None)):
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
    t0 = time.time(); nb, nf, nbytes = A.compress(); tc = time.time() - t0
    ms1 = timed()
    same = bool(np.array_equal(Y.to_numpy(), Z.to_numpy()))
    # the one-block-per-workgroup kernel on the same records (MXG_CSP=0 at compress time clears the persistent bit)
    os.environ["MXG_CSP"] = "0"; A.compress(); ms_old = timed(); same = same and bool(np.array_equal(Y.to_numpy(), Z.to_numpy()))
    os.environ.pop("MXG_CSP")
    cs_bytes = nbytes + 16 * A.nrows                       # packed records (incl. row offsets) + x + y
    rec = dict(op=name, rows=A.nrows, nnz=A.nnz, csr_ms=round(ms0, 4), cs_ms=round(ms1, 4), speedup=round(ms0 / ms1, 3),
               cs_one_block_per_wg_ms=round(ms_old, 4), cs_own_bytes=cs_bytes, cs_own_GBps=round(cs_bytes / ms1 / 1e6, 1),
               cs_frac_of_8TBps=round(cs_bytes / ms1 / 1e6 / 8000.0, 3),
               csr_GBps=round(alg / ms0 / 1e6, 1), cs_effective_GBps=round(alg / ms1 / 1e6, 1), blocks=nb, fallback_blocks=nf,
               packed_stream_bytes=nbytes, csr_matrix_bytes=A.nnz * 12, compress_seconds=round(tc, 2), bit_identical=same)
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/tune_cs.json", "w"), indent=1)
