"""Roofline check of the SpMV/SpMM kernel beyond the headline case:
real k = 1, 2, 4, 8, complex k = 1, 2 on the pillbox operator; the 13-nnz curlCurl and the scalar
Laplacian; the curved-cylinder (config 3) operator on all rows.  Algorithmic bytes per SURVEY §8d."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import maxwell_amd as mx
from maxwell_amd.api import _ptr

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = mx.Ctx(0)
reps = 40
out = []
shift = 0.05 * (2 * np.pi) ** 2


def timed(name, A, k, cx=False):
    X = mx.MultiVec(ctx, A.ncols, k, cx); X.random(1)
    Y = mx.MultiVec(ctx, A.nrows, k, cx)
    for _ in range(4): A.apply(X, Y)
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): A.apply(X, Y)
    ms = ctx.timer_stop() / reps
    nb = A.apply_bytes(k)
    out.append(dict(case=name, rows=A.nrows, nnz=A.nnz, k=k, complex=cx, ms=round(ms, 4), alg_GB=round(nb / 1e9, 3),
                    GBps=round(nb / ms / 1e6, 1), frac_of_8TBps=round(nb / ms / 1e6 / 8000, 3)))
    print(out[-1], flush=True)
    del X, Y


sim = mx.Sim.pillbox(N)
A = sim.to_device(ctx, "vecLapl", shift=shift)
for k in (1, 2, 4, 8):
    timed("pillbox %d^3 shiftedVecLapl" % N, A, k)
# complex copy of the same operator (interleaved values), as the loss-tangent / phase-shift decks use
nr, nc, nz, rp, ci, va = sim.op_raw("vecLapl", shift)
vals = np.zeros(2 * nz); vals[0::2] = np.ctypeslib.as_array(va, (nz,))
Ac = mx.CsrMatrix(ctx, nr, nc, C.cast(rp, C.c_void_p), C.cast(ci, C.c_void_p), _ptr(vals), True)
del A
for k in (1, 2):
    timed("pillbox %d^3 shiftedVecLapl (complex)" % N, Ac, k, True)
del Ac, vals
for name in ("curlCurl", "scaLapl"):
    B = sim.to_device(ctx, name)
    timed("pillbox %d^3 %s" % (N, name), B, 1)
    del B
sim.close()
# config 3: curved PEC cylinder without caps, periodic z, irregular wall rows
cyl = mx.Sim(N, origin=(-0.5,) * 3, size=(1.0,) * 3, pec_shape="cylinder(0.4;0,0,1;0,0,0)")
Acyl = cyl.to_device(ctx, "vecLapl", shift=shift)
timed("cylinder %d^3 shiftedVecLapl (config 3)" % N, Acyl, 1)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_spmm.json", "w"), indent=1)
