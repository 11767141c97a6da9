"""Halo overhead rehearsal on ONE GPU: rank 0's operator of the N-rank weak-scaling workload of bench.py
(256^3 cells per rank), with every neighbour replaced by a 1-rank RCCL self-communicator: the ghost counts,
pack kernel, grouped ncclSend/ncclRecv per neighbour, interior/boundary split and stream overlap are the real
ones; only the ghost VALUES come from rank 0 itself (timing only, no parity).  Compares with the same rows
without any exchange (ghost columns folded onto owned ones)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx
from maxwell_amd import distributed as mxd
from maxwell_amd.api import PILLBOX_SHAPE, CsrMatrix, _ptr

n_local = int(sys.argv[1]) if len(sys.argv) > 1 else 256
worlds = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 8]
shift = 0.05 * (2 * np.pi) ** 2
ctx = mx.Ctx(0, 0, 1, mx.Ctx.unique_id())
out = []
for world in worlds:
    pg, _, _ = mxd.partition([n_local] * 3, world, 0)
    n = [n_local * p for p in pg]
    sim = mxd.BoxSim(n, 0, world, origin=(-0.5,) * 3, size=(1.0,) * 3, pec_shape=PILLBOX_SHAPE)
    op = mxd.LocalOperator(sim.dist_op("shifted", shift), n, sim.proc_grid, 0)
    # self-communication plan: one "neighbour" (rank 0) per real neighbour, same ghost counts both ways
    owners = sorted(op.needs)
    send_off, recv_off, send_idx = [0], [0], []
    boundary = np.unique(op.colidx[op.colidx < op.nowned][:1])   # placeholder, replaced below
    for p in owners:
        cnt = len(op.needs[p])
        send_idx.append(np.linspace(0, op.nowned - 1, cnt).astype(np.int64))    # any owned entries: timing only
        send_off.append(send_off[-1] + cnt); recv_off.append(recv_off[-1] + cnt)
    op.plan = dict(neigh=np.zeros(len(owners), dtype=np.int32), send_off=np.array(send_off, dtype=np.int64),
                   send_idx=np.concatenate(send_idx), recv_off=np.array(recv_off, dtype=np.int64))
    A = op.to_device(ctx)
    # same rows without an exchange: ghost columns folded onto owned columns
    folded = np.where(op.colidx >= op.nowned, op.colidx % op.nowned, op.colidx).astype(np.int32)
    B = CsrMatrix(ctx, op.nrows, op.nowned, _ptr(op.rowptr), _ptr(folded), _ptr(op.vals), False)
    X = mx.MultiVec(ctx, op.nowned, 1); Y = mx.MultiVec(ctx, op.nrows, 1); X.random(1)
    def timed(M):
        for _ in range(5): M.apply(X, Y)
        ctx.sync(); ctx.timer_start()
        for _ in range(100): M.apply(X, Y)
        return ctx.timer_stop() / 100
    t_plain, t_halo = timed(B), timed(A)
    rec = dict(world=world, proc_grid=list(pg), rows=op.nrows, nnz=int(op.rowptr[-1]), ghosts=op.nghost, neighbours=len(owners),
               spmv_no_exchange_ms=round(t_plain, 4), spmv_with_halo_ms=round(t_halo, 4),
               halo_overhead_percent=round(100 * (t_halo / t_plain - 1), 1), ghost_bytes=op.nghost * 8)
    print(json.dumps(rec), flush=True)
    out.append(rec)
    A.close(); B.close(); X.close(); Y.close(); sim.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/halo_selfcomm.json", "w"), indent=1)
ctx.close()
