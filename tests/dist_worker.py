"""Worker of tests/test_distributed.py: one process per rank, gloo backend, CPU only.
Checks the distributed path end to end WITHOUT a GPU: box assembly + halo (product code) gives, on
every rank, exactly the owned rows of the global operator of the oracle, and the halo plan moves
the right x entries (ghost exchange emulated with torch.distributed send/recv)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from maxwell_amd import distributed as mxd
from maxwell_amd.api import PILLBOX_SHAPE
from oracle import yee, geometry as geo


def exchange(op, x_owned):
    """What mxg_csr_apply's halo step does, with gloo point-to-point instead of RCCL."""
    p = op.plan
    ghosts = np.zeros(op.nghost)
    reqs, bufs = [], []
    for i, peer in enumerate(p["neigh"]):
        s0, s1 = p["send_off"][i], p["send_off"][i + 1]
        r0, r1 = p["recv_off"][i], p["recv_off"][i + 1]
        if s1 > s0:
            reqs.append(dist.isend(torch.from_numpy(x_owned[p["send_idx"][s0:s1]].copy()), int(peer)))
        if r1 > r0:
            buf = torch.zeros(int(r1 - r0), dtype=torch.float64)
            bufs.append((r0, r1, buf))
            reqs.append(dist.irecv(buf, int(peer)))
    for r in reqs:
        r.wait()
    for r0, r1, buf in bufs:
        ghosts[r0:r1] = buf.numpy()
    return ghosts


def main():
    case = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if case == "pillbox":
        n = [12, 12, 12]
        kw = dict(origin=(-0.5,) * 3, size=(1.0,) * 3)
        ref = yee.YeeSim(n, pec=geo.pillbox(0.4, 0.8), **kw)
        sim = mxd.BoxSim(n, rank, world, pec_shape=PILLBOX_SHAPE, **kw)
    else:   # periodic vacuum box: ghosts wrap around the domain, every rank talks across the seam
        n = [12, 6, 6]
        ref = yee.YeeSim(n, strict=False)
        sim = mxd.BoxSim(n, rank, world, strict=False)
    o = ref.wave_ops()
    shift = 1.9739208802178716
    o["shifted"] = yee.shifted(o["vecLapl"], o["mRhs"], shift)
    fields = {"shifted": (ref.B, ref.B), "curlCurl": (ref.B, ref.B), "scaLapl": (ref.P, ref.P),
              "divB": (ref.P, ref.B), "gradPsi": (ref.B, ref.P), "curlE": (ref.B, ref.E)}
    rng = np.random.default_rng(7)
    total_rows = 0
    for name, (rowF, colF) in fields.items():
        op = mxd.LocalOperator(sim.dist_op(name, shift), n, sim.proc_grid, rank)
        op.agree_plan(dist)
        G = o[name].tocsr()
        # ownership: my rows are exactly the global rows whose cell lies in my box
        owner = mxd.owner_of_gids(rowF.gids, rowF.ncomp, n, sim.proc_grid)
        mine = rowF.gids[owner == rank]
        assert np.array_equal(op.rowgid, mine), (name, len(op.rowgid), len(mine))
        assert np.array_equal(op.colowned, colF.gids[mxd.owner_of_gids(colF.gids, colF.ncomp, n, sim.proc_grid) == rank])
        # row by row identical entries (values bit-exact, columns as global ids)
        L = op.scipy()
        local_to_gid = np.concatenate([op.colowned, op.ghost_gid])
        rows_global = np.searchsorted(rowF.gids, op.rowgid)
        for lr in rng.choice(op.nrows, size=min(op.nrows, 200), replace=False) if op.nrows else []:
            a0, a1 = L.indptr[lr], L.indptr[lr + 1]
            g0, g1 = G.indptr[rows_global[lr]], G.indptr[rows_global[lr] + 1]
            # the product keeps explicit zeros (like Epetra), scipy's diagonal matrices do not
            lk, gk = L.data[a0:a1] != 0, G.data[g0:g1] != 0
            lc = local_to_gid[L.indices[a0:a1]][lk]
            gc = colF.gids[G.indices[g0:g1]][gk]
            ol, og = np.argsort(lc), np.argsort(gc)
            assert np.array_equal(lc[ol], gc[og]), name
            assert np.array_equal(L.data[a0:a1][lk][ol], G.data[g0:g1][gk][og]), name
        # halo exchange + local product == owned rows of the global product
        x_global = np.random.default_rng(11).uniform(-1, 1, G.shape[1])       # same on every rank
        x_owned = x_global[np.searchsorted(colF.gids, op.colowned)]
        ghosts = exchange(op, x_owned)
        assert np.array_equal(ghosts, x_global[np.searchsorted(colF.gids, op.ghost_gid)]), name
        y_local = L @ np.concatenate([x_owned, ghosts])
        y_ref = (G @ x_global)[rows_global]
        assert np.abs(y_local - y_ref).max() <= 1e-12 * max(np.abs(y_ref).max(), 1.0), name
        total_rows += op.nrows
        # the all-reduced inner product the Krylov solvers need (C2)
        t = torch.tensor([float(np.dot(y_local, y_local))], dtype=torch.float64)
        dist.all_reduce(t)
        assert abs(t.item() - float(np.dot(G @ x_global, G @ x_global))) <= 1e-10 * t.item()
    assert total_rows > 0
    dist.barrier()
    if rank == 0:
        print("DIST_OK %s world=%d" % (case, world))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
