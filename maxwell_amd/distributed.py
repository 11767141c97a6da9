"""One GPU per MxGridDomain box: partition, row-partitioned operators, halo plan.

Mirrors what Epetra does for the reference under MPI (src/MxGrid.cpp:100-193 partition,
src/MxGridField.cpp:274-293 row ownership, the Epetra_Import built by FillComplete): every rank
owns the rows of its cell box, needs the x entries of neighbouring boxes that its rows touch
("ghosts"), and receives them before each product.  Here

  * the box operators come from the host assembler run on the box + a two-cell halo, so no
    communication is needed to build the matrices (mxh_sim_create_box / mxh_sim_dist_op);
  * the halo plan (who sends which owned entries to whom) is agreed once at setup with
    torch.distributed object collectives (gloo on CPU, RCCL on the GPUs);
  * the per-product exchange is ncclSend/ncclRecv inside mxg_csr_apply (C1), the inner products
    are ncclAllReduce inside mxg_mv_dot / norm2 / trans_mv (C2).
"""
import ctypes as C

import numpy as np

from . import _lib
from .api import Sim, CsrMatrix, _chk, _ptr, PILLBOX_SHAPE


def partition(n, nranks, rank):
    """(proc_grid, box_lo, box_hi) of `rank` -- MxGrid::partition / getGridDomain."""
    lib = _lib.load_mxhost()
    nn = (C.c_int * 3)(*[int(v) for v in n])
    pg, lo, hi = (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)()
    _chk(lib.mxh_partition(nn, int(nranks), int(rank), pg, lo, hi), lib, "mxh")
    return list(pg), list(lo), list(hi)


def owner_of_gids(gids, ncomp, n, proc_grid):
    """Rank owning each global id (cell = gid // ncomp on the (N+1)^3 node lattice)."""
    gids = np.asarray(gids, dtype=np.int64)
    cell = gids // ncomp
    ni = [int(v) + 1 for v in n]
    k = cell % ni[2]
    j = (cell // ni[2]) % ni[1]
    i = cell // (ni[2] * ni[1])
    pid = np.zeros(len(gids), dtype=np.int64)
    factor = 1
    for ax, c in ((2, k), (1, j), (0, i)):
        std = int(np.floor(float(n[ax]) / float(proc_grid[ax]) + 0.5))
        p = np.minimum(c // max(std, 1), proc_grid[ax] - 1)
        pid += p * factor
        factor *= proc_grid[ax]
    return pid


class BoxSim(Sim):
    """Sim restricted to the cell box of one rank."""

    def __init__(self, n, rank, nranks, origin=(0.0, 0.0, 0.0), size=(1.0, 1.0, 1.0),
                 lower=("periodic",) * 3, upper=("periodic",) * 3, pec_shape=None, strict=True, dielectrics=None,
                 dm_frac=0.0):
        n = [int(n)] * 3 if np.ndim(n) == 0 else [int(v) for v in n]
        self.rank, self.nranks = rank, nranks
        self.proc_grid, self.box_lo, self.box_hi = partition(n, nranks, rank)
        Sim.__init__(self, n, origin=origin, size=size, lower=lower, upper=upper, pec_shape=pec_shape, strict=strict,
                     dielectrics=dielectrics, dm_frac=dm_frac, box=(self.box_lo, self.box_hi))

    def dist_op(self, name, shift=0.0):
        """Owned rows with GLOBAL column ids: dict(rowptr, colgid, vals, rowgid, colowned, ncomp)."""
        nr, nz, nown, ncomp = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        rp, cg, rg, co = _lib.c_i64p(), _lib.c_i64p(), _lib.c_i64p(), _lib.c_i64p()
        va = _lib.c_dp()
        _chk(self.lib.mxh_sim_dist_op(self.h, name.encode(), float(shift), C.byref(nr), C.byref(nz),
                                      C.byref(rp), C.byref(cg), C.byref(va), C.byref(rg), C.byref(nown),
                                      C.byref(co), C.byref(ncomp)), self.lib, "mxh")
        def arr(p, n):
            if n == 0 or not p:
                return np.zeros(0, dtype=p._type_)
            return np.ctypeslib.as_array(p, (n,)).copy()
        return dict(rowptr=arr(rp, nr.value + 1), colgid=arr(cg, nz.value), vals=arr(va, nz.value),
                    rowgid=arr(rg, nr.value), colowned=arr(co, nown.value), ncomp=ncomp.value)


class LocalOperator:
    """Row-partitioned operator in local numbering: columns [0, nowned) are owned entries (ascending
    global id), ghosts follow grouped by owner rank (ascending rank, then ascending global id)."""

    def __init__(self, op, n, proc_grid, rank):
        self.rank = rank
        self.rowgid, self.colowned = op["rowgid"], op["colowned"]
        colgid = op["colgid"]
        owner = owner_of_gids(colgid, op["ncomp"], n, proc_grid) if len(colgid) else np.zeros(0, np.int64)
        is_ghost = owner != rank
        ghosts = np.unique(colgid[is_ghost])
        gowner = owner_of_gids(ghosts, op["ncomp"], n, proc_grid) if len(ghosts) else np.zeros(0, np.int64)
        order = np.lexsort((ghosts, gowner))
        self.ghost_gid, self.ghost_owner = ghosts[order], gowner[order]
        nown = len(self.colowned)
        local = np.empty(len(colgid), dtype=np.int64)
        pos = np.searchsorted(self.colowned, colgid[~is_ghost])
        if np.any(self.colowned[np.minimum(pos, max(nown - 1, 0))] != colgid[~is_ghost]):
            raise RuntimeError("an owned column is missing from the owned map of the column field")
        local[~is_ghost] = pos
        # ghost slot = position in the (owner, gid)-sorted ghost list
        key_sorted = self.ghost_owner * (1 << 40) + self.ghost_gid
        key = owner[is_ghost] * (1 << 40) + colgid[is_ghost]
        local[is_ghost] = nown + np.searchsorted(key_sorted, key)
        self.rowptr = np.ascontiguousarray(op["rowptr"], dtype=np.int64)
        self.colidx = np.ascontiguousarray(local, dtype=np.int32)
        self.vals = np.ascontiguousarray(op["vals"], dtype=np.float64)
        self.nrows, self.nowned, self.nghost = len(self.rowgid), nown, len(self.ghost_gid)
        # what I need from whom: {owner: ascending gids}
        self.needs = {int(p): self.ghost_gid[self.ghost_owner == p] for p in np.unique(self.ghost_owner)}
        self.plan = None

    def agree_plan(self, dist):
        """Exchange the need lists (setup-time collective) and build the halo plan."""
        world = dist.get_world_size()
        gathered = [None] * world
        dist.all_gather_object(gathered, self.needs)
        neigh = sorted(set(self.needs) | {p for p in range(world) if self.rank in gathered[p]})
        send_off, send_idx, recv_off = [0], [], [0]
        for p in neigh:
            wanted = gathered[p].get(self.rank, np.zeros(0, np.int64))
            pos = np.searchsorted(self.colowned, wanted)
            if len(wanted) and np.any(self.colowned[np.minimum(pos, self.nowned - 1)] != wanted):
                raise RuntimeError("rank %d asks rank %d for entries it does not own" % (p, self.rank))
            send_idx.append(pos)
            send_off.append(send_off[-1] + len(wanted))
            recv_off.append(recv_off[-1] + len(self.needs.get(p, ())))
        self.plan = dict(neigh=np.array(neigh, dtype=np.int32),
                         send_off=np.array(send_off, dtype=np.int64),
                         send_idx=np.concatenate(send_idx).astype(np.int64) if send_idx else np.zeros(0, np.int64),
                         recv_off=np.array(recv_off, dtype=np.int64))
        assert self.plan["recv_off"][-1] == self.nghost
        return self.plan

    def scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.vals, self.colidx, self.rowptr), shape=(self.nrows, self.nowned + self.nghost))

    def to_device(self, ctx):
        p = self.plan
        hp = None
        keep = []
        if p is not None and len(p["neigh"]) > 0:
            neigh = np.ascontiguousarray(p["neigh"], dtype=np.int32)
            so, si, ro = [np.ascontiguousarray(p[k], dtype=np.int64) for k in ("send_off", "send_idx", "recv_off")]
            if len(si) == 0:
                si = np.zeros(1, np.int64)
            keep = [neigh, so, si, ro]
            hp = _lib.HaloPlan(len(neigh), neigh.ctypes.data_as(C.POINTER(C.c_int)),
                               so.ctypes.data_as(_lib.c_i64p), si.ctypes.data_as(_lib.c_i64p),
                               ro.ctypes.data_as(_lib.c_i64p))
        A = CsrMatrix(ctx, self.nrows, self.nowned, _ptr(self.rowptr), _ptr(self.colidx), _ptr(self.vals),
                      False, nghost=self.nghost, plan=C.byref(hp) if hp is not None else None)
        A._keep = keep
        return A


class DistOps:
    """The C++ route to the same matrices (mxh_dist_ops_*): MxEMSim maps of the owned entries and MxCrsMatrix, which
    agrees the ghost import with the other ranks through the collectives of the ctx (RCCL, or the in-process
    world) - no torch.distributed involved.  Every call is collective."""

    def __init__(self, ctx, sim):
        self.lib, self.ctx, self.sim = _lib.load_mxhost(), ctx, sim
        h = C.c_void_p()
        _chk(self.lib.mxh_dist_ops_create(ctx.h, sim.h, C.byref(h)), self.lib, "mxh")
        self.h = h

    def matrix(self, name, shift=0.0):
        return DistMatrix(self, name, shift)

    def close(self):
        if self.h:
            self.lib.mxh_dist_ops_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DistMatrix:
    """Owned rows of a named operator on the device (borrowed from its DistOps)."""

    def __init__(self, ops, name, shift=0.0):
        if not isinstance(ops, DistOps):            # DistMatrix(ctx, sim, name, shift)
            raise TypeError("DistMatrix needs a DistOps")
        self.ops, self.ctx, lib = ops, ops.ctx, ops.lib
        A = C.c_void_p()
        nr, nc, ng = C.c_int64(), C.c_int64(), C.c_int64()
        rg, cg = _lib.c_i64p(), _lib.c_i64p()
        _chk(lib.mxh_dist_ops_matrix(ops.h, name.encode(), float(shift), C.byref(A), C.byref(nr), C.byref(rg),
                                     C.byref(nc), C.byref(cg), C.byref(ng)), lib, "mxh")
        self.h = A
        self.nrows, self.ncols_owned, self.nghost = nr.value, nc.value, ng.value
        self.row_gids = np.ctypeslib.as_array(rg, (nr.value,)).copy() if nr.value else np.zeros(0, np.int64)
        self.col_gids = np.ctypeslib.as_array(cg, (nc.value,)).copy() if nc.value else np.zeros(0, np.int64)

    def apply(self, x, y):
        _chk(self.ctx.lib.mxg_csr_apply(self.ctx.h, self.h, x.h, y.h), self.ctx.lib)

    def apply_bytes(self, k=1):
        return self.ctx.lib.mxg_csr_apply_bytes(self.h, k)

    def compress(self):
        nb, nf, sb = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(self.ctx.lib.mxg_csr_compress(self.h, C.byref(nb), C.byref(nf), C.byref(sb)), self.ctx.lib)
        return nb.value, nf.value, sb.value

    def product_form(self):
        """(form, tiles, tile patterns, irregular rows, coverage): form 0 plain CSR, 1 compressed stream, 3 tile patterns."""
        t, p, g, cov = C.c_int64(), C.c_int64(), C.c_int64(), C.c_double()
        form = self.ctx.lib.mxg_csr_product_form(self.h, C.byref(t), C.byref(p), C.byref(g), C.byref(cov))
        return form, t.value, p.value, g.value, cov.value


def bench_setup(ctx, n_local, shift):
    """Weak-scaling workload of bench.py: the pillbox cavity of example/pillbox.py on a grid refined so
    that every rank owns an n_local^3-cell box (N = 8: 512^3 Yee cells on 2 x 2 x 2 boxes)."""
    import torch.distributed as dist
    world, rank = ctx.nranks, ctx.rank
    pg, _, _ = partition([n_local] * 3, world, rank)      # cube: the grid shape does not change pg
    n = [n_local * p for p in pg]
    sim = BoxSim(n, rank, world, origin=(-0.5,) * 3, size=(1.0,) * 3, pec_shape=PILLBOX_SHAPE)
    op = LocalOperator(sim.dist_op("shifted", shift), n, sim.proc_grid, rank)
    op.agree_plan(dist)
    A = op.to_device(ctx)
    workload = ("pillbox %dx%dx%d Yee cells on %dx%dx%d boxes of %d^3, vecLapl - shift*dmA, "
                "RCCL halo exchange per SpMV" % (n[0], n[1], n[2], pg[0], pg[1], pg[2], n_local))
    return sim, A, workload, "%d GPUs, %dx%dx%d MxGridDomain boxes" % (world, pg[0], pg[1], pg[2])
