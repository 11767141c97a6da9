"""Python mirror of the reference's host interface for the hot path, over the C ABI.

Names follow the reference: CsrMatrix.apply = MxCrsMatrix::apply, MultiVec methods =
MxMultiVector / MxAnasaziMV methods (src/MxMultiVector.hpp:14-109, src/MxAnasaziMV.hpp:23-132),
Sim = MxEMSim + MxEMOps, Solver = MxSolver.  Every call goes to the HIP library; numpy is only
used to move data in and out.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MxError


def _chk(rc, lib, which="mxg"):
    if rc != 0:
        msg = (lib.mxg_last_error() if which == "mxg" else lib.mxh_last_error()) or b""
        raise MxError(msg.decode(errors="replace"))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class LocalWorld:
    """In-process world (mxg_local_world): `nranks` rank threads of this process share one device, with the
    same collectives as the RCCL world.  `run(fn)` starts one thread per rank with its own Ctx and returns the
    list of results; an exception in any rank is re-raised."""

    def __init__(self, nranks, device=0):
        self.lib = _lib.load_mxg()
        self.nranks, self.device = nranks, device
        h = C.c_void_p()
        _chk(self.lib.mxg_local_world_create(nranks, C.byref(h)), self.lib)
        self.h = h

    def run(self, fn):
        import threading
        out, err = [None] * self.nranks, [None] * self.nranks

        def body(r):
            ctx = None
            try:
                ctx = Ctx(self.device, r, self.nranks, world=self)
                out[r] = fn(ctx)
            except BaseException as e:          # noqa: BLE001 - reported to the caller below
                err[r] = e
            finally:
                if ctx is not None:
                    ctx.close()

        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.nranks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        failed = [(r, e) for r, e in enumerate(err) if e is not None]
        if failed:
            # the rank that failed first matters: the others usually only report that it never reached a collective
            first = [f for f in failed if "did not reach the collective" not in str(f[1])] or failed
            raise MxError("rank %d: %s: %s" % (first[0][0], type(first[0][1]).__name__, first[0][1])) from first[0][1]
        return out

    def close(self):
        if self.h:
            self.lib.mxg_local_world_destroy(self.h)
            self.h = None


class Ctx:
    """One GPU / rank (replaces MxComm, src/MxComm.hpp:14-37)."""

    def __init__(self, device=0, rank=0, nranks=1, unique_id=None, world=None):
        self.lib = _lib.load_mxg()
        h = C.c_void_p()
        uid = None
        self.world = world
        if world is not None:            # in-process world: this rank is a thread sharing the device
            _chk(self.lib.mxg_ctx_create_local(device, rank, nranks, world.h, C.byref(h)), self.lib)
            self.h = h
            self.rank, self.nranks = rank, nranks
            return
        if unique_id is not None:
            self._uid = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
            uid = C.cast(self._uid, C.c_void_p)
        _chk(self.lib.mxg_ctx_create(device, rank, nranks, uid, C.byref(h)), self.lib)
        self.h = h
        self.rank, self.nranks = rank, nranks

    @staticmethod
    def unique_id():
        lib = _lib.load_mxg()
        buf = (C.c_char * 128)()
        _chk(lib.mxg_get_unique_id(C.cast(buf, C.c_void_p)), lib)
        return bytes(buf)

    def comm_ranks(self):
        """Ranks RCCL itself reports for this context's communicator (ncclCommCount); 1 without one."""
        return int(self.lib.mxg_ctx_comm_ranks(self.h))

    def sync(self):
        _chk(self.lib.mxg_ctx_sync(self.h), self.lib)

    def barrier(self):
        _chk(self.lib.mxg_barrier(self.h), self.lib)

    def stream_ceiling_ms(self, read_bytes, write_bytes, reps=20):
        """ms per pass of a pure stream with this byte mix on this device (mxg_bench_stream)."""
        ms = C.c_float()
        _chk(self.lib.mxg_bench_stream(self.h, int(read_bytes), int(write_bytes), reps, C.byref(ms)), self.lib)
        return ms.value

    def timer_start(self):
        _chk(self.lib.mxg_timer_start(self.h), self.lib)

    def timer_stop(self):
        ms = C.c_float()
        _chk(self.lib.mxg_timer_stop(self.h, C.byref(ms)), self.lib)
        return ms.value

    def close(self):
        if self.h:
            self.lib.mxg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiVec:
    """MxMultiVector / MxAnasaziMV on the device (column-major, complex interleaved)."""

    def __init__(self, ctx, nlocal=None, k=1, is_complex=False, _handle=None, _parent=None):
        self.ctx, self.lib = ctx, ctx.lib
        self._parent = _parent   # keep the owner alive while a view exists
        if _handle is None:
            h = C.c_void_p()
            _chk(self.lib.mxg_mv_create(ctx.h, int(nlocal), int(k), int(bool(is_complex)), C.byref(h)),
                 self.lib)
            _handle = h
        self.h = _handle
        n, kk, cx = C.c_int64(), C.c_int(), C.c_int()
        self.lib.mxg_mv_info(self.h, C.byref(n), C.byref(kk), C.byref(cx))
        self.nlocal, self.k, self.is_complex = n.value, kk.value, bool(cx.value)

    # -- construction helpers
    @classmethod
    def from_numpy(cls, ctx, a):
        a = np.asarray(a)
        if a.ndim == 1:
            a = a[:, None]
        cx = np.iscomplexobj(a)
        mv = cls(ctx, a.shape[0], a.shape[1], cx)
        mv.set(a)
        return mv

    def set(self, a):
        a = np.asarray(a)
        if a.ndim == 1:
            a = a[:, None]
        dt = np.complex128 if self.is_complex else np.float64
        for j in range(self.k):
            col = np.ascontiguousarray(a[:, j], dtype=dt)
            _chk(self.lib.mxg_mv_put(self.h, j, 0, self.nlocal, _ptr(col)), self.lib)

    def to_numpy(self):
        dt = np.complex128 if self.is_complex else np.float64
        out = np.empty((self.nlocal, self.k), dtype=dt, order="F")
        for j in range(self.k):
            col = np.empty(self.nlocal, dtype=dt)
            _chk(self.lib.mxg_mv_fetch(self.h, j, 0, self.nlocal, _ptr(col)), self.lib)
            out[:, j] = col
        return out

    def view(self, cols):        # CloneView / CloneViewNonConst
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        h = C.c_void_p()
        _chk(self.lib.mxg_mv_view(self.h, cols.ctypes.data_as(C.POINTER(C.c_int)), len(cols), C.byref(h)),
             self.lib)
        return MultiVec(self.ctx, _handle=h, _parent=self)

    def clone(self, cols=None):  # CloneCopy
        h = C.c_void_p()
        if cols is None:
            _chk(self.lib.mxg_mv_clone(self.h, None, 0, C.byref(h)), self.lib)
        else:
            cols = np.ascontiguousarray(cols, dtype=np.int32)
            _chk(self.lib.mxg_mv_clone(self.h, cols.ctypes.data_as(C.POINTER(C.c_int)), len(cols),
                                       C.byref(h)), self.lib)
        return MultiVec(self.ctx, _handle=h)

    # -- MxMultiVector interface
    def copy_from(self, other):
        _chk(self.lib.mxg_mv_copy(self.h, other.h), self.lib)

    def fill(self, val):         # set / MvInit
        val = complex(val)
        _chk(self.lib.mxg_mv_fill(self.h, val.real, val.imag), self.lib)

    def random(self, seed=0):
        _chk(self.lib.mxg_mv_random(self.h, int(seed)), self.lib)

    def conj(self):
        _chk(self.lib.mxg_mv_conj(self.h), self.lib)

    def scale(self, a):
        if np.ndim(a) == 0:
            a = complex(a)
            _chk(self.lib.mxg_mv_scale(self.h, a.real, a.imag), self.lib)
        else:
            dt = np.complex128 if self.is_complex else np.float64
            al = np.ascontiguousarray(a, dtype=dt)
            assert al.shape == (self.k,)
            _chk(self.lib.mxg_mv_scale_cols(self.h, _ptr(al)), self.lib)

    def mask_rows(self, mask):
        _chk(self.lib.mxg_mv_mask_rows(self.h, mask.h), self.lib)

    def update(self, a, A, b):   # this = a*A + b*this
        a, b = complex(a), complex(b)
        _chk(self.lib.mxg_mv_axpby(a.real, a.imag, A.h, b.real, b.imag, self.h), self.lib)

    def add_mv(self, a, A, b, B):   # MvAddMv: this = a*A + b*B
        a, b = complex(a), complex(b)
        _chk(self.lib.mxg_mv_add(a.real, a.imag, A.h, b.real, b.imag, B.h, self.h), self.lib)

    def dot(self, A):            # res[j] = A_j^H this_j  (MvDot / dot(mv, res))
        w = 2 if self.is_complex else 1
        res = np.empty(self.k * w)
        _chk(self.lib.mxg_mv_dot(A.h, self.h, _ptr(res)), self.lib)
        return res.view(np.complex128) if self.is_complex else res

    def norm2(self):
        res = np.empty(self.k)
        _chk(self.lib.mxg_mv_norm2(self.h, _ptr(res)), self.lib)
        return res

    def set_block(self, A, index):
        index = np.ascontiguousarray(index, dtype=np.int32)
        _chk(self.lib.mxg_mv_set_block(self.h, A.h, index.ctypes.data_as(C.POINTER(C.c_int)), len(index)),
             self.lib)

    def times_mat_add_mv(self, alpha, A, B, beta):   # this = alpha*A*B + beta*this
        dt = np.complex128 if self.is_complex else np.float64
        B = np.asfortranarray(B, dtype=dt)
        m, k = B.shape
        alpha, beta = complex(alpha), complex(beta)
        _chk(self.lib.mxg_mv_times_mat(alpha.real, alpha.imag, A.h, _ptr(B), m, m, k,
                                       beta.real, beta.imag, self.h), self.lib)

    def trans_mv(self, alpha, A, A2=None):   # B = alpha * [A A2]^H * this (one pass over the rows)
        dt = np.complex128 if self.is_complex else np.float64
        m = A.k + (A2.k if A2 is not None else 0)
        B = np.empty((m, self.k), dtype=dt, order="F")
        alpha = complex(alpha)
        _chk(self.lib.mxg_mv_trans_mv2(alpha.real, alpha.imag, A.h, A2.h if A2 is not None else None, self.h, _ptr(B), m),
             self.lib)
        return B

    def close(self):
        if self.h:
            self.lib.mxg_mv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CsrMatrix:
    """MxCrsMatrix on the device; apply() = SpMV / SpMM."""

    def __init__(self, ctx, nrows, ncols, rowptr, colidx, vals, is_complex=False, nghost=0, plan=None,
                 _keep=None):
        self.ctx, self.lib = ctx, ctx.lib
        h = C.c_void_p()
        _chk(self.lib.mxg_csr_create(ctx.h, int(nrows), int(ncols), int(nghost), rowptr, colidx, vals,
                                     int(bool(is_complex)), plan, C.byref(h)), self.lib)
        self.h = h
        self.nrows, self.ncols, self.is_complex = int(nrows), int(ncols), bool(is_complex)
        n = C.c_int64()
        self.lib.mxg_csr_info(self.h, None, None, None, C.byref(n), None)
        self.nnz = n.value

    @classmethod
    def from_scipy(cls, ctx, m, nghost=0, plan=None):
        m = m.tocsr()
        cx = np.iscomplexobj(m.data)
        rowptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
        col = np.ascontiguousarray(m.indices, dtype=np.int32)
        val = np.ascontiguousarray(m.data, dtype=np.complex128 if cx else np.float64)
        return cls(ctx, m.shape[0], m.shape[1] - nghost, _ptr(rowptr), _ptr(col), _ptr(val), cx,
                   nghost=nghost, plan=plan)

    def apply(self, x, y):
        _chk(self.lib.mxg_csr_apply(self.ctx.h, self.h, x.h, y.h), self.lib)

    # ---- setup-time sparse algebra on the device (mxg_spalg.hip); results are new device matrices
    @classmethod
    def _wrap(cls, ctx, h):
        m = cls.__new__(cls)
        m.ctx, m.lib, m.h = ctx, ctx.lib, h
        nr, nc, ng, nz, cx = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        _chk(m.lib.mxg_csr_info(h, C.byref(nr), C.byref(nc), C.byref(ng), C.byref(nz), C.byref(cx)), m.lib)
        m.nrows, m.ncols, m.nnz, m.is_complex = nr.value, nc.value + ng.value, nz.value, bool(cx.value)
        return m

    def multiply(self, other):       # MxCrsMatrix::matrixMatrixMultiply(this, other)
        h = C.c_void_p()
        _chk(self.lib.mxg_csr_multiply(self.ctx.h, self.h, other.h, C.byref(h)), self.lib)
        return CsrMatrix._wrap(self.ctx, h)

    def add(self, sa, other, sb):    # MxCrsMatrix::matrixMatrixAdd: sa * this + sb * other
        h = C.c_void_p()
        za = (C.c_double * 2)(complex(sa).real, complex(sa).imag)
        zb = (C.c_double * 2)(complex(sb).real, complex(sb).imag)
        _chk(self.lib.mxg_csr_add_z(self.ctx.h, self.h, za, other.h, zb, C.byref(h)), self.lib)
        return CsrMatrix._wrap(self.ctx, h)

    def purge(self, tol=1e-12):      # MxCrsMatrix::purgeZeros
        h = C.c_void_p()
        _chk(self.lib.mxg_csr_purge(self.ctx.h, self.h, float(tol), C.byref(h)), self.lib)
        return CsrMatrix._wrap(self.ctx, h)

    def to_scipy(self):
        import scipy.sparse as sp
        rp = np.empty(self.nrows + 1, dtype=np.int64)
        ci = np.empty(max(self.nnz, 1), dtype=np.int32)
        va = np.empty(max(self.nnz, 1) * (2 if self.is_complex else 1))
        _chk(self.lib.mxg_csr_download(self.h, _ptr(rp), _ptr(ci), _ptr(va)), self.lib)
        if self.is_complex:
            va = va.view(np.complex128)
        return sp.csr_matrix((va[:self.nnz], ci[:self.nnz], rp), shape=(self.nrows, self.ncols))

    def apply_bytes(self, k=1):
        return self.lib.mxg_csr_apply_bytes(self.h, k)

    def set_kernel(self, which):
        """0 default CSR kernels, 1 row-gather kernel on the int64 row pointers, 2 complex stream kernel (mxg.h)."""
        _chk(self.lib.mxg_csr_set_kernel(self.h, which), self.lib)

    def kernel_name(self):
        return (self.lib.mxg_csr_kernel_name(self.h) or b"").decode()

    def compress(self):
        """Re-encode for the solver's product kernels (tile form or compressed stream, mxg.h); returns (work units,
        fallback blocks, bytes of the re-encoded form)."""
        nb, nf, sb = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(self.lib.mxg_csr_compress(self.h, C.byref(nb), C.byref(nf), C.byref(sb)), self.lib)
        return nb.value, nf.value, sb.value

    def product_form(self):
        """(form, tiles, tile patterns, irregular rows, coverage): form 0 plain CSR, 1 compressed stream, 3 tile patterns."""
        t, p, g, cov = C.c_int64(), C.c_int64(), C.c_int64(), C.c_double()
        form = self.lib.mxg_csr_product_form(self.h, C.byref(t), C.byref(p), C.byref(g), C.byref(cov))
        return form, t.value, p.value, g.value, cov.value

    def solve(self, b, x, method="cg", tol=1e-6, restart=40, max_iters=0, precond="none", mg=None):
        opts = _lib.SolveOpts(tol, restart, max_iters, {"none": 0, "jacobi": 1, "mg": 3}[precond], 0, 0.0, 0.0,
                              mg.handle if mg is not None else None)
        info = _lib.SolveInfo()
        fn = {"cg": self.lib.mxg_solve_cg, "gmres": self.lib.mxg_solve_gmres, "bicgstab": self.lib.mxg_solve_bicgstab}[method]
        _chk(fn(self.ctx.h, self.h, b.h, x.h, C.byref(opts), C.byref(info)), self.lib)
        return info.iters, bool(info.converged), info.true_res

    def close(self):
        if self.h:
            self.lib.mxg_csr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


PILLBOX_SHAPE = "cylinder(0.4;0,0,1;0,0,0)&slab(0.8;0,0,1;0,0,0)"


class Sim:
    """Grid + BCs + PEC shape -> assembled operators (MxEMSim::setup + MxEMOps + initMatrices)."""

    def __init__(self, n, origin=(0.0, 0.0, 0.0), size=(1.0, 1.0, 1.0), lower=("periodic",) * 3,
                 upper=("periodic",) * 3, pec_shape=None, strict=True, dielectrics=None, dm_frac=0.0, box=None, mus=None):
        """dielectrics: list of (eps, shape_spec[, loss_tangent]); eps = 3 diagonal numbers, or 6 (diag + the
        deck's `eps off diag` triple), or a scalar for an isotropic material.  A loss tangent makes the problem
        complex: its operators come from Problem.op, not from Sim.
        box: (lo, hi) cell box of this rank on a decomposed grid (maxwell_amd.distributed.BoxSim)."""
        self.lib = _lib.load_mxhost()
        n = [int(n)] * 3 if np.ndim(n) == 0 else [int(v) for v in n]
        self.n = n
        nn = (C.c_int * 3)(*n)
        oo = (C.c_double * 3)(*origin)
        ss = (C.c_double * 3)(*size)
        lo = (C.c_char_p * 3)(*[s.encode() for s in lower])
        up = (C.c_char_p * 3)(*[s.encode() for s in upper])
        h = C.c_void_p()
        blo = (C.c_int * 3)(*box[0]) if box is not None else None
        bhi = (C.c_int * 3)(*box[1]) if box is not None else None
        dspec = None
        if dielectrics:
            items = []
            for item in dielectrics:
                eps, shape = item[0], item[1]
                tan = float(item[2]) if len(item) > 2 else 0.0
                e = [float(eps)] * 3 if np.ndim(eps) == 0 else [float(v) for v in eps]
                items.append(",".join(repr(v) for v in e) + (" tan=%r" % tan if tan else "") + " : " + shape)
            dspec = " | ".join(items).encode()
        if mus:                    # magnetic materials: list of (mu, shape_spec), mu = scalar, 3 diagonal numbers, or 6
            items = []
            for mu, shape in mus:
                m = [float(mu)] * 3 if np.ndim(mu) == 0 else [float(v) for v in mu]
                items.append("mu " + ",".join(repr(v) for v in m) + " : " + shape)
            dspec = ((dspec.decode() + " | ") if dspec else "").encode() + " | ".join(items).encode()
        _chk(self.lib.mxh_sim_create_ex(nn, oo, ss, lo, up, pec_shape.encode() if pec_shape else None, dspec,
                                        float(dm_frac), int(bool(strict)), blo, bhi, C.byref(h)), self.lib, "mxh")
        self.h = h

    @classmethod
    def pillbox(cls, n, **kw):
        """example/pillbox.py geometry at resolution n."""
        return cls(n, origin=(-0.5,) * 3, size=(1.0,) * 3, pec_shape=PILLBOX_SHAPE, **kw)

    def op_raw(self, name, shift=None):
        """(nrows, ncols, nnz, rowptr*, colidx*, vals*) borrowed from the C++ side (no copies)."""
        nr, nc, nz = C.c_int64(), C.c_int64(), C.c_int64()
        rp, ci, va = _lib.c_i64p(), _lib.c_i32p(), _lib.c_dp()
        if shift is None:
            _chk(self.lib.mxh_sim_op(self.h, name.encode(), C.byref(nr), C.byref(nc), C.byref(nz),
                                     C.byref(rp), C.byref(ci), C.byref(va)), self.lib, "mxh")
        else:
            _chk(self.lib.mxh_sim_shifted(self.h, float(shift), C.byref(nr), C.byref(nz), C.byref(rp),
                                          C.byref(ci), C.byref(va)), self.lib, "mxh")
            nc = nr
        return nr.value, nc.value, nz.value, rp, ci, va

    def op(self, name, shift=None):
        """scipy.sparse copy of a named operator (tests)."""
        import scipy.sparse as sp
        nr, nc, nz, rp, ci, va = self.op_raw(name, shift)
        rowptr = np.ctypeslib.as_array(rp, (nr + 1,)).copy()
        col = np.ctypeslib.as_array(ci, (max(nz, 1),))[:nz].copy()
        val = np.ctypeslib.as_array(va, (max(nz, 1),))[:nz].copy()
        return sp.csr_matrix((val, col, rowptr), shape=(nr, nc))

    def form_chain_on_device(self, ctx):
        """Form the operator chain (curlCurl, gradDiv, vecLapl, scaLapl, mRhs) on the GPU of `ctx` and read it back
        (mxh_sim_device_chain); what mxh_solver_create does for the window of a rank of a decomposed grid."""
        _chk(self.lib.mxh_sim_device_chain(ctx.h, self.h), self.lib, "mxh")

    def to_device(self, ctx, name, shift=None):
        nr, nc, nz, rp, ci, va = self.op_raw(name, shift)
        return CsrMatrix(ctx, nr, nc, C.cast(rp, C.c_void_p), C.cast(ci, C.c_void_p),
                         C.cast(va, C.c_void_p), False)

    def map(self, field):
        n, g = C.c_int64(), _lib.c_i64p()
        _chk(self.lib.mxh_sim_map(self.h, field.encode(), C.byref(n), C.byref(g)), self.lib, "mxh")
        return np.ctypeslib.as_array(g, (max(n.value, 1),))[:n.value].copy()

    def used_mask(self):
        n, m = C.c_int64(), _lib.c_dp()
        _chk(self.lib.mxh_sim_used_mask(self.h, C.byref(n), C.byref(m)), self.lib, "mxh")
        return np.ctypeslib.as_array(m, (max(n.value, 1),))[:n.value].copy()

    @property
    def setup_seconds(self):
        return self.lib.mxh_sim_setup_seconds(self.h)

    def close(self):
        if self.h:
            self.lib.mxh_sim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Multigrid:
    """Aggregation multigrid for the shifted vector Laplacian ("shifted") or the scalar Laplacian
    ("scaLapl") of a Sim: hierarchy built on the host, V-cycle on the GPU."""

    def __init__(self, ctx, sim, op="shifted", shift=0.0, min_coarse=2000, max_levels=10, sweeps=2,
                 coarse_sweeps=30, omega=0.85, coarse_scale=1.9, smoother="jacobi", cheb_ratio=7.0):
        """smoother: "jacobi" (damping omega) or "chebyshev" (ML's default; `sweeps` Chebyshev steps for D^-1 A on
        [lmax / cheb_ratio, lmax], lmax per level by power iteration)."""
        self.lib = _lib.load_mxhost()
        self.ctx, self.sim = ctx, sim
        h = C.c_void_p()
        _chk(self.lib.mxh_mg_create(ctx.h, sim.h, op.encode(), float(shift), int(min_coarse), int(max_levels),
                                    int(sweeps), int(coarse_sweeps), float(omega), float(coarse_scale),
                                    C.byref(h)), self.lib, "mxh")
        self.h = h
        self.handle = C.c_void_p(self.lib.mxh_mg_handle(h))
        if smoother.lower() == "chebyshev":
            _chk(ctx.lib.mxg_mg_set_chebyshev(self.handle, float(cheb_ratio)), ctx.lib)
        rows, nnz, nl = (C.c_int64 * 16)(), (C.c_int64 * 16)(), C.c_int()
        self.lib.mxh_mg_level_sizes(h, rows, nnz, C.byref(nl))
        self.levels = [(rows[i], nnz[i]) for i in range(nl.value)]

    def fine_matrix(self):
        """Borrowed CsrMatrix wrapper of the level-0 operator (owned by this object)."""
        m = CsrMatrix.__new__(CsrMatrix)
        m.ctx, m.lib = self.ctx, self.ctx.lib
        m.h = C.c_void_p(self.lib.mxh_mg_fine_matrix(self.h))
        m.nrows = m.ncols = self.levels[0][0]
        m.nnz, m.is_complex = self.levels[0][1], False
        m.close = lambda: None
        m._owner = self
        return m

    def level_matrix(self, level, which="A"):
        """Borrowed wrapper of a matrix of the hierarchy: 'A' level operator, 'P' prolongation level + 1 -> level,
        'R' restriction level -> level + 1 (tests)."""
        h = self.lib.mxh_mg_level_matrix(self.h, int(level), which.encode())
        if not h:
            raise MxError("Multigrid.level_matrix: no matrix %s at level %d" % (which, level))
        m = CsrMatrix._wrap(self.ctx, C.c_void_p(h))
        m.close = lambda: None
        m._owner = self
        return m

    def apply(self, r, z):
        _chk(self.ctx.lib.mxg_mg_apply(self.ctx.h, self.handle, r.h, z.h), self.ctx.lib)

    def close(self):
        if self.h:
            self.lib.mxh_mg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SolverParams:
    """The `eigensolver : *` / `linear solver : *` parameters MxSolver reads
    (src/MxSolver.cpp:27-38, src/MxProblem.cpp:26-67)."""

    def __init__(self, **kw):
        lib = _lib.load_mxhost()
        self.c = _lib.SolverParamsC()
        lib.mxh_solver_default_params(C.byref(self.c))
        self._keep = []
        for k, v in kw.items():
            self.set(k, v)

    def set(self, k, v):
        if k in ("lin_type", "prec_type", "amg_smoother", "eig_type"):
            b = v.encode()
            self._keep.append(b)
            setattr(self.c, k, b)
        else:
            setattr(self.c, k, v)


class Solver:
    """MxSolver: block Krylov-Schur in shift-invert mode around MxMagWaveOp::Apply."""

    def __init__(self, ctx, sim, params=None, _handle=None):
        self.lib = _lib.load_mxhost()
        self.ctx, self.sim = ctx, sim
        if _handle is not None:          # created by Problem.solver() from a deck
            self.params, self.h = None, _handle
            return
        self.params = params or SolverParams()
        h = C.c_void_p()
        _chk(self.lib.mxh_solver_create(ctx.h, sim.h, C.byref(self.params.c), C.byref(h)), self.lib, "mxh")
        self.h = h

    def solve(self):
        n = C.c_int()
        _chk(self.lib.mxh_solver_solve(self.h, C.byref(n)), self.lib, "mxh")
        self.nconv = n.value
        return self.nconv

    def result(self, i):
        th, k2, fr = (C.c_double * 2)(), (C.c_double * 2)(), (C.c_double * 2)()
        res, dv = C.c_double(), C.c_double()
        _chk(self.lib.mxh_solver_result(self.h, i, th, k2, fr, C.byref(res), C.byref(dv)), self.lib, "mxh")
        return dict(theta=complex(th[0], th[1]), k2=complex(k2[0], k2[1]), freq=complex(fr[0], fr[1]),
                    residual=res.value, divergence=dv.value)

    def stats(self):
        a, v, s = C.c_int(), C.c_int(), C.c_int()
        tv, ts, te = C.c_double(), C.c_double(), C.c_double()
        _chk(self.lib.mxh_solver_stats(self.h, C.byref(a), C.byref(v), C.byref(s), C.byref(tv),
                                       C.byref(ts), C.byref(te)), self.lib, "mxh")
        return dict(applies=a.value, vec_iters=v.value, sca_iters=s.value, vec_seconds=tv.value,
                    sca_seconds=ts.value, eigensolve_seconds=te.value)

    def eigvec(self, i, n):
        out = np.empty(n)
        _chk(self.lib.mxh_solver_eigvec(self.h, i, _ptr(out)), self.lib, "mxh")
        return out

    def close(self):
        if self.h:
            self.lib.mxh_solver_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Problem:
    """A `.mx` input deck (MxProblem, src/MxProblem.cpp:201-706): parsed on the host, assembled on
    demand (`sim()`), solved on the GPU with the deck's own parameters (`solver(ctx)`)."""

    BC_NAMES = ("periodic", "pec", "pmc")

    def __init__(self, text=None, path=None, strict=True):
        self.lib = _lib.load_mxhost()
        h = C.c_void_p()
        if (text is None) == (path is None):
            raise ValueError("Problem: give exactly one of text= or path=")
        if text is not None:
            _chk(self.lib.mxh_problem_parse(text.encode(), int(bool(strict)), C.byref(h)), self.lib, "mxh")
        else:
            _chk(self.lib.mxh_problem_load(str(path).encode(), int(bool(strict)), C.byref(h)), self.lib, "mxh")
        self.h = h
        self._sim = None

    def param(self, key, default=None):
        buf = C.create_string_buffer(256)
        if self.lib.mxh_problem_param(self.h, key.encode(), buf, len(buf)) != 0:
            if default is None:
                raise KeyError(key)
            return default
        return buf.value.decode()

    def grid(self):
        n, lo, up = (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)()
        o, s = (C.c_double * 3)(), (C.c_double * 3)()
        dm, has = C.c_double(), C.c_int()
        _chk(self.lib.mxh_problem_grid(self.h, n, o, s, lo, up, C.byref(dm), C.byref(has)), self.lib, "mxh")
        return dict(n=list(n), origin=list(o), size=list(s), lower=[self.BC_NAMES[i] for i in lo],
                    upper=[self.BC_NAMES[i] for i in up], dm_frac=dm.value, has_pec=bool(has.value))

    def pec_func(self, xyz):
        p = (C.c_double * 3)(*xyz)
        f = C.c_double()
        _chk(self.lib.mxh_problem_pec_func(self.h, p, C.byref(f)), self.lib, "mxh")
        return f.value

    @property
    def is_complex(self):
        return bool(self.lib.mxh_problem_is_complex(self.h))

    def op(self, name):
        """scipy.sparse copy of a named operator; complex for decks with phase shifts."""
        import scipy.sparse as sp
        nr, nc, nz, cx = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        rp, ci, va = _lib.c_i64p(), _lib.c_i32p(), _lib.c_dp()
        _chk(self.lib.mxh_problem_op(self.h, name.encode(), C.byref(nr), C.byref(nc), C.byref(nz), C.byref(rp),
                                     C.byref(ci), C.byref(va), C.byref(cx)), self.lib, "mxh")
        n = nz.value
        rowptr = np.ctypeslib.as_array(rp, (nr.value + 1,)).copy()
        col = np.ctypeslib.as_array(ci, (max(n, 1),))[:n].copy()
        w = 2 if cx.value else 1
        val = np.ctypeslib.as_array(va, (max(n * w, 1),))[:n * w].copy()
        if cx.value:
            val = val.view(np.complex128)
        return sp.csr_matrix((val, col, rowptr), shape=(nr.value, nc.value))

    def sim(self):
        """The assembled simulation (owned by the problem; a Sim view that does not free it)."""
        if self._sim is None:
            h = C.c_void_p()
            _chk(self.lib.mxh_problem_sim(self.h, C.byref(h)), self.lib, "mxh")
            view = Sim.__new__(Sim)
            view.lib, view.h, view.n = self.lib, h, self.grid()["n"]
            view.close = lambda: None
            view._owner = self
            self._sim = view
        return self._sim

    def solver(self, ctx, verbose=0):
        h = C.c_void_p()
        _chk(self.lib.mxh_problem_solver(ctx.h, self.h, int(verbose), C.byref(h)), self.lib, "mxh")
        s = Solver(ctx, None if self.is_complex else self.sim(), _handle=h)
        s._owner = self
        return s

    def close(self):
        if self.h:
            self.lib.mxh_problem_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
