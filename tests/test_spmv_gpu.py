"""K1: CSR SpMV / SpMM through the C ABI against the serial row loop of the oracle -- bit exact
(products are rounded, then added in storage order on both sides)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import maxwell_amd as mx
from oracle import cref, yee

pytestmark = pytest.mark.gpu


def _apply(ctx, m, x):
    A = mx.CsrMatrix.from_scipy(ctx, m)
    X = mx.MultiVec.from_numpy(ctx, x)
    Y = mx.MultiVec(ctx, m.shape[0], X.k, X.is_complex)
    A.apply(X, Y)
    return Y.to_numpy()


@pytest.mark.parametrize("name", ["vecLapl", "scaLapl", "curlCurl", "gradDiv", "mRhs", "divB",
                                  "gradPsi", "curlE", "curlB"])
def test_pillbox_operators_bit_exact(ctx, oracle_pillbox16, rng, name):
    """Every operator MxMagWaveOp applies (a2-a5 of SURVEY 8a), square and rectangular."""
    _, o = oracle_pillbox16
    m = o[name].tocsr()
    x = rng.uniform(-1, 1, m.shape[1])
    got = _apply(ctx, m, x)[:, 0]
    assert np.array_equal(got, cref.spmv(m, x))


def test_shifted_operator_from_product_assembler(ctx, oracle_pillbox16, rng):
    """The matrix the hot loop really uses: vecLapl - shift*dmA built by the C++ assembler."""
    _, o = oracle_pillbox16
    shift = yee.default_shift([1, 1, 1])
    sim = mx.Sim.pillbox(16)
    host = sim.op("vecLapl", shift=shift)
    ref = yee.shifted(o["vecLapl"], o["mRhs"], shift)
    x = rng.uniform(-1, 1, host.shape[1])
    A = sim.to_device(ctx, "vecLapl", shift=shift)
    X = mx.MultiVec.from_numpy(ctx, x)
    Y = mx.MultiVec(ctx, host.shape[0], 1)
    A.apply(X, Y)
    assert np.array_equal(Y.to_numpy()[:, 0], cref.spmv(ref, x))


@pytest.mark.parametrize("k", [2, 3, 8])
def test_spmm_block_of_vectors(ctx, oracle_pillbox12, rng, k):
    _, o = oracle_pillbox12
    m = o["curlCurl"]
    x = rng.uniform(-1, 1, (m.shape[1], k))
    assert np.array_equal(_apply(ctx, m, x), cref.spmv(m, x))


@pytest.mark.parametrize("code", [0, 1])
def test_spmm_views_every_row_gather_kernel(ctx, oracle_pillbox16, rng, code):
    """Blocks of vectors and views of scattered columns through the row-gather kernel with 16-bit in-block row offsets
    (0, the default) and with the int64 row pointers of the CSR arrays (1)."""
    _, o = oracle_pillbox16
    m = o["curlCurl"]
    x = rng.uniform(-1, 1, (m.shape[1], 6))
    A = mx.CsrMatrix.from_scipy(ctx, m)
    A.set_kernel(code)
    X = mx.MultiVec.from_numpy(ctx, x)
    Y = mx.MultiVec(ctx, m.shape[0], 6)
    A.apply(X, Y)
    assert np.array_equal(Y.to_numpy(), cref.spmv(m, x))
    Y.fill(0.0)
    A.apply(X.view([4, 5, 0, 2]), Y.view([1, 2, 5, 0]))
    got = Y.to_numpy()
    assert np.array_equal(got[:, [1, 2, 5, 0]], cref.spmv(m, x[:, [4, 5, 0, 2]])) and np.all(got[:, [3, 4]] == 0)
    A.apply(X, X)                                        # aliasing: works from a copy of x
    assert np.array_equal(X.to_numpy(), cref.spmv(m, x))


def test_apply_in_place_aliasing(ctx, oracle_pillbox12, rng):
    """apply(x, x) is used by the reference (src/MxMagWaveOp.cpp:650,1249)."""
    _, o = oracle_pillbox12
    m = o["vecLapl"]
    x = rng.uniform(-1, 1, (m.shape[1], 2))
    A = mx.CsrMatrix.from_scipy(ctx, m)
    X = mx.MultiVec.from_numpy(ctx, x)
    A.apply(X, X)
    assert np.array_equal(X.to_numpy(), cref.spmv(m, x))
    # partial overlap through views of one allocation
    Z = mx.MultiVec.from_numpy(ctx, np.concatenate([x, x], axis=1))
    A.apply(Z.view([0, 1]), Z.view([1, 2]))
    want = cref.spmv(m, x)
    got = Z.to_numpy()
    assert np.array_equal(got[:, 1:3], want) and np.array_equal(got[:, 0], x[:, 0])


def test_views_of_noncontiguous_columns(ctx, oracle_pillbox12, rng):
    _, o = oracle_pillbox12
    m = o["vecLapl"]
    x = rng.uniform(-1, 1, (m.shape[1], 5))
    X = mx.MultiVec.from_numpy(ctx, x)
    Y = mx.MultiVec(ctx, m.shape[0], 5)
    A = mx.CsrMatrix.from_scipy(ctx, m)
    A.apply(X.view([4, 0, 2]), Y.view([1, 3, 0]))
    got = Y.to_numpy()
    want = cref.spmv(m, x[:, [4, 0, 2]])
    assert np.array_equal(got[:, [1, 3, 0]], want) and np.all(got[:, [2, 4]] == 0)


@pytest.mark.parametrize("code", [None, 2])
def test_complex_matrix(ctx, rng, code):
    """Complex scalars are interleaved pairs (the reference's 2x2 real embedding,
    src/MxCrsMatrix.cpp:146-170, collapses to complex multiply-add).  Default: the persistent row-gather kernel
    over 16-byte values; code 2: the one-block-per-workgroup stream kernel.  Includes an over-long row, empty rows
    and views of scattered columns."""
    n = 3000
    m = (sp.random(n, n, density=4.0 / n, random_state=7, format="csr")
         + 1j * sp.random(n, n, density=4.0 / n, random_state=8, format="csr")).tolil()
    m[5, :1500] = rng.uniform(-1, 1, 1500) + 1j * rng.uniform(-1, 1, 1500)      # longer than a block
    for r in range(700, 1100):
        m[r, :] = 0
    m = m.tocsr(); m.eliminate_zeros(); m.sort_indices()
    x = rng.uniform(-1, 1, (n, 3)) + 1j * rng.uniform(-1, 1, (n, 3))
    A = mx.CsrMatrix.from_scipy(ctx, m)
    if code is not None:
        A.set_kernel(code)
    X = mx.MultiVec.from_numpy(ctx, x)
    Y = mx.MultiVec(ctx, n, 3, True)
    A.apply(X, Y)
    got, want = Y.to_numpy(), cref.spmv(m, x)
    mask = np.ones(n, bool); mask[5] = False
    assert np.array_equal(got[mask], want[mask])
    assert np.abs(got[5] - want[5]).max() <= 1e-12 * np.abs(m[5].data).sum()
    Y.fill(0.0)
    A.apply(X.view([2, 0]), Y.view([0, 2]))
    got = Y.to_numpy()
    assert np.array_equal(got[mask][:, [0, 2]], want[mask][:, [2, 0]]) and np.all(got[:, 1] == 0)


def test_ragged_rows_empty_rows_and_long_row(ctx, rng):
    """Edge cases: empty rows, rows longer than one row block (falls back to the whole-workgroup
    path), more empty rows than a block can index, odd nnz offsets."""
    n = 5000
    rows = [rng.choice(n, size=int(c), replace=False) for c in rng.integers(0, 30, size=n)]
    rows[17] = np.arange(n)[: 4500]            # > NNZ_BLK entries in one row
    for r in range(100, 2400):                 # > ROWS_BLK consecutive empty rows
        rows[r] = np.array([], dtype=int)
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])])
    indices = np.concatenate([np.sort(r) for r in rows])
    data = rng.uniform(-1, 1, len(indices))
    m = sp.csr_matrix((data, indices, indptr), shape=(n, n))
    x = rng.uniform(-1, 1, n)
    got = _apply(ctx, m, x)[:, 0]
    want = cref.spmv(m, x)
    long_row = 17
    mask = np.ones(n, bool); mask[long_row] = False
    assert np.array_equal(got[mask], want[mask])
    assert abs(got[long_row] - want[long_row]) <= 1e-12 * np.abs(data[indptr[17]:indptr[18]]).sum()


def test_empty_and_tiny(ctx):
    m = sp.csr_matrix((3, 3))
    assert np.array_equal(_apply(ctx, m, np.ones(3))[:, 0], np.zeros(3))
    m = sp.csr_matrix(np.array([[2.0]]))
    assert _apply(ctx, m, np.array([3.0]))[0, 0] == 6.0


def test_error_behaviour(ctx, oracle_pillbox12):
    _, o = oracle_pillbox12
    A = mx.CsrMatrix.from_scipy(ctx, o["divB"])
    X = mx.MultiVec(ctx, o["divB"].shape[0], 1)      # wrong length on purpose
    Y = mx.MultiVec(ctx, o["divB"].shape[0], 1)
    with pytest.raises(mx.MxError, match="rows"):
        A.apply(X, Y)
    bad = o["divB"].copy().tocsr()
    bad.indices = bad.indices.copy(); bad.indices[0] = bad.shape[1] + 5
    with pytest.raises(mx.MxError, match="out of range"):
        mx.CsrMatrix.from_scipy(ctx, bad)


def test_full_size_linearity_and_constant_nullspace(ctx):
    """Size-independent properties at a larger size (64^3 pillbox, ~3.4e5 rows):
    linearity A(ax+by) = aAx + bAy to rounding, and divB*curlE = 0 applied as two SpMVs."""
    sim = mx.Sim.pillbox(64)
    rng = np.random.default_rng(1)
    A = sim.to_device(ctx, "vecLapl", shift=1.97)
    n = A.nrows
    x = rng.uniform(-1, 1, n); y = rng.uniform(-1, 1, n)
    X = mx.MultiVec.from_numpy(ctx, np.stack([x, y, 2.0 * x - 3.0 * y], axis=1))
    Y = mx.MultiVec(ctx, n, 3)
    A.apply(X, Y)
    out = Y.to_numpy()
    scale = np.abs(out).max()
    assert np.abs(out[:, 2] - (2.0 * out[:, 0] - 3.0 * out[:, 1])).max() < 1e-12 * scale
    curlE = sim.to_device(ctx, "curlE"); divB = sim.to_device(ctx, "divB")
    e = mx.MultiVec.from_numpy(ctx, rng.uniform(-1, 1, curlE.ncols))
    b = mx.MultiVec(ctx, curlE.nrows, 1); p = mx.MultiVec(ctx, divB.nrows, 1)
    curlE.apply(e, b); divB.apply(b, p)
    assert p.norm2()[0] < 1e-9 * b.norm2()[0]


@pytest.mark.parametrize("code", [0, 1])
def test_every_kernel_variant_bit_exact(ctx, oracle_pillbox16, rng, code):
    """The shipped CSR kernels of a real matrix (16-bit row offsets / int64 row pointers) give the bits of the serial
    row loop; an unknown code is an error, not a silent default."""
    _, o = oracle_pillbox16
    m = yee.shifted(o["vecLapl"], o["mRhs"], 1.9739208802178716)
    x = rng.uniform(-1, 1, m.shape[1])
    A = mx.CsrMatrix.from_scipy(ctx, m)
    A.set_kernel(code)
    X = mx.MultiVec.from_numpy(ctx, x)
    Y = mx.MultiVec(ctx, m.shape[0], 1)
    A.apply(X, Y)
    assert np.array_equal(Y.to_numpy()[:, 0], cref.spmv(m, x))
    with pytest.raises(mx.MxError, match="unknown kernel code"):
        A.set_kernel(0x750000)


def test_row_gather_kernel_ragged(ctx, rng):
    """Row-gather kernels on ragged rows incl. empty stretches and an over-long row."""
    n = 6000
    rows = [rng.choice(n, size=int(c), replace=False) for c in rng.integers(0, 25, size=n)]
    rows[3] = np.arange(n)[:3000]
    for r in range(500, 1700):
        rows[r] = np.array([], dtype=int)
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])])
    indices = np.concatenate([np.sort(r) for r in rows])
    m = sp.csr_matrix((rng.uniform(-1, 1, len(indices)), indices, indptr), shape=(n, n))
    x = rng.uniform(-1, 1, n)
    want = cref.spmv(m, x)
    for code in (0, 1):
        A = mx.CsrMatrix.from_scipy(ctx, m)
        A.set_kernel(code)
        X = mx.MultiVec.from_numpy(ctx, x); Y = mx.MultiVec(ctx, n, 1)
        A.apply(X, Y)
        got = Y.to_numpy()[:, 0]
        mask = np.ones(n, bool); mask[3] = False
        assert np.array_equal(got[mask], want[mask])
        assert abs(got[3] - want[3]) <= 1e-12 * np.abs(m[3].data).sum()


def test_halo_exchange_on_a_one_rank_communicator(built, oracle_pillbox12, rng):
    """C1 on one GPU: a 1-rank RCCL communicator sends to itself.  Part of the columns is re-routed
    through ghost slots fed by the halo plan (pack kernel -> ncclSend/ncclRecv -> [owned | ghosts]);
    the product must equal the plain one bit for bit."""
    import ctypes as C
    from maxwell_amd import _lib
    _, o = oracle_pillbox12
    m = yee.shifted(o["vecLapl"], o["mRhs"], 1.97).tocsr()
    n = m.shape[0]
    ghosted = np.nonzero(np.arange(n) % 5 == 0)[0]                # owned entries that also travel as ghosts
    slot = -np.ones(n, dtype=np.int64); slot[ghosted] = n + np.arange(len(ghosted))
    cols = m.indices.astype(np.int64).copy()
    reroute = (slot[cols] >= 0) & (rng.uniform(0, 1, len(cols)) < 0.7)
    cols[reroute] = slot[cols[reroute]]
    rowptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
    colidx = np.ascontiguousarray(cols, dtype=np.int32)
    vals = np.ascontiguousarray(m.data)
    neigh = np.array([0], dtype=np.int32)
    off = np.array([0, len(ghosted)], dtype=np.int64)
    send_idx = np.ascontiguousarray(ghosted, dtype=np.int64)
    plan = _lib.HaloPlan(1, neigh.ctypes.data_as(C.POINTER(C.c_int)), off.ctypes.data_as(_lib.c_i64p),
                         send_idx.ctypes.data_as(_lib.c_i64p), off.ctypes.data_as(_lib.c_i64p))
    ctx1 = mx.Ctx(0, 0, 1, mx.Ctx.unique_id())
    try:
        A = mx.CsrMatrix(ctx1, n, n, rowptr.ctypes.data_as(C.c_void_p), colidx.ctypes.data_as(C.c_void_p),
                         vals.ctypes.data_as(C.c_void_p), False, nghost=len(ghosted), plan=C.byref(plan))
        x = rng.uniform(-1, 1, (n, 2))
        X = mx.MultiVec.from_numpy(ctx1, x); Y = mx.MultiVec(ctx1, n, 2)
        A.apply(X, Y)
        assert np.array_equal(Y.to_numpy(), cref.spmv(m, x))
        # default kernel: interior blocks run while the ghosts travel on the communication stream, the blocks
        # that read ghosts follow; repeated products re-use the ghost buffer
        for rep in range(3):
            x2 = rng.uniform(-1, 1, (n, 1))
            X2 = mx.MultiVec.from_numpy(ctx1, x2); Y2 = mx.MultiVec(ctx1, n, 1)
            A.apply(X2, Y2)
            assert np.array_equal(Y2.to_numpy(), cref.spmv(m, x2))
        # the int64-row-pointer variant of the kernel, then back; x aliasing y takes the [owned | ghosts] copy
        A.set_kernel(1)
        A.apply(X, Y)
        assert np.array_equal(Y.to_numpy(), cref.spmv(m, x))
        A.set_kernel(0)
        A.apply(X, Y)
        assert np.array_equal(Y.to_numpy(), cref.spmv(m, x))
        Z = mx.MultiVec.from_numpy(ctx1, x)
        A.apply(Z, Z)
        assert np.array_equal(Z.to_numpy(), cref.spmv(m, x))
        # compressed forms on the ghosted matrix, over real ncclSend/ncclRecv.  Compressed stream (MXG_PAT=0): interior
        # blocks first, boundary blocks read the ghost buffer after the exchange (k_spmv_cs<..., HALO>).  Tile form
        # (forced: coverage threshold 0): tiles and ghost-free groups first, ghost-reading groups behind the halo event,
        # two columns in one pass.
        for pat, cov, form in (("0", None, 1), ("1", "0", 3)):
            os.environ["MXG_PAT"] = pat
            if cov is not None:
                os.environ["MXG_TILE_MIN_COVERAGE"] = cov
            try:
                nb, nfall, _ = A.compress()
            finally:
                os.environ.pop("MXG_PAT", None); os.environ.pop("MXG_TILE_MIN_COVERAGE", None)
            assert nb > 0 and A.product_form()[0] == form
            for rep in range(2):
                x3 = rng.uniform(-1, 1, (n, 1))
                X3 = mx.MultiVec.from_numpy(ctx1, x3); Y3 = mx.MultiVec(ctx1, n, 1)
                A.apply(X3, Y3)
                assert np.array_equal(Y3.to_numpy(), cref.spmv(m, x3))
            A.apply(X, Y)
            assert np.array_equal(Y.to_numpy(), cref.spmv(m, x))
        # fused CG (SpMV + dot with the halo in flight) on the ghosted matrix
        spd = (m + m.T).tocsr() * 0.5 + sp.identity(n) * 40.0
        cols2 = spd.indices.astype(np.int64).copy()
        rows2 = np.repeat(np.arange(n), np.diff(spd.indptr))
        rer = (slot[cols2] >= 0) & (rng.uniform(0, 1, len(cols2)) < 0.7) & (cols2 != rows2)   # a diagonal is never a ghost
        cols2[rer] = slot[cols2[rer]]
        rp2 = np.ascontiguousarray(spd.indptr, dtype=np.int64); ci2 = np.ascontiguousarray(cols2, dtype=np.int32)
        va2 = np.ascontiguousarray(spd.data)
        S = mx.CsrMatrix(ctx1, n, n, rp2.ctypes.data_as(C.c_void_p), ci2.ctypes.data_as(C.c_void_p),
                         va2.ctypes.data_as(C.c_void_p), False, nghost=len(ghosted), plan=C.byref(plan))
        bvec = rng.uniform(-1, 1, n)
        Bv = mx.MultiVec.from_numpy(ctx1, bvec); Xs = mx.MultiVec(ctx1, n, 1); Xs.fill(0.0)
        it, conv, res = S.solve(Bv, Xs, method="cg", tol=1e-10, precond="jacobi")
        assert conv and np.linalg.norm(spd @ Xs.to_numpy()[:, 0] - bvec) < 1e-8
        for pat, cov in (("0", None), ("1", "0")):   # the same solve from the compressed stream / the tile form (fused p.Ap with ghosts)
            os.environ["MXG_PAT"] = pat
            if cov is not None:
                os.environ["MXG_TILE_MIN_COVERAGE"] = cov
            try:
                S.compress()
            finally:
                os.environ.pop("MXG_PAT", None); os.environ.pop("MXG_TILE_MIN_COVERAGE", None)
            Xs.fill(0.0)
            it2, conv2, _ = S.solve(Bv, Xs, method="cg", tol=1e-10, precond="jacobi")
            assert conv2 and abs(it2 - it) <= 2 and np.linalg.norm(spd @ Xs.to_numpy()[:, 0] - bvec) < 1e-8
        S.close()
        # the collectives of C2 on the same communicator
        assert np.abs(X.norm2() - np.linalg.norm(x, axis=0)).max() < 1e-12
        ctx1.barrier()
        for obj in (A, X, Y):
            obj.close()
    finally:
        ctx1.close()


def test_config3_cylinder_and_config5_sphere_geometries(ctx, rng):
    """Parity cases of BASELINE configs 3 and 5 (geometry only: curved PEC cylinder periodic in z;
    PEC sphere octant with PEC/PMC walls) at test size: bit-exact SpMV and SpMM (8 right-hand sides)."""
    cases = [dict(n=14, origin=(-0.5,) * 3, size=(1.0,) * 3, pec_shape="cylinder(0.4;0,0,1;0,0,0)"),
             dict(n=[12, 13, 14], origin=(0.0,) * 3, size=(0.5,) * 3, pec_shape="sphere(0.49;0,0,0)",
                  lower=("pmc", "pec", "pmc"), upper=("pec", "pec", "pec"))]
    for kw in cases:
        n = kw.pop("n")
        sim = mx.Sim(n, **kw)
        for name in ("vecLapl", "curlCurl"):
            host = sim.op(name)
            A = sim.to_device(ctx, name)
            x = rng.uniform(-1, 1, (host.shape[1], 8))
            X = mx.MultiVec.from_numpy(ctx, x); Y = mx.MultiVec(ctx, host.shape[0], 8)
            A.apply(X, Y)
            assert np.array_equal(Y.to_numpy(), cref.spmv(host, x))


# MXG_PAT at compress time: "tile" forces the tile form (coverage threshold 0), "cs" the compressed stream, "default"
# lets mxg_csr_compress choose (tile form where the tiles cover >= 80 % of the rows)
CS_MODES = {"tile": ("1", {"MXG_TILE_MIN_COVERAGE": "0"}), "cs": ("0", {}), "default": (None, {})}


def _cs_env(monkeypatch, mode):
    if CS_MODES[mode][0] is None:
        monkeypatch.delenv("MXG_PAT", raising=False)
    else:
        monkeypatch.setenv("MXG_PAT", CS_MODES[mode][0])
    monkeypatch.delenv("MXG_TILE_MIN_COVERAGE", raising=False)
    for k, v in CS_MODES[mode][1].items():
        monkeypatch.setenv(k, v)


@pytest.mark.parametrize("mode", list(CS_MODES))
def test_compressed_stream_bit_identical(ctx, oracle_pillbox16, rng, mode, monkeypatch):
    """mxg_csr_compress: every operator of the pillbox, SpMV and SpMM, and the fused SpMV+dot of CG give the same bits as
    the serial row loop.  Modes: "tile" wavefront-uniform tile patterns through scalar loads, irregular rows as ELL groups
    in the same launch, all columns of a block of vectors in one pass (k_spmv_tile); "cs" 16-bit band codes + 8-bit
    value-table indices per row block (k_spmv_cs: bulk blocks pack, cut-cell blocks fall back to plain CSR rows)."""
    _cs_env(monkeypatch, mode)
    _, o = oracle_pillbox16
    packed_any = False
    for name in ("vecLapl", "scaLapl", "curlCurl", "curlE", "divB", "gradPsi", "mRhs"):
        m = o[name].tocsr()
        A = mx.CsrMatrix.from_scipy(ctx, m)
        nb, nf, nbytes = A.compress()
        assert nb > 0 and 0 <= nf <= nb
        packed_any = packed_any or nf < nb
        for k in (1, 3, 8):
            x = rng.uniform(-1, 1, (m.shape[1], k))
            X = mx.MultiVec.from_numpy(ctx, x); Y = mx.MultiVec(ctx, m.shape[0], k)
            A.apply(X, Y)
            got = Y.to_numpy()
            for j in range(k):
                assert np.array_equal(got[:, j], cref.spmv(m, x[:, j])), (name, k, j)
    assert packed_any
    # the solver path: CG with the fused SpMV+dot on a compressed matrix reaches the same solution
    m = yee.shifted(o["vecLapl"], o["mRhs"], -3.0).tocsr()
    b = rng.uniform(-1, 1, m.shape[0]) * (o["mRhs"].diagonal() > 0)
    sols = []
    for comp in (False, True):
        A = mx.CsrMatrix.from_scipy(ctx, m)
        if comp:
            A.compress()
        B = mx.MultiVec.from_numpy(ctx, b); X = mx.MultiVec(ctx, m.shape[0], 1); X.fill(0.0)
        it, conv, res = A.solve(B, X, method="cg", tol=1e-9, max_iters=5000, precond="jacobi")
        assert conv
        sols.append((it, X.to_numpy()[:, 0]))
    # the products are bit-identical; the fused p.Ap partial sums are combined in a different (fixed) order
    assert abs(sols[0][0] - sols[1][0]) <= 2
    assert np.abs(sols[0][1] - sols[1][1]).max() <= 1e-8 * np.abs(sols[0][1]).max()


@pytest.mark.parametrize("mode", list(CS_MODES))
def test_compressed_stream_fallbacks(ctx, rng, mode, monkeypatch):
    """Blocks with more than 16 column bands or 256 distinct values, empty rows and an over-long row."""
    _cs_env(monkeypatch, mode)
    n = 9000
    rows = [rng.choice(n, size=int(c), replace=False) for c in rng.integers(0, 12, size=n)]   # scattered columns
    rows[5] = np.arange(n)[:2600]                        # longer than a block
    for r in range(700, 1500):
        rows[r] = np.array([], dtype=int)
    for r in range(3000, 6000):                          # banded part with few distinct values: packs
        rows[r] = np.array([c for c in (r - 600, r - 1, r, r + 1, r + 600) if 0 <= c < n])
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])])
    indices = np.concatenate([np.sort(r) for r in rows])
    vals = rng.uniform(-1, 1, len(indices))
    lo, hi = indptr[3000], indptr[6000]
    vals[lo:hi] = rng.choice([-1.0, 4.0, 0.25], size=hi - lo)
    m = sp.csr_matrix((vals, indices, indptr), shape=(n, n))
    x = rng.uniform(-1, 1, n)
    want = cref.spmv(m, x)
    A = mx.CsrMatrix.from_scipy(ctx, m)
    nb, nf, nbytes = A.compress()
    if mode == "tile":                                    # forced tile form (coverage threshold 0): tiles + irregular rows, no records
        assert nb > 0 and nf == 0 and A.product_form()[0] == 3
    else:
        assert 0 < nf < nb                                # both kinds of blocks are present
    X = mx.MultiVec.from_numpy(ctx, x); Y = mx.MultiVec(ctx, n, 1)
    A.apply(X, Y)
    got = Y.to_numpy()[:, 0]
    mask = np.ones(n, bool); mask[5] = False
    assert np.array_equal(got[mask], want[mask])
    assert abs(got[5] - want[5]) <= 1e-12 * np.abs(m[5].data).sum()


def test_tile_form_complex_operator(ctx, rng, monkeypatch):
    """Complex lattice operator (Bloch phases across the periodic seams): tiles carry (re, im) value lists, one row per
    lane; blocks of vectors and scattered views; bit-identical to the complex row loop of the oracle."""
    monkeypatch.setenv("MXG_TILE_MIN_COVERAGE", "0")
    s = yee.YeeSim(12, strict=False, phase_shifts=(0.3, 0.0, 0.7))
    o = s.wave_ops()
    m = yee.shifted(o["vecLapl"], o["mRhs"], 0.5).tocsr().astype(np.complex128)
    A = mx.CsrMatrix.from_scipy(ctx, m)
    nb, nf, _ = A.compress()
    form, tiles, pats, irregular, cov = A.product_form()
    assert form == 3 and tiles > 0 and cov > 0.2
    x = rng.uniform(-1, 1, (m.shape[1], 5)) + 1j * rng.uniform(-1, 1, (m.shape[1], 5))
    X = mx.MultiVec.from_numpy(ctx, x); Y = mx.MultiVec(ctx, m.shape[0], 5, True)
    A.apply(X, Y)
    assert np.array_equal(Y.to_numpy(), cref.spmv(m, x))
    Y.fill(0.0)
    A.apply(X.view([4, 0, 2]), Y.view([1, 3, 0]))
    got = Y.to_numpy()
    assert np.array_equal(got[:, [1, 3, 0]], cref.spmv(m, x[:, [4, 0, 2]])) and np.all(got[:, [2, 4]] == 0)
    A.apply(X, X)                                        # aliasing: works from a copy of x
    assert np.array_equal(X.to_numpy(), cref.spmv(m, x))
    # 13-point curl-curl: per-component shapes, read per lane from the dictionary
    m = o["curlCurl"].tocsr().astype(np.complex128)
    A = mx.CsrMatrix.from_scipy(ctx, m)
    A.compress()
    form, tiles, pats, irregular, cov = A.product_form()
    assert form == 3 and cov > 0.5
    x = rng.uniform(-1, 1, (m.shape[1], 2)) + 1j * rng.uniform(-1, 1, (m.shape[1], 2))
    X = mx.MultiVec.from_numpy(ctx, x); Y = mx.MultiVec(ctx, m.shape[0], 2, True); Y.fill(float("nan"))
    A.apply(X, Y)
    assert np.array_equal(Y.to_numpy(), cref.spmv(m, x))


def test_tile_form_long_patterns_and_wide_groups(ctx, rng, monkeypatch):
    """A one-component 13-point stencil: long-pattern tiles (9..16 entries, one tile per wavefront); rows of 33..40
    entries are ELL group rows, rows of more entries long rows; SpMV, SpMM and the fused epilogues."""
    monkeypatch.setenv("MXG_TILE_MIN_COVERAGE", "0")
    n = 6000
    offs = np.array([-900, -600, -300, -30, -20, -10, 0, 10, 20, 30, 300, 600, 900])
    vals = rng.uniform(-1, 1, 13)
    rows, cols, data = [], [], []
    for r in range(n):
        if 1000 <= r < 4000:
            rows += [r] * 13; cols += list(r + offs); data += list(vals)
        elif r % 7 == 0:
            c = np.sort(rng.choice(n, size=int(rng.integers(30, 48)), replace=False))
            rows += [r] * len(c); cols += list(c); data += list(rng.uniform(-1, 1, len(c)))
        else:
            rows += [r]; cols += [r]; data += [2.0]
    m = sp.csr_matrix((data, (rows, cols)), shape=(n, n)); m.sort_indices()
    A = mx.CsrMatrix.from_scipy(ctx, m)
    A.compress()
    assert A.product_form()[0] == 3
    x = rng.uniform(-1, 1, (n, 4))
    X = mx.MultiVec.from_numpy(ctx, x); Y = mx.MultiVec(ctx, n, 4)
    A.apply(X, Y)
    assert np.array_equal(Y.to_numpy(), cref.spmv(m, x))


def test_recompress_after_switching_the_form(ctx, oracle_pillbox16, rng, monkeypatch):
    """mxg_csr_compress follows MXG_PAT at the time of the call: tile form -> compressed stream -> tile form on one
    matrix, and mxg_csr_set_kernel returns to the CSR kernels in between (a stale form must never be
    launched with an empty grid)."""
    _, o = oracle_pillbox16
    m = yee.shifted(o["vecLapl"], o["mRhs"], 1.97).tocsr()
    x = rng.uniform(-1, 1, m.shape[1])
    want = cref.spmv(m, x)
    A = mx.CsrMatrix.from_scipy(ctx, m)
    X = mx.MultiVec.from_numpy(ctx, x); Y = mx.MultiVec(ctx, m.shape[0], 1)
    monkeypatch.setenv("MXG_TILE_MIN_COVERAGE", "0")
    for pat, form in (("1", 3), ("0", 1), ("1", 3)):
        monkeypatch.setenv("MXG_PAT", pat)
        A.compress()
        assert A.product_form()[0] == form
        Y.fill(0.0); A.apply(X, Y)
        assert np.array_equal(Y.to_numpy()[:, 0], want)
        A.set_kernel(0)
        assert A.product_form()[0] == 0
        Y.fill(0.0); A.apply(X, Y)
        assert np.array_equal(Y.to_numpy()[:, 0], want)
