"""Host-side plan of the tile product (mxg_tile.hip) checked WITHOUT a GPU: mxg_debug_tile_plan_apply builds the plan
(tiles of short / long patterns, ELL groups of irregular rows with the ghost-free ones first, long rows) and evaluates
y = A x by walking the plan with the kernels' arithmetic; the result must equal the oracle's serial row loop bit for
bit and every row must be produced exactly once.  The kernels themselves are checked on the GPU
(tests/test_spmv_gpu.py, tests/test_fullsize_gpu.py)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import cref, yee


def _plan_apply(built, m, x, ncols_owned=None, xghost=None):
    from maxwell_amd import _lib
    lib = _lib.load_mxg()
    m = m.tocsr()
    cplx = np.iscomplexobj(m.data) or np.iscomplexobj(x)
    n = m.shape[0]
    nown = m.shape[1] if ncols_owned is None else ncols_owned
    rp = np.ascontiguousarray(m.indptr, dtype=np.int64)
    ci = np.ascontiguousarray(m.indices, dtype=np.int32)
    dt = np.complex128 if cplx else np.float64
    va = np.ascontiguousarray(m.data, dtype=dt)
    xx = np.ascontiguousarray(x, dtype=dt)
    xg = np.ascontiguousarray(xghost if xghost is not None else np.zeros(1), dtype=dt)
    y = np.full(n, np.nan, dtype=dt)
    stats = np.zeros(9, dtype=np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.mxg_debug_tile_plan_apply(n, nown, p(rp), p(ci), p(va), 1 if cplx else 0, p(xx), p(xg), p(y),
                                       stats.ctypes.data_as(C.POINTER(C.c_int64)))
    keys = ("tilesA", "tilesB", "patterns", "groups", "groups_interior", "long_rows", "covered", "missing", "errors")
    return rc, y, dict(zip(keys, stats.tolist()))


@pytest.mark.parametrize("name", ["vecLapl", "scaLapl", "curlCurl", "divB", "gradPsi", "mRhs"])
def test_pillbox_operators(built, oracle_pillbox16, rng, name):
    _, o = oracle_pillbox16
    m = o[name].tocsr()
    x = rng.uniform(-1, 1, m.shape[1])
    rc, y, st = _plan_apply(built, m, x)
    assert rc == 0 and st["missing"] == 0 and st["errors"] == 0, st
    assert np.array_equal(y, cref.spmv(m, x)), name
    if name == "vecLapl":
        assert st["tilesA"] > 0 and st["covered"] > 0.3 * m.shape[0]


def test_shifted_operator_larger_grid(built, rng):
    import maxwell_amd as mx
    sim = mx.Sim.pillbox(40)
    m = sim.op("vecLapl", shift=1.9739208802178716)
    x = rng.uniform(-1, 1, m.shape[1])
    rc, y, st = _plan_apply(built, m, x)
    assert rc == 0, st
    assert np.array_equal(y, cref.spmv(m, x))
    assert st["covered"] > 0.6 * m.shape[0] and st["long_rows"] == 0


def test_complex_operator(built, rng):
    """Bloch phases make the seam entries complex; the bulk tiles carry (re, im) value lists."""
    s = yee.YeeSim(10, strict=False, phase_shifts=(0.3, 0.0, 0.7))
    o = s.wave_ops()
    m = yee.shifted(o["vecLapl"], o["mRhs"], 0.5).tocsr().astype(np.complex128)
    x = rng.uniform(-1, 1, m.shape[1]) + 1j * rng.uniform(-1, 1, m.shape[1])
    rc, y, st = _plan_apply(built, m, x)
    assert rc == 0, st
    assert np.array_equal(y, cref.spmv(m, x))
    assert st["tilesA"] > 0


def test_complex_tiles_with_per_component_shapes(built, rng):
    """The 13-point curl-curl: the three components of a cell have different column offsets.  A complex lane holds
    one row and reads its own component's shape from the dictionary, so these rows tile (a real lane pairs two rows on
    shared gathers and needs one shape: the real curl-curl has no tiles)."""
    s = yee.YeeSim(12, strict=False, phase_shifts=(0.3, 0.0, 0.7))
    m = s.wave_ops()["curlCurl"].tocsr()
    x = rng.uniform(-1, 1, m.shape[1]) + 1j * rng.uniform(-1, 1, m.shape[1])
    rc, y, st = _plan_apply(built, m.astype(np.complex128), x)
    assert rc == 0, st
    assert np.array_equal(y, cref.spmv(m.astype(np.complex128), x))
    assert st["covered"] > 0.5 * m.shape[0], st
    rc, y, st = _plan_apply(built, m.real.tocsr(), x.real)
    assert rc == 0 and st["covered"] < 0.2 * m.shape[0], st
    assert np.array_equal(y, cref.spmv(m.real.tocsr(), x.real))


def test_ghost_columns_are_group_rows(built, oracle_pillbox12, rng):
    """Part of the columns re-routed to ghost slots: no tile and no ghost-free group may read them."""
    _, o = oracle_pillbox12
    m = yee.shifted(o["vecLapl"], o["mRhs"], 1.97).tocsr()
    n = m.shape[0]
    ghosted = np.nonzero(np.arange(n) % 5 == 0)[0]
    slot = -np.ones(n, dtype=np.int64); slot[ghosted] = n + np.arange(len(ghosted))
    cols = m.indices.astype(np.int64).copy()
    reroute = (slot[cols] >= 0) & (rng.uniform(0, 1, len(cols)) < 0.7)
    cols[reroute] = slot[cols[reroute]]
    mg = sp.csr_matrix((m.data, cols, m.indptr), shape=(n, n + len(ghosted)))
    x = rng.uniform(-1, 1, n)
    rc, y, st = _plan_apply(built, mg, x, ncols_owned=n, xghost=x[ghosted])
    assert rc == 0 and st["errors"] == 0, st
    assert np.array_equal(y, cref.spmv(m, x))
    assert 0 < st["groups_interior"] < st["groups"]


def test_ragged_matrix_with_long_rows(built, rng):
    n = 3000
    rows = [rng.choice(n, size=int(c), replace=False) for c in rng.integers(0, 30, size=n)]
    rows[17] = np.arange(n)[:2500]
    rows[40] = np.arange(n)[5:50]                   # 45 entries: a long row, still summed in storage order
    for r in range(100, 900):
        rows[r] = np.array([], dtype=int)
    for r in range(1500, 2500):                     # banded part: tiles
        rows[r] = np.array([c for c in (r - 300, r - 3, r, r + 3, r + 300) if 0 <= c < n])
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])])
    indices = np.concatenate([np.sort(r) for r in rows])
    vals = rng.uniform(-1, 1, len(indices))
    lo, hi = indptr[1500], indptr[2500]
    vals[lo:hi] = np.tile([-1.0, 0.5, 4.0, 0.5, -1.0], 1000)
    m = sp.csr_matrix((vals, indices, indptr), shape=(n, n))
    x = rng.uniform(-1, 1, n)
    rc, y, st = _plan_apply(built, m, x)
    assert rc == 0, st
    assert np.array_equal(y, cref.spmv(m, x))       # long rows are sequential sums here: exact
    assert st["long_rows"] >= 2 and st["tilesA"] > 0 and st["groups"] > 0


def test_long_patterns_and_rows_up_to_group_width(built, rng):
    """A one-component 13-point stencil (every row the same shape and values): long-pattern tiles; rows of 33..40
    entries are still group rows."""
    n = 4000
    offs = np.array([-900, -600, -300, -30, -20, -10, 0, 10, 20, 30, 300, 600, 900])
    vals = rng.uniform(-1, 1, 13)
    rows, cols, data = [], [], []
    for r in range(n):
        if 1000 <= r < 3000:
            rows += [r] * 13; cols += list(r + offs); data += list(vals)
        elif r % 7 == 0:
            c = np.sort(rng.choice(n, size=int(rng.integers(33, 41)), replace=False))
            rows += [r] * len(c); cols += list(c); data += list(rng.uniform(-1, 1, len(c)))
        else:
            rows += [r]; cols += [r]; data += [2.0]
    m = sp.csr_matrix((data, (rows, cols)), shape=(n, n)); m.sort_indices()
    x = rng.uniform(-1, 1, n)
    rc, y, st = _plan_apply(built, m, x)
    assert rc == 0, st
    assert np.array_equal(y, cref.spmv(m, x))
    assert st["tilesB"] > 0 and st["long_rows"] == 0


def test_empty_and_tiny(built):
    rc, y, st = _plan_apply(built, sp.csr_matrix((3, 3)), np.ones(3))
    assert rc == 0 and np.array_equal(y, np.zeros(3))
    rc, y, st = _plan_apply(built, sp.csr_matrix(np.array([[2.0]])), np.array([3.0]))
    assert rc == 0 and y[0] == 6.0
