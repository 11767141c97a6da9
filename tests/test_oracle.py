"""Pin the oracle: structural identities, exact discrete spectrum of the periodic vacuum box,
analytic pillbox modes (example/pillbox.py geometry) and the committed golden fixtures."""
import json
import os

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import yee, geometry as geo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_periodic_identities(oracle_periodic6):
    s, o = oracle_periodic6
    assert abs(o["curlE"] - o["curlB"].T).max() == 0.0       # curlE = curlB^T
    assert abs(o["gradPsi"] + o["divB"].T).max() == 0.0      # gradPsi = -divB^T
    assert set(np.diff(o["curlE"].indptr)) == {4} and set(np.diff(o["curlB"].indptr)) == {4}
    assert set(np.diff(o["divB"].indptr)) == {6} and set(np.diff(o["gradPsi"].indptr)) == {2}
    assert set(np.diff(o["curlCurl"].indptr)) == {13}
    assert set(np.diff(o["gradDiv"].indptr)) == {11}
    assert set(np.diff(o["vecLapl"].indptr)) == {7}          # purgeZeros collapses 15 -> 7
    assert set(np.diff(o["scaLapl"].indptr)) == {7}
    # div curl = 0 and curl grad = 0 on the lattice
    assert abs(o["divB"] @ o["curlE"]).max() < 1e-12
    assert abs(o["curlB"] @ o["gradPsi"]).max() < 1e-12


def test_periodic_vacuum_exact_spectrum(oracle_periodic6):
    """vecLapl = three decoupled 7-point Laplacians: k^2 = (4/h^2) sum_i sin^2(pi m_i / N)."""
    s, o = oracle_periodic6
    N, h = 6, 1.0 / 6
    w = np.sort(np.linalg.eigvalsh(o["vecLapl"].toarray()))
    exact = np.sort([(4 / h**2) * sum(np.sin(np.pi * np.array(m) / N) ** 2)
                     for m in np.ndindex(N, N, N)] * 3)
    assert np.abs(w - exact).max() < 1e-10


def test_reference_periodic_wrap_quirk_documented():
    """strict mode (reference as written) drops the wrap-around E entries of curlE at the upper
    periodic boundary (src/MxYeeFitEField.cpp:96 + src/MxGridField.cpp:70-73)."""
    strict = yee.YeeSim(6, strict=True).curlE()
    fixed = yee.YeeSim(6, strict=False).curlE()
    assert strict.nnz == 2376 and fixed.nnz == 2592


def test_pillbox_unaffected_by_wrap_quirk():
    a = yee.pillbox_sim(8, strict=True).wave_ops()
    b = yee.pillbox_sim(8, strict=False).wave_ops()
    for k in a:
        assert a[k].shape == b[k].shape and (a[k] != b[k]).nnz == 0, k


def test_fraction_sums_match_geometry(oracle_pillbox16):
    """Sum of z-face area fractions in one interior z layer ~ pi R^2 / h^2; sum of cell volume
    fractions ~ pi R^2 L / h^3 (Dey-Mittra fractions are second-order accurate)."""
    s, _ = oracle_pillbox16
    h = 1.0 / 16
    fz = s.B.frac[2][1:-1, 1:-1, 1 + 8]            # z-normal faces at k = 8 (inside the cavity)
    assert abs(fz.sum() * h * h - np.pi * 0.16) < 4e-3   # chords under-estimate arcs by O(h^2)
    vol = (1.0 / s.fracs(s.P, inverse=True).diagonal()[s.fracs(s.P, inverse=True).diagonal() > 0]).sum()
    assert abs(vol * h**3 - np.pi * 0.16 * 0.8) < 6e-3


def test_pillbox_operator_structure(oracle_pillbox16):
    s, o = oracle_pillbox16
    L, A, M = o["vecLapl"], o["curlCurl"], o["mRhs"]
    assert abs(A - A.T).max() == 0.0                      # curlE dmL curlB symmetric (PEC only)
    assert abs(L - L.T).max() < 1e-11 * abs(L).max()
    assert (M - sp.diags(M.diagonal())).nnz == 0 and M.diagonal().min() >= 0.0
    empty = np.diff(L.indptr) == 0
    assert empty.sum() > 0 and np.all(M.diagonal()[empty] == 0.0)   # unused comps: zero rows, zero mass
    assert np.diff(L.indptr).max() <= 15


def _physical_modes(o, shift, k):
    """Shift-invert on the pencil the reference solves, then the projection of Apply
    (src/MxMagWaveOp.cpp:895-922) removes the gradient modes: keep modes with ~zero divergence."""
    L, M, D = o["vecLapl"], o["mRhs"], o["divB"]
    d = M.diagonal()
    keep = d > 0
    vals, vecs = spla.eigsh(L[keep][:, keep].tocsc(), k=k, M=sp.diags(d[keep]).tocsc(), sigma=shift, which="LM")
    out = []
    for lam, v in zip(vals, vecs.T):
        full = np.zeros(L.shape[0]); full[keep] = v
        div = np.linalg.norm(D @ (M @ full)) / np.linalg.norm(full)
        if div < 1e-6:
            out.append(lam)
    return np.sort(out)


def test_pillbox_converges_to_analytic_modes():
    """TM010: k^2 = (x01/R)^2, TE111: k^2 = (x'11/R)^2 + (pi/L)^2 (doubly degenerate);
    the Dey-Mittra discretisation is second order, so the error at N=24 is O(h^2)."""
    tm010 = (2.404825557695773 / 0.4) ** 2
    te111 = (1.8411837813406593 / 0.4) ** 2 + (np.pi / 0.8) ** 2
    o = yee.pillbox_sim(24).wave_ops()
    modes = _physical_modes(o, yee.default_shift([1, 1, 1]), 10)
    assert len(modes) >= 3
    assert abs(modes[0] - tm010) / tm010 < 2e-3
    assert abs(modes[1] - te111) / te111 < 2e-3 and abs(modes[2] - te111) / te111 < 2e-3
    assert abs(modes[1] - modes[2]) / te111 < 1e-6     # degenerate pair


def test_golden_fixtures(oracle_pillbox12):
    """Committed fixture written by tests/golden/make_golden.py from this oracle."""
    s, o = oracle_pillbox12
    g = np.load(os.path.join(GOLD, "pillbox12.npz"))
    meta = json.load(open(os.path.join(GOLD, "pillbox12.json")))
    assert len(s.B.gids) == meta["n_b"] and len(s.E.gids) == meta["n_e"] and len(s.P.gids) == meta["n_psi"]
    L = o["vecLapl"]
    assert np.array_equal(L.indptr, g["vecLapl_indptr"]) and np.array_equal(L.indices, g["vecLapl_indices"])
    assert np.array_equal(L.data, g["vecLapl_data"])
    assert np.array_equal(o["mRhs"].diagonal(), g["dmA"])
    assert np.array_equal(s.B.gids, g["b_gids"])


def test_root_find_and_fractions_basic():
    cyl = geo.Cylinder(0.4, [0, 0, 1], [0, 0, 0])
    p = geo.root_find(cyl, (0.0, 0.0, 0.0), (1.0, 0.0, 0.0))
    assert abs(p[0] - 0.4) < 1e-12
    half = geo.HalfSpace([0, 0, 0.25], [0, 0, 1])
    assert geo.segment_fractions(half, np.array([[0.0, 0.0, 0.25]]), 2, 1.0)[0] == 0.5
    assert abs(geo.rect_fractions(half, np.array([[0.0, 0.0, 0.3]]), 0, 1.0, 1.0)[0] - 0.55) < 1e-11
    assert abs(geo.box_fractions(half, np.array([[0.0, 0.0, 0.3]]), 1.0, 1.0, 1.0)[0] - 0.55) < 1e-11
