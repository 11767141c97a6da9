"""The product's host solver stack (MxSolver, block Krylov-Schur, MxMagWaveOp::Apply, the Krylov loops of mxg_krylov.h,
the multigrid hierarchy builder) WITHOUT a GPU: it runs unchanged over the CPU backend of the C ABI
(oracle/cpu_backend/mxg_cpu.cpp - the baseline bench.py times as `cpu_baseline.eigen`) in a child process, and its
eigenvalues must match the oracle's shift-invert of the same assembled pencil to north_star's 1e-8."""
import numpy as np
import pytest

from oracle import cpu_backend, yee


def _oracle_physical_modes(o, shift, want):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    L, M, D = o["vecLapl"], o["mRhs"], o["divB"]
    d = M.diagonal(); keep = d > 0
    vals, vecs = spla.eigsh(L[keep][:, keep].tocsc(), k=4 * want + 4, M=sp.diags(d[keep]).tocsc(), sigma=shift, which="LM", tol=1e-13)
    out = []
    for lam, v in zip(vals, vecs.T):
        full = np.zeros(L.shape[0]); full[keep] = v
        if np.linalg.norm(D @ (M @ full)) / np.linalg.norm(full) < 1e-6:
            out.append(lam)
    return np.sort(out)[:want]


def test_cpu_backend_eigenvalues_match_oracle(oracle_pillbox16):
    _, o = oracle_pillbox16
    want = _oracle_physical_modes(o, yee.default_shift([1, 1, 1]), 5)
    r = cpu_backend.eigen(16, nev=5, tol=1e-10, lin_tol=1e-11, threads=4)
    assert r["converged"] == 5
    k2 = np.sort(r["k2"])
    assert np.abs(k2 - want).max() / np.abs(want).max() < 1e-8, (k2, want)
    assert r["max_rel_residual"] < 1e-7
    assert r["applies"] >= 5 and r["vec_lin_iters"] > 0 and r["sca_lin_iters"] > 0


def test_cpu_backend_is_not_reachable_from_the_product():
    """The CPU libraries live under oracle/_cpu and are loaded only when a caller points _lib._LIBDIR there."""
    import os
    from maxwell_amd import _lib
    assert os.path.basename(_lib._LIBDIR) == "lib" and "oracle" not in _lib._LIBDIR


def test_block_davidson_finds_the_same_modes(oracle_pillbox16):
    """`eigensolver : type = davidson` (Anasazi::BlockDavidsonSolMgr with locking, src/MxSolver.cpp:88-91): the block
    Davidson manager of AnasaziCompat.hpp on the host solver stack - the eigenvalues of the Krylov-Schur solve and of
    the oracle's shift-invert of the same pencil."""
    _, o = oracle_pillbox16
    want = _oracle_physical_modes(o, yee.default_shift([1, 1, 1]), 5)
    ks = cpu_backend.eigen(16, nev=5, tol=1e-10, lin_tol=1e-11, threads=4)
    dv = cpu_backend.eigen(16, nev=5, tol=1e-10, lin_tol=1e-11, threads=4, eig_type="davidson")
    assert dv["converged"] == 5
    k2 = np.sort(dv["k2"])
    assert np.abs(k2 - want).max() / np.abs(want).max() < 1e-8, (k2, want)
    assert np.abs(k2 - np.sort(ks["k2"])).max() / np.abs(want).max() < 1e-8
    assert dv["max_rel_residual"] < 1e-7
    # no preconditioner of its own: the same Krylov spaces, about the same number of operator applications
    assert dv["applies"] <= 2 * ks["applies"] + 10, (dv["applies"], ks["applies"])
