"""a1 / a9 + the caller: MxSolver (block Krylov-Schur, shift-invert) around MxMagWaveOp::Apply on the
GPU against the oracle's eigenvalues of the SAME assembled pencil (scipy shift-invert with a sparse
direct solve), to the 1e-8 relative tolerance north_star states; plus the analytic pillbox modes."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import maxwell_amd as mx
from oracle import yee

pytestmark = pytest.mark.gpu
REL_TOL = 1e-8          # BASELINE.json north_star: "eigenvalues within 1e-8 rel. of reference"


def _oracle_physical_modes(o, shift, want):
    L, M, D = o["vecLapl"], o["mRhs"], o["divB"]
    d = M.diagonal(); keep = d > 0
    vals, vecs = spla.eigsh(L[keep][:, keep].tocsc(), k=4 * want + 4, M=sp.diags(d[keep]).tocsc(), sigma=shift,
                            which="LM", tol=1e-13)
    out = []
    for lam, v in zip(vals, vecs.T):
        full = np.zeros(L.shape[0]); full[keep] = v
        if np.linalg.norm(D @ (M @ full)) / np.linalg.norm(full) < 1e-6:
            out.append(lam)
    return np.sort(out)[:want]


@pytest.mark.parametrize("lin_type,prec", [("cg", "jacobi"), ("gmres", "none"), ("cg", "amg"), ("bicgstab", "amg")])
def test_pillbox16_eigenvalues_match_oracle(ctx, oracle_pillbox16, lin_type, prec):
    _, o = oracle_pillbox16
    shift = yee.default_shift([1, 1, 1])
    want = _oracle_physical_modes(o, shift, 5)
    sim = mx.Sim.pillbox(16)
    params = mx.SolverParams(nev=5, basis=30, tol=1e-10, shift=shift, lin_type=lin_type, lin_basis=40,
                             lin_tol=1e-11, prec_type=prec, block_size=1)
    solver = mx.Solver(ctx, sim, params)
    nconv = solver.solve()
    assert nconv == 5
    res = [solver.result(i) for i in range(nconv)]
    k2 = np.sort([r["k2"].real for r in res])
    assert np.abs(k2 - want).max() / np.abs(want).max() < REL_TOL, (k2, want)
    for r in res:
        assert abs(r["k2"].imag) == 0.0
        assert r["residual"] < 1e-7          # ||curlCurl v - k^2 M v|| / |k^2| as the reference prints it
        assert r["divergence"] < 1e-6        # ||divB M v||
        f = np.sqrt(r["k2"].real) * 299792458.0 / (2 * np.pi)
        assert abs(r["freq"].real - f) < 1e-6 * f
    st = solver.stats()
    assert st["applies"] >= 5 and st["vec_iters"] > 0 and st["sca_iters"] > 0 and st["eigensolve_seconds"] > 0


def test_degenerate_pair_with_block_size_two(ctx, oracle_pillbox12):
    """TE111 is doubly degenerate: block size 2 must deliver both copies."""
    _, o = oracle_pillbox12
    shift = yee.default_shift([1, 1, 1])
    want = _oracle_physical_modes(o, shift, 4)
    sim = mx.Sim.pillbox(12)
    params = mx.SolverParams(nev=4, basis=20, tol=1e-10, shift=shift, lin_type="cg", lin_tol=1e-11,
                             prec_type="jacobi", block_size=2)
    solver = mx.Solver(ctx, sim, params)
    assert solver.solve() == 4
    k2 = np.sort([solver.result(i)["k2"].real for i in range(4)])
    assert np.abs(k2 - want).max() / np.abs(want).max() < REL_TOL, (k2, want)


def test_pillbox32_analytic_modes(ctx):
    """Config 1 (example/pillbox.py at 32^3, nev 5): TM010 and the TE111 pair within the O(h^2)
    discretisation error of the analytic cavity spectrum."""
    tm010 = (2.404825557695773 / 0.4) ** 2
    te111 = (1.8411837813406593 / 0.4) ** 2 + (np.pi / 0.8) ** 2
    sim = mx.Sim.pillbox(32)
    params = mx.SolverParams(nev=3, basis=30, tol=1e-8, shift=yee.default_shift([1, 1, 1]), lin_type="cg",
                             lin_tol=1e-9, prec_type="amg")
    solver = mx.Solver(ctx, sim, params)
    assert solver.solve() == 3
    k2 = np.sort([solver.result(i)["k2"].real for i in range(3)])
    assert abs(k2[0] - tm010) / tm010 < 2e-3
    assert abs(k2[1] - te111) / te111 < 2e-3 and abs(k2[2] - te111) / te111 < 2e-3
    v = solver.eigvec(0, sim.map("bfield").size)
    assert np.all(v[sim.used_mask() == 0] == 0)       # unused components stay zero


def test_block_davidson_on_the_gpu(ctx, oracle_pillbox12):
    """`eigensolver : type = davidson` (src/MxSolver.cpp:88-91) through the C ABI on the device: the block Davidson
    manager returns the eigenvalues of the Krylov-Schur solve and of the oracle's shift-invert of the same pencil, with
    about as many operator applications (no preconditioner of its own: the same Krylov spaces)."""
    _, o = oracle_pillbox12
    shift = yee.default_shift([1, 1, 1])
    want = _oracle_physical_modes(o, shift, 4)
    sim = mx.Sim.pillbox(12)
    out = {}
    for eig_type in ("krylov-schur", "davidson"):
        params = mx.SolverParams(nev=4, basis=20, tol=1e-10, shift=shift, lin_type="cg", lin_tol=1e-11, prec_type="jacobi",
                                 block_size=2, eig_type=eig_type)
        solver = mx.Solver(ctx, sim, params)
        assert solver.solve() == 4
        out[eig_type] = (np.sort([solver.result(i)["k2"].real for i in range(4)]), solver.stats()["applies"])
        solver.close()
    for eig_type, (k2, applies) in out.items():
        assert np.abs(k2 - want).max() / np.abs(want).max() < REL_TOL, (eig_type, k2, want)
    assert out["davidson"][1] <= 2 * out["krylov-schur"][1] + 10
