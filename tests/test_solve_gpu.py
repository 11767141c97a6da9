"""K10: the inner Krylov solves (AztecOO replacement) against a direct solve of the same system."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import maxwell_amd as mx
from oracle import yee

pytestmark = pytest.mark.gpu


def _system(o, shift):
    L, M = o["vecLapl"], o["mRhs"]
    A = yee.shifted(L, M, shift)
    used = M.diagonal() > 0
    return A, used


@pytest.mark.parametrize("method,precond", [("cg", "none"), ("cg", "jacobi"), ("gmres", "none"), ("gmres", "jacobi"),
                                            ("bicgstab", "none"), ("bicgstab", "jacobi")])
def test_shifted_veclapl_solve(ctx, oracle_pillbox12, rng, method, precond):
    """(vecLapl - shift dmA) x = dmA b, the first solve of MxMagWaveOp::Apply
    (src/MxMagWaveOp.cpp:865-885); unused rows are zero rows with zero right-hand side."""
    _, o = oracle_pillbox12
    A, used = _system(o, yee.default_shift([1, 1, 1]))
    rhs = o["mRhs"] @ rng.uniform(-1, 1, A.shape[0])
    assert np.all(rhs[~used] == 0)
    dev = mx.CsrMatrix.from_scipy(ctx, A)
    b = mx.MultiVec.from_numpy(ctx, rhs)
    x = mx.MultiVec(ctx, A.shape[0], 1)
    iters, conv, res = dev.solve(b, x, method=method, tol=1e-9, restart=40, precond=precond)
    assert conv and res <= 1e-9
    sol = x.to_numpy()[:, 0]
    assert np.linalg.norm(A @ sol - rhs) <= 2e-9          # unscaled true residual (AZ_noscaled)
    ref = np.zeros_like(sol)
    ref[used] = spla.spsolve(A[used][:, used].tocsc(), rhs[used])
    assert np.linalg.norm(sol - ref) <= 1e-7 * np.linalg.norm(ref)
    assert np.all(sol[~used] == 0)


def test_scalar_laplacian_solve_two_columns(ctx, oracle_pillbox12, rng):
    """scaLapl psi = divB dmA b (second solve, :903-913): singular with the constant vector in the
    null space; the right-hand side is in the range, CG converges to a solution."""
    _, o = oracle_pillbox12
    S = o["scaLapl"]
    rhs = (o["divB"] @ (o["mRhs"] @ rng.uniform(-1, 1, (o["mRhs"].shape[0], 2))))
    dev = mx.CsrMatrix.from_scipy(ctx, S)
    b = mx.MultiVec.from_numpy(ctx, rhs)
    x = mx.MultiVec(ctx, S.shape[0], 2)
    iters, conv, res = dev.solve(b, x, method="cg", tol=1e-8)
    assert conv
    assert np.linalg.norm(S @ x.to_numpy() - rhs, axis=0).max() <= 2e-8


@pytest.mark.parametrize("method", ["gmres", "bicgstab"])
def test_gmres_nonsymmetric_complex(ctx, rng, method):
    """Config 5 flavour: non-Hermitian complex system (diagonally dominant); GMRES and BiCGStab (AZ_bicgstab)."""
    n = 4000
    R = sp.random(n, n, density=5.0 / n, random_state=3, format="csr")
    A = (sp.diags(np.full(n, 8.0 + 1.0j)) + R + 0.5j * sp.random(n, n, density=3.0 / n, random_state=4, format="csr")).tocsr()
    A.sort_indices()
    rhs = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    dev = mx.CsrMatrix.from_scipy(ctx, A)
    b = mx.MultiVec.from_numpy(ctx, rhs)
    x = mx.MultiVec(ctx, n, 1, True)
    iters, conv, res = dev.solve(b, x, method=method, tol=1e-10, restart=30)
    assert conv
    assert np.linalg.norm(A @ x.to_numpy()[:, 0] - rhs) <= 2e-10


def test_unconverged_is_reported_not_spun(ctx, oracle_pillbox12, rng):
    """The reference loops forever on an unconverged solve (src/MxMagWaveOp.cpp:876); here the
    iteration cap returns converged = False."""
    _, o = oracle_pillbox12
    A, _ = _system(o, yee.default_shift([1, 1, 1]))
    dev = mx.CsrMatrix.from_scipy(ctx, A)
    b = mx.MultiVec.from_numpy(ctx, o["mRhs"] @ rng.uniform(-1, 1, A.shape[0]))
    x = mx.MultiVec(ctx, A.shape[0], 1)
    iters, conv, res = dev.solve(b, x, method="cg", tol=1e-14, max_iters=5)
    assert not conv and iters <= 6 and res > 1e-14


@pytest.mark.parametrize("smoother", ["jacobi", "chebyshev"])
def test_multigrid_preconditioner(ctx, rng, smoother):
    """K11 replacement: the aggregation V-cycle is a fixed symmetric operator (legal for CG), cuts the
    iteration count by an order of magnitude, and the solution is the same.  Smoothers: damped Jacobi and ML's
    default Chebyshev (polynomial in D^-1 A: the cycle stays symmetric)."""
    sim = mx.Sim.pillbox(32)
    shift = yee.default_shift([1, 1, 1])
    for op in ("shifted", "scaLapl"):
        mg = mx.Multigrid(ctx, sim, op, shift, min_coarse=500, smoother=smoother)
        assert len(mg.levels) >= 3 and mg.levels[-1][0] < mg.levels[0][0] // 30
        A = mg.fine_matrix()
        n = A.nrows
        host = sim.op("vecLapl", shift=shift) if op == "shifted" else sim.op("scaLapl")
        # symmetry of the cycle: <u, M v> == <M u, v>
        u, v = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        U, V = mx.MultiVec.from_numpy(ctx, u), mx.MultiVec.from_numpy(ctx, v)
        MU, MV = mx.MultiVec(ctx, n, 1), mx.MultiVec(ctx, n, 1)
        mg.apply(U, MU); mg.apply(V, MV)
        a, b = float(np.dot(u, MV.to_numpy()[:, 0])), float(np.dot(MU.to_numpy()[:, 0], v))
        assert abs(a - b) <= 1e-10 * max(abs(a), abs(b))
        assert a != 0.0
        if op == "shifted":
            rhs = sim.op("mRhs") @ rng.uniform(-1, 1, n)
        else:
            rhs = sim.op("divB") @ (sim.op("mRhs") @ rng.uniform(-1, 1, sim.op("mRhs").shape[0]))
        bvec = mx.MultiVec.from_numpy(ctx, rhs)
        xj, xm = mx.MultiVec(ctx, n, 1), mx.MultiVec(ctx, n, 1)
        itj, cj, _ = A.solve(bvec, xj, method="cg", tol=1e-9, precond="jacobi")
        itm, cm, resm = A.solve(bvec, xm, method="cg", tol=1e-9, precond="mg", mg=mg)
        assert cj and cm and itm * 5 < itj and itm < 60
        assert np.linalg.norm(host @ xm.to_numpy()[:, 0] - rhs) <= 2e-9
        if op == "shifted":   # non-singular: both solves agree
            assert np.linalg.norm(xm.to_numpy() - xj.to_numpy()) <= 1e-7 * np.linalg.norm(xj.to_numpy())


def test_multigrid_hierarchy_is_galerkin(ctx):
    """The hierarchy builder (mx_mg.hpp: 2x2x2 aggregates per component, one-pass Galerkin product with merge insertion)
    against scipy on the matrices it uploads: P has one unit entry per row with a non-zero diagonal, R = P^T, every
    coarse operator equals R A P on the same pattern (entries summed in another order: to rounding)."""
    import scipy.sparse as sp
    sim = mx.Sim.pillbox(24)
    shift = yee.default_shift([1, 1, 1])
    for op in ("shifted", "scaLapl"):
        mg = mx.Multigrid(ctx, sim, op, shift, min_coarse=200)
        nl = len(mg.levels)
        assert nl >= 3
        for l in range(nl - 1):
            A, P, R = (mg.level_matrix(l, w).to_scipy() for w in "APR")
            Ac = mg.level_matrix(l + 1, "A").to_scipy()
            assert (A.shape[0], A.nnz) == tuple(mg.levels[l]) and (Ac.shape[0], Ac.nnz) == tuple(mg.levels[l + 1])
            active = A.diagonal() != 0.0
            assert np.array_equal(np.diff(P.indptr), active.astype(np.int64)) and np.all(P.data == 1.0)
            assert (R - P.T).nnz == 0
            want = (R @ A @ P).tocsr(); want.sort_indices()
            got = Ac.tocsr(); got.sort_indices()
            assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
            d = got - want
            scale = np.abs(want.data).max()
            assert d.nnz == 0 or np.abs(d.data).max() <= 1e-13 * scale, (op, l)
            assert all(np.all(np.diff(got.indices[got.indptr[i]:got.indptr[i + 1]]) > 0) for i in range(0, got.shape[0], 97))   # sorted, unique
        with pytest.raises(mx.MxError):
            mg.level_matrix(nl, "A")


def test_multigrid_cg_loop_switches_do_not_change_the_iterates(ctx):
    """The round-3 rearrangements of the multigrid-CG loop (mxg_solve.hip: convergence test at once near the tolerance,
    x += a p in the direction kernel, the next cycle's first smoother step written by the update kernel) move work
    between kernels without changing one operation: with each of them switched off in a child process the solution is
    the same bit for bit and so is the iteration count.  The fused r.z (another order of the partial sums) agrees to
    rounding."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, hashlib; sys.path.insert(0, %r); import numpy as np; import maxwell_amd as mx\n"
            "from oracle import yee\n"
            "ctx = mx.Ctx(0); sim = mx.Sim.pillbox(40); shift = yee.default_shift([1, 1, 1])\n"
            "mg = mx.Multigrid(ctx, sim, 'shifted', shift, smoother='chebyshev'); A = mg.fine_matrix()\n"
            "rng = np.random.default_rng(11); b = sim.op('mRhs') @ rng.uniform(-1, 1, A.nrows)\n"
            "B = mx.MultiVec.from_numpy(ctx, b); X = mx.MultiVec(ctx, A.nrows, 1)\n"
            "it, conv, res = A.solve(B, X, method='cg', tol=1e-9 * np.linalg.norm(b), precond='mg', mg=mg)\n"
            "x = X.to_numpy()[:, 0]\n"
            "print('RESULT ' + json.dumps(dict(it=it, conv=bool(conv), res=res, sha=hashlib.sha256(x.tobytes()).hexdigest(), nrm=float(np.linalg.norm(x)), x0=x[:64].tolist())))\n" % root)
    outs = {}
    for name, env in (("default", {}), ("near", {"MXG_CG_NEAR": "0"}), ("xmove", {"MXG_CG_XMOVE": "0"}),
                      ("first", {"MXG_MG_FUSED_FIRST": "0"}), ("dot", {"MXG_MG_FUSED_DOT": "0"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        outs[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    d = outs["default"]
    assert d["conv"] and 5 <= d["it"] <= 40
    for name in ("near", "xmove", "first"):
        assert outs[name]["it"] == d["it"] and outs[name]["sha"] == d["sha"], name
    assert abs(outs["dot"]["it"] - d["it"]) <= 1 and outs["dot"]["conv"]
    assert np.abs(np.array(outs["dot"]["x0"]) - np.array(d["x0"])).max() <= 1e-8 * d["nrm"]
