"""Row f2: the operator chain of MxMagWaveOp::initMatrices formed ON THE DEVICE (mxg_csr_multiply / add / purge,
maxwell_amd/csrc/device/mxg_spalg.hip) against the chain the host assembler forms (include/mx/mx_sparse.hpp, which
tests/test_assembly.py pins bit for bit to the oracle): same pattern, same bits, for vacuum / PEC and dielectric
problems, and the shifted operator of setShift (src/MxMagWaveOp.cpp:248-255)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import maxwell_amd as mx

pytestmark = pytest.mark.gpu


def _same(dev, host):
    d = dev.to_scipy()
    h = host.tocsr()
    assert d.shape == h.shape and d.nnz == h.nnz, (d.shape, h.shape, d.nnz, h.nnz)
    assert np.array_equal(d.indptr, h.indptr) and np.array_equal(d.indices, h.indices)
    assert np.array_equal(d.data, h.data)


def _same_values(dev, ref):
    """Against the oracle's scipy operators, which drop the explicit zeros the product keeps (as Epetra does) in the
    intermediate products: every entry equal on the union pattern."""
    d = dev.to_scipy(); r = ref.tocsr()
    d.sort_indices(); r.sort_indices()
    diff = d - r
    assert d.shape == r.shape and (diff.nnz == 0 or np.abs(diff.data).max() == 0.0)


def _chain(ops, names):
    """massiveCrsMultiply (src/MxUtil.cpp:318-371): the first name is applied first, each step a full product."""
    res = ops[names[0]]
    for nm in names[1:]:
        res = ops[nm].multiply(res)
    return res


def _oracle_ops(case):
    """The same problems in the ORACLE (oracle/yee.py): the device chain is compared with it directly, not only through
    the product's host chain."""
    from oracle import geometry as geo, yee
    if case == "pillbox":
        return yee.pillbox_sim(16).wave_ops()
    return yee.YeeSim(12, origin=(0.0,) * 3, size=(0.5,) * 3, lower=("pmc", "pmc", "pec"), upper=("pec",) * 3,
                      pec=geo.Sphere(0.49, [0, 0, 0]),
                      dielectrics=[(geo.Sphere(0.37, [0, 0, 0]), 10.0 * np.eye(3), "dielectric0")]).wave_ops()


@pytest.mark.parametrize("case", ["pillbox", "dielectric"])
def test_device_operator_chain_bit_identical_to_host(ctx, case):
    if case == "pillbox":
        sim = mx.Sim.pillbox(16)
        diel = False
    else:
        sim = mx.Sim(12, origin=(0, 0, 0), size=(0.5,) * 3, lower=("pmc", "pmc", "pec"), upper=("pec",) * 3,
                     pec_shape="sphere(0.49;0,0,0)", dielectrics=[(10.0, "sphere(0.37;0,0,0)")])
        diel = True
    oracle = _oracle_ops(case)
    single = ["curlE", "curlB", "divB", "gradPsi", "dmA", "dmL", "dmVInv"] + (["invEps", "invEpsVolAve"] if diel else [])
    ops = {nm: sim.to_device(ctx, nm) for nm in single}
    curlcurl = _chain(ops, ["curlB"] + (["invEps"] if diel else []) + ["dmL", "curlE"])
    _same(curlcurl, sim.op("curlCurl"))
    _same_values(curlcurl, oracle["curlCurl"])
    graddiv = _chain(ops, ["dmA", "divB", "dmVInv"] + (["invEpsVolAve"] if diel else []) + ["gradPsi", "dmA"])
    _same(graddiv, sim.op("gradDiv"))
    _same_values(graddiv, oracle["gradDiv"])
    veclapl = curlcurl.add(1.0, graddiv, -1.0).purge(1e-12)
    _same(veclapl, sim.op("vecLapl"))
    _same(veclapl, oracle["vecLapl"])
    shift = 1.9739208802178716
    shifted = veclapl.add(1.0, ops["dmA"], -shift)          # mRhs = dmA with a PEC (src/MxMagWaveOp.cpp:233-236)
    _same(shifted, sim.op("vecLapl", shift=shift))
    from oracle import yee as _yee
    _same_values(shifted, _yee.shifted(oracle["vecLapl"], oracle["mRhs"], shift))
    scalapl = _chain(ops, ["gradPsi", "dmA", "divB"])
    neg = scalapl.add(-1.0, scalapl, 0.0)                   # -(divB dmA gradPsi), :208-220
    host = sim.op("scaLapl")
    d = neg.to_scipy()
    assert np.array_equal(d.indices, host.indices) and np.array_equal(d.data, host.data)
    # the device-born matrix is a full mxg_csr: products with it match the host-born one bit for bit
    x = np.random.default_rng(3).uniform(-1, 1, shifted.ncols)
    X = mx.MultiVec.from_numpy(ctx, x)
    Y1 = mx.MultiVec(ctx, shifted.nrows, 1); Y2 = mx.MultiVec(ctx, shifted.nrows, 1)
    shifted.apply(X, Y1)
    sim.to_device(ctx, "vecLapl", shift=shift).apply(X, Y2)
    assert np.array_equal(Y1.to_numpy(), Y2.to_numpy())
    shifted.compress()
    shifted.apply(X, Y1)
    assert np.array_equal(Y1.to_numpy(), Y2.to_numpy())


def test_device_algebra_general_matrices(ctx, rng):
    """Rectangular, ragged, empty rows, cancellation to exact zeros: against the same accumulation order in numpy."""
    n, m, k = 700, 500, 600
    A = sp.random(n, m, density=0.01, random_state=1, format="csr"); A.sort_indices()
    B = sp.random(m, k, density=0.02, random_state=2, format="csr"); B.sort_indices()
    dA, dB = mx.CsrMatrix.from_scipy(ctx, A), mx.CsrMatrix.from_scipy(ctx, B)
    C = dA.multiply(dB).to_scipy()
    # reference: Gustavson in meeting order
    want_rows = []
    for i in range(n):
        acc = {}
        order = []
        for p in range(A.indptr[i], A.indptr[i + 1]):
            kk, a = A.indices[p], A.data[p]
            for q in range(B.indptr[kk], B.indptr[kk + 1]):
                j = B.indices[q]
                prod = a * B.data[q]
                if j in acc:
                    acc[j] = acc[j] + prod
                else:
                    acc[j] = prod; order.append(j)
        want_rows.append(sorted(acc.items()))
    for i in range(n):
        cols = C.indices[C.indptr[i]:C.indptr[i + 1]]
        vals = C.data[C.indptr[i]:C.indptr[i + 1]]
        assert list(cols) == [j for j, _ in want_rows[i]]
        assert list(vals) == [v for _, v in want_rows[i]]
    E = sp.random(n, m, density=0.01, random_state=5, format="csr"); E.sort_indices()
    S = dA.add(2.0, mx.CsrMatrix.from_scipy(ctx, E), -0.5).to_scipy()
    want = (2.0 * A - 0.5 * E).tocsr(); want.sort_indices()
    U = (abs(A) + abs(E)).tocsr(); U.sort_indices()                                  # union pattern (no cancellation possible)
    assert np.array_equal(S.indptr, U.indptr) and np.array_equal(S.indices, U.indices)   # kept entry for entry, explicit zeros too
    assert abs(S - want).max() < 1e-15
    Z = dA.add(1.0, dA, -1.0)                                                         # exact cancellation
    assert Z.nnz == A.nnz and Z.purge().nnz == 0
    P = mx.CsrMatrix.from_scipy(ctx, A).purge(0.5).to_scipy()
    keep = np.abs(A.data) > 0.5
    assert P.nnz == int(keep.sum()) and np.array_equal(P.data, A.data[keep])


def test_eigen_solve_with_the_chain_formed_on_the_device(ctx):
    """MXH_DEVICE_CHAIN=1: the sim assembles the single operators only, MxMagWaveOp::initMatrices forms curlCurl,
    gradDiv, vecLapl, scaLapl and the shifted operator through MxCrsMatrix::matrixMatrixMultiply / Add / purgeZeros on
    the GPU, the multigrid setup reads the finished matrices back.  Same matrices bit for bit, so the whole solve
    reproduces the host-chain solve exactly.  (The switch is read once per process: the device-chain run is a child.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json; sys.path.insert(0, %r); import numpy as np; import maxwell_amd as mx\n"
            "ctx = mx.Ctx(0); sim = mx.Sim.pillbox(16)\n"
            "p = mx.SolverParams(nev=4, basis=25, tol=1e-9, shift=1.9739208802178716, lin_type='cg', lin_tol=1e-10, prec_type='amg')\n"
            "s = mx.Solver(ctx, sim, p); n = s.solve(); st = s.stats()\n"
            "print('RESULT ' + json.dumps(dict(k2=[s.result(i)['k2'].real for i in range(n)], vec=st['vec_iters'], sca=st['sca_iters'])))\n" % root)
    outs = {}
    for mode in ("0", "1"):
        env = dict(os.environ, MXH_DEVICE_CHAIN=mode)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
        outs[mode] = json.loads(line[7:])
    assert len(outs["0"]["k2"]) == 4
    assert outs["0"] == outs["1"], outs


@pytest.mark.parametrize("nranks", [2, 8])
def test_device_chain_on_the_window_of_a_decomposed_grid(ctx, nranks):
    """Row f2 on a decomposed grid (BASELINE config 4): every rank forms the operator chain on its box + halo WINDOW on
    its GPU (mxh_sim_device_chain, what mxh_solver_create does) - bit-identical to the host chain on the same window,
    and the owned rows of vecLapl - shift mRhs equal the rows of the ORACLE's global operator entry for entry."""
    from maxwell_amd import distributed as mxd
    from maxwell_amd.api import PILLBOX_SHAPE
    from oracle import geometry as geo, yee
    n, shift = [12, 10, 14], 1.7
    box = dict(origin=(-0.5,) * 3, size=(1.0,) * 3)
    osim = yee.YeeSim(n, pec=geo.pillbox(), **box)
    o = osim.wave_ops()
    A = yee.shifted(o["vecLapl"], o["mRhs"], shift).tocsr(); A.sort_indices()
    gids = osim.B.gids
    seen = 0
    for rank in range(nranks):
        dev = mxd.BoxSim(n, rank, nranks, pec_shape=PILLBOX_SHAPE, **box)
        dev.form_chain_on_device(ctx)
        os.environ["MXH_DEVICE_CHAIN"] = "0"
        try:
            host = mxd.BoxSim(n, rank, nranks, pec_shape=PILLBOX_SHAPE, **box)
        finally:
            os.environ.pop("MXH_DEVICE_CHAIN")
        for name in ("curlCurl", "gradDiv", "vecLapl", "scaLapl", "mRhs"):
            d, h = dev.op(name), host.op(name)
            assert d.shape == h.shape and np.array_equal(d.indptr, h.indptr) and np.array_equal(d.indices, h.indices), (rank, name)
            assert np.array_equal(d.data, h.data), (rank, name)
        op = dev.dist_op("shifted", shift)
        pos = np.searchsorted(gids, op["rowgid"])
        assert np.array_equal(gids[pos], op["rowgid"])
        for i, r in enumerate(pos):
            q0, q1 = op["rowptr"][i], op["rowptr"][i + 1]
            got = dict(zip(op["colgid"][q0:q1].tolist(), op["vals"][q0:q1].tolist()))
            want = dict(zip(gids[A.indices[A.indptr[r]:A.indptr[r + 1]]].tolist(), A.data[A.indptr[r]:A.indptr[r + 1]].tolist()))
            for c in set(got) | set(want):                 # (an explicit zero of the product is no entry of the oracle)
                assert got.get(c, 0.0) == want.get(c, 0.0), (rank, i, c)
        seen += len(pos)
        dev.close(); host.close()
    assert seen == A.shape[0]


def _golden(name):
    root = os.path.dirname(os.path.abspath(__file__))
    return open(os.path.join(root, "golden", name)).read()


def _complex_decks():
    lossy = _golden("dsphmsph.mx").replace("eps off diag = [0.0, 0.0, 0.0]", "eps off diag = [0.0, 0.0, 0.0]\n    loss tangent = 0.001")
    lossy = lossy.replace("resolution = [12, 12, 12]", "resolution = [10, 10, 10]")
    bloch = _golden("fiber.mx").replace("resolution = [2, 12, 12]", "resolution = [4, 10, 10]")
    lossymu = _golden("magsph.mx").replace("mu off diag = [0.1, -0.2, 0.3]", "mu off diag = [0.1, -0.2, 0.3]\n    loss tangent = 0.02")
    lossymu = lossymu.replace("resolution = [10, 10, 10]", "resolution = [8, 8, 8]")
    return {"lossy dielectric": lossy, "bloch phases": bloch, "lossy mu": lossymu}


@pytest.mark.parametrize("case", ["lossy dielectric", "bloch phases", "lossy mu"])
def test_device_operator_chain_complex_scalars(ctx, case):
    """Row f2 for complex scalars (loss tangents, Bloch phases): the chain of MxMagWaveOp::initMatrices formed by the
    complex instances of mxg_csr_multiply / add / purge - bit-identical to the host chain of YeeAssembler<complex>
    (include/mx/mx_sparse.hpp on std::complex) and equal, entry for entry, to the complex ORACLE's operators."""
    from oracle import mxdeck
    text = _complex_decks()[case]
    prob = mx.Problem(text=text)
    assert prob.is_complex
    oracle = mxdeck.Deck(text).sim().wave_ops()
    have = lambda nm: nm in oracle
    ops = {}
    for nm in ("curlE", "curlB", "divB", "gradPsi", "dmA", "dmL", "dmVInv", "invEps", "invEpsVolAve", "mu", "invMuVolAve"):
        if have(nm):
            m = prob.op(nm).tocsr()
            assert np.iscomplexobj(m.data)
            ops[nm] = mx.CsrMatrix.from_scipy(ctx, m)
            assert ops[nm].is_complex
    opt = lambda names: [nm for nm in names if nm in ops]
    curlcurl = _chain(ops, opt(["curlB", "invEps", "dmL", "curlE"]))
    assert curlcurl.is_complex
    _same(curlcurl, prob.op("curlCurl"))
    _same_values(curlcurl, oracle["curlCurl"])
    graddiv = _chain(ops, opt(["mu", "dmA", "divB", "dmVInv", "invMuVolAve", "invEpsVolAve"]) + (["invMuVolAve"] if "invMuVolAve" in ops else []) +
                     opt(["gradPsi", "mu", "dmA"]))
    _same(graddiv, prob.op("gradDiv"))
    _same_values(graddiv, oracle["gradDiv"])
    veclapl = curlcurl.add(1.0, graddiv, -1.0).purge(1e-12)
    _same(veclapl, prob.op("vecLapl"))
    _same(veclapl, oracle["vecLapl"])
    mrhs = prob.op("mRhs").tocsr()
    shift = 1.25
    shifted = veclapl.add(1.0, mx.CsrMatrix.from_scipy(ctx, mrhs), -shift)
    want = (prob.op("vecLapl") - shift * mrhs).tocsr(); want.sort_indices()
    got = shifted.to_scipy(); got.sort_indices()
    d = got - want
    assert d.nnz == 0 or np.abs(d.data).max() == 0.0
    # a complex scalar of matrixMatrixAdd (mxg_csr_add_z): (a + i b) A + B against numpy on the union pattern
    z = veclapl.add(0.5 + 0.25j, mx.CsrMatrix.from_scipy(ctx, mrhs), -1.0j).to_scipy()
    V = prob.op("vecLapl").tocsr()
    zw = ((0.5 + 0.25j) * V + (-1.0j) * mrhs).tocsr()
    dz = z - zw
    assert dz.nnz == 0 or np.abs(dz.data).max() <= 4e-16 * np.abs(zw.data).max()
    # the device-born complex matrix multiplies like the host-born one, bit for bit
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, veclapl.ncols) + 1j * rng.uniform(-1, 1, veclapl.ncols)
    X = mx.MultiVec.from_numpy(ctx, x)
    Y1 = mx.MultiVec(ctx, veclapl.nrows, 1, is_complex=True); Y2 = mx.MultiVec(ctx, veclapl.nrows, 1, is_complex=True)
    veclapl.apply(X, Y1)
    mx.CsrMatrix.from_scipy(ctx, V).apply(X, Y2)
    assert np.array_equal(Y1.to_numpy(), Y2.to_numpy())


def test_complex_eigen_solve_with_the_chain_formed_on_the_device(ctx):
    """A lossy-dielectric deck solved twice in child processes, operator chain on the host (MXH_DEVICE_CHAIN=0) and on
    the device (default): the matrices are the same bit for bit, so are the eigenvalues and the iteration counts."""
    import json, subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = _complex_decks()["lossy dielectric"]
    with tempfile.NamedTemporaryFile("w", suffix=".mx", delete=False) as f:
        f.write(text)
        deck = f.name
    code = ("import sys, json; sys.path.insert(0, %r); import numpy as np; import maxwell_amd as mx\n"
            "ctx = mx.Ctx(0); prob = mx.Problem(path=%r); s = prob.solver(ctx); n = s.solve(); st = s.stats()\n"
            "print('RESULT ' + json.dumps(dict(k2=[[s.result(i)['k2'].real, s.result(i)['k2'].imag] for i in range(n)], vec=st['vec_iters'], sca=st['sca_iters'])))\n"
            % (root, deck))
    outs = {}
    try:
        for mode in ("0", "1"):
            env = dict(os.environ, MXH_DEVICE_CHAIN=mode)
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
            assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
            outs[mode] = json.loads(line[7:])
    finally:
        os.unlink(deck)
    assert len(outs["0"]["k2"]) >= 1
    assert outs["0"] == outs["1"], outs
