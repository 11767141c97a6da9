"""The reference's command line on the GPU: `maxwell --infile=deck.mx` (src/maxwell.cpp:41-120) and the
same decks through mxh_problem_solver, against scipy's shift-invert of the oracle-assembled pencil.

Decks are hand-written (tests/golden/*.mx headers).  The deck's own tolerances apply
(eigensolver tol 1e-6 by default), so the comparison tolerance here is looser than the 1e-8 of
tests/test_eigen_gpu.py, which drives the same solver with tight tolerances."""
import os
import re
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import maxwell_amd as mx
from oracle import mxdeck, yee

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "maxwell_amd", "lib", "maxwell")
C0 = 299792458.0


def _oracle_modes(deck, want, strict=True):
    """k^2 of the operator MxSolver iterates on, from the oracle's dense restatement of
    MxMagWaveOp::Apply on the oracle-assembled matrices (oracle/yee.py mag_wave_modes)."""
    o = deck.sim(strict=strict).wave_ops()
    return yee.mag_wave_modes(o, deck.params["eigensolver : shift"], want)


def _sparse_physical_modes(deck, want):
    """For decks too large for the dense restatement (D-M frac 0 only): divergence-free modes of the
    oracle-assembled pencil by sparse shift-invert, as tests/test_eigen_gpu.py does."""
    o = deck.sim().wave_ops()
    L, M, D = o["vecLapl"], o["mRhs"], o["divB"]
    d = M.diagonal(); keep = d > 0
    vals, vecs = spla.eigsh(L[keep][:, keep].tocsc(), k=4 * want + 6, M=sp.diags(d[keep]).tocsc(),
                            sigma=deck.params["eigensolver : shift"] - 1e-3, which="LM", tol=1e-13)
    out = []
    for lam, v in zip(vals, vecs.T):
        full = np.zeros(L.shape[0]); full[keep] = v
        if np.linalg.norm(D @ (M @ full)) / np.linalg.norm(full) < 1e-6:
            out.append(lam)
    return np.sort(out)[:want]


def _vacuum_modes(deck, want):
    """Periodic vacuum: the exact discrete spectrum sum_i (2/h_i sin(pi m_i/N_i))^2, two
    polarisations per non-zero wave vector, three constant fields at k^2 = 0 (in a degenerate
    eigenspace longitudinal and transverse fields mix, so a divergence filter cannot be used)."""
    n, size = np.array(deck.n), np.array(deck.size)
    h = size / n
    vals = [0.0, 0.0, 0.0]
    for m in np.ndindex(*n):
        if any(m):
            vals += [float(np.sum((2.0 / h * np.sin(np.pi * np.array(m) / n)) ** 2))] * 2
    return np.sort(vals)[:want]


def test_maxwell_cli_pillbox_deck(built, tmp_path):
    """`maxwell --infile=pillbox.mx`: prints the table of src/MxSolver.cpp:176-196; TM010 and a TE111 copy."""
    deck = mxdeck.load(os.path.join(GOLDEN, "pillbox.mx"))
    want = _sparse_physical_modes(deck, 3)   # TM010, TE111 x2 (the pair is degenerate: either copy is fine)
    eig = tmp_path / "eig.txt"
    r = subprocess.run([CLI, "--infile=" + os.path.join(GOLDEN, "pillbox.mx"), "--eigvals=" + str(eig)],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Running simulation: Simulation" in r.stdout
    assert "========= Eigenfrequencies (Hz) =========" in r.stdout
    rows = re.findall(r"^\s*(\d+):\s+(\S+)\s+(\S+)\s+(\S+)\s*$", r.stdout, flags=re.M)
    assert len(rows) == 2
    table = np.array([[float(x) for x in row[1:]] for row in rows])
    saved = np.loadtxt(str(eig), ndmin=2)
    assert np.allclose(saved[:, 1], table[:, 0], rtol=1e-8)          # table prints 9 significant digits
    k2 = np.sort((2 * np.pi * saved[:, 1] / C0) ** 2)
    assert abs(k2[0] - want[0]) / want[0] < 1e-5, (k2, want)
    assert abs(k2[1] - want[1]) / want[1] < 1e-5, (k2, want)
    assert np.all(saved[:, 2] == 0.0) and np.all(saved[:, 3] < 1e-4)
    tm010 = (2.404825557695773 / 0.4) ** 2
    assert abs(k2[0] - tm010) / tm010 < 6e-3                           # O(h^2) at resolution 18
    # eigenvector files of MxIO::save (src/MxSolver.cpp:222-224, src/MxIO.cpp:167-222): (N+1)^3 x 3 per vector
    files = sorted(os.listdir(str(tmp_path)))
    for field in ("YeeFitBField", "YeeFitEField"):
        for vec in (0, 1):
            stem = "mx%s_eigVecs_vec%02d" % (field, vec)
            hit = [f for f in files if f.startswith(stem)]
            assert len(hit) == 1, (stem, files)
            path = os.path.join(str(tmp_path), hit[0])
            if hit[0].endswith(".npy"):
                arr = np.load(path)
            else:
                import ctypes as C
                from maxwell_amd import _lib
                lib = _lib.load_mxhost()
                dims = (C.c_int64 * 4)()
                arr = np.empty(19 * 19 * 19 * 3)
                assert lib.mxh_field_read_h5(path.encode(), b"real", dims, arr.ctypes.data_as(_lib.c_dp), arr.size) == 0
                arr = arr.reshape(tuple(dims))
            assert arr.shape == (19, 19, 19, 3)
            assert np.abs(arr).max() > 0 and np.all(arr[0, 0, :, :] == 0.0)     # metal corner column: not in the map


def test_reference_writer_layout_deck(built, ctx, tmp_path):
    """tests/golden/pillbox_mxpy.mx is config 1 (example/pillbox.py at 32^3, nev 5, basis 100, GMRES(40) + ILU) laid
    out the way the reference's writer does it - hand-derived from src/mx.py:48-55 (attributes, then children),
    :468-511 (BC lists spelled `[ periodic, periodic, periodic]`, the always-present <Output> block), :577-648 (solver
    blocks), :694-716 (root <Maxwell>, `domain`, the default shift as Python prints it); the reference's Python is not
    run.  Through `maxwell --infile=` and through the C ABI; `ilu` is served by the multigrid:
    <= 60 inner iterations per Apply."""
    name = os.path.join(GOLDEN, "pillbox_mxpy.mx")
    deck = mxdeck.load(name)
    assert deck.params["eigensolver : shift"] == 1.9739208802178716 and deck.params["linear solver : prec type"] == "ilu"
    solver = mx.Problem(path=name).solver(ctx)
    assert solver.solve() == 5
    st = solver.stats()
    assert st["vec_iters"] <= 60 * st["applies"] and st["sca_iters"] <= 60 * st["applies"], st
    k2 = np.sort([solver.result(i)["k2"].real for i in range(5)])
    tm010 = (2.404825557695773 / 0.4) ** 2
    te111 = (1.8411837813406593 / 0.4) ** 2 + (np.pi / 0.8) ** 2
    assert abs(k2[0] - tm010) / tm010 < 2e-3 and abs(k2[1] - te111) / te111 < 2e-3 and abs(k2[2] - te111) / te111 < 2e-3
    r = subprocess.run([CLI, "--infile=" + name, "--no-fields"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "served by the aggregation multigrid" in r.stdout
    rows = re.findall(r"^\s*(\d+):\s+(\S+)\s+(\S+)\s+(\S+)\s*$", r.stdout, flags=re.M)
    assert len(rows) == 5
    f = np.sort([float(row[1]) for row in rows])
    assert np.allclose((2 * np.pi * f / C0) ** 2, k2, rtol=1e-4)


def test_maxwell_cli_errors(built, tmp_path):
    r = subprocess.run([CLI], capture_output=True, text=True, timeout=60)
    assert "Please specify an input file using --infile=your_file.mx" in r.stdout
    bad = tmp_path / "bad.mx"
    bad.write_text("<S>\n polarization = TE\n <PML>\n </PML>\n</S>\n")
    r = subprocess.run([CLI, "--infile=" + str(bad)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "PML object" in r.stdout


def test_transformed_cavity_deck_through_the_c_abi(ctx):
    """mxh_problem_parse -> mxh_problem_solver: the deck's parameters drive MxSolver; eigenvalues against
    the oracle's dense restatement of the iterated operator (transformed shapes, PEC/PMC/periodic walls).
    The deck's D-M frac is set to 0 here: dropping small faces breaks div curl = 0 on the cut cells, the
    iterated operator stops being self-adjoint and the (reference's) Hermitian Krylov-Schur does not
    converge on it - the D-M frac > 0 assembly itself is covered bit for bit in tests/test_mx_deck.py."""
    with open(os.path.join(GOLDEN, "tilted_cavity.mx")) as f:
        text = f.read().replace("D-M frac = 0.05", "D-M frac = 0.0")
    deck = mxdeck.Deck(text)
    solver = mx.Problem(text=text).solver(ctx)
    assert solver.solve() == 3
    k2 = np.sort([solver.result(i)["k2"].real for i in range(3)])
    want = _oracle_modes(deck, 3)
    tol = 50 * deck.params["eigensolver : tol"]
    assert np.abs(k2 - want).max() / np.abs(want).max() < tol, (k2, want)


def test_periodic_vacuum_deck_exact_spectrum(ctx):
    """Periodic vacuum deck with strict=False: the exact discrete spectrum.  (In strict mode the
    reference's periodic-wrap quirk, DESIGN.md §7, drops the wrap-around entries of curlE only, so
    curlE != curlB^T, the operator is not symmetric and the deck's CG is not applicable - the
    reference's default there is GMRES; `maxwell --fix-periodic-wrap` selects strict=False.)"""
    name = os.path.join(GOLDEN, "vacuum.mx")
    deck = mxdeck.load(name)
    solver = mx.Problem(path=name, strict=False).solver(ctx)
    assert solver.solve() == 5
    k2 = np.sort([solver.result(i)["k2"].real for i in range(5)])
    want = _vacuum_modes(deck, 5)
    assert np.abs(k2 - want).max() / want.max() < 50 * deck.params["eigensolver : tol"], (k2, want)


def test_dielectric_sphere_deck_non_hermitian(ctx):
    """Config 5 physics through the deck (tests/golden/dsphmsph.mx): FIT inverse permittivity makes the
    operator non-symmetric, MxSolver leaves the problem non-Hermitian (src/MxSolver.cpp:68-73) and the
    general Krylov-Schur path runs with GMRES + multigrid inner solves.  Checked against the oracle's dense
    restatement of the iterated operator and against the analytic wave numbers of example/dsphmsph.py:512-561."""
    name = os.path.join(GOLDEN, "dsphmsph.mx")
    deck = mxdeck.load(name)
    solver = mx.Problem(path=name).solver(ctx)
    assert solver.solve() == 3
    res = [solver.result(i) for i in range(3)]
    k2 = np.sort([r["k2"].real for r in res])
    want = _oracle_modes(deck, 3)
    assert np.abs(k2 - want).max() / np.abs(want).max() < 50 * deck.params["eigensolver : tol"], (k2, want)
    assert all(abs(r["k2"].imag) < 1e-9 for r in res)
    k = np.sqrt(k2)
    assert abs(k[0] - 2.7914257502896397024) / 2.7914 < 5e-3       # TM_11, O(h^2) at resolution 12
    assert abs(k[1] - 4.0593020005144315490) / 4.0593 < 1e-2       # TE_21


def test_magnetic_sphere_deck(ctx):
    """tests/golden/magsph.mx: anisotropic <Mu> sphere around an eps core in a PEC sphere - the FIT permeability
    operator in the chain (mu dmA divB ... , mRhs = mu dmA; src/MxMagWaveOp.cpp:156-239), non-Hermitian Krylov-Schur
    (src/MxSolver.cpp:57: hasMu -> not Hermitian).  Against the oracle's dense restatement of the iterated operator,
    with the operator chain formed on the device (default) and on the host."""
    name = os.path.join(GOLDEN, "magsph.mx")
    deck = mxdeck.load(name)
    want = _oracle_modes(deck, 3)
    solver = mx.Problem(path=name).solver(ctx)
    assert solver.solve() == 3
    res = [solver.result(i) for i in range(3)]
    k2 = np.sort([r["k2"].real for r in res])
    assert np.abs(k2 - want).max() / np.abs(want).max() < 50 * deck.params["eigensolver : tol"], (k2, want)
    assert all(abs(r["k2"].imag) < 1e-9 for r in res)
    # the empty cavity's lowest mode is k^2 = (2.7437 / 0.49)^2 = 31.4 (TM_11 of a PEC sphere): the magnetic
    # filling pulls it down, and the anisotropy splits the threefold degeneracy
    assert k2[0] < 0.6 * (2.743707269992265 / 0.49) ** 2 and k2[1] - k2[0] > 0.5 and k2[2] - k2[1] > 0.5


def test_lossy_magnetic_sphere_deck(ctx):
    """tests/golden/magsph.mx with `loss tangent = 0.02` on the <Mu> block: complex permeability, complex non-Hermitian
    Krylov-Schur in the Euclidean inner product; damped modes against the oracle's dense restatement."""
    text = open(os.path.join(GOLDEN, "magsph.mx")).read().replace(
        "mu off diag = [0.1, -0.2, 0.3]", "mu off diag = [0.1, -0.2, 0.3]\n    loss tangent = 0.02").replace(
        "resolution = [10, 10, 10]", "resolution = [8, 8, 8]")
    deck = mxdeck.Deck(text)
    assert deck.params["is complex"]
    solver = mx.Problem(text=text).solver(ctx)
    assert solver.solve() == 3
    got = np.array(sorted((solver.result(i)["k2"] for i in range(3)), key=lambda z: z.real))
    want = yee.mag_wave_modes(deck.sim().wave_ops(), deck.params["eigensolver : shift"], 3)
    assert np.iscomplexobj(want) and np.all(want.imag < 0)
    assert np.abs(got - want).max() / np.abs(want).max() < 50 * deck.params["eigensolver : tol"], (got, want)


def test_pml_decks_complex_frequencies(ctx):
    """<PML> through MxProblem -> complex MxSolver (src/MxProblem.cpp:596-597 sets "is complex").
    (a) a constant stretch over the whole box (tests/test_pml.py UNIFORM): the exact complex spectrum of the stretched
    Yee grid; (b) tests/golden/pmlbox.mx, a layer at one end of a PEC box with a dielectric sphere, against the
    oracle's dense restatement of the iterated operator: damped modes, Im k^2 < 0."""
    from test_pml import UNIFORM, stretched_box_spectrum
    text = UNIFORM.replace("DIR", "[1.0, 0.0, 0.0]")
    deck = mxdeck.Deck(text)
    prob = mx.Problem(text=text)
    assert prob.is_complex
    solver = prob.solver(ctx)
    assert solver.solve() == 4
    got = np.sort_complex(np.array([solver.result(i)["k2"] for i in range(4)]))
    want = np.sort_complex(stretched_box_spectrum(deck.n, deck.size, [1.0 + 0.3j, 1.0, 1.0], 1.0, 4))
    assert np.abs(got - want).max() / np.abs(want).max() < 50 * deck.params["eigensolver : tol"], (got, want)

    name = os.path.join(GOLDEN, "pmlbox.mx")
    deck = mxdeck.load(name)
    solver = mx.Problem(path=name).solver(ctx)
    assert solver.solve() == 3
    got = np.array(sorted((solver.result(i)["k2"] for i in range(3)), key=lambda z: z.real))
    want = yee.mag_wave_modes(deck.sim().wave_ops(), deck.params["eigensolver : shift"], 3)
    assert np.abs(got - want).max() / np.abs(want).max() < 50 * deck.params["eigensolver : tol"], (got, want)
    assert got[0].imag < -0.5 and got[1].imag < -0.5


def test_complex_decks_bloch_phase_shifts(ctx):
    """Complex arithmetic end to end (src/MxProblem.cpp:160-175: MxSolver<DIM, MxComplex> when "is complex"):
    complex assembler -> complex CSR / multivectors -> complex GMRES -> complex Krylov-Schur.
    (a) periodic vacuum with Bloch phases: the exact spectrum sum_i (2/h_i sin((pi m_i + phi_i/2)/N_i))^2;
    (b) the dielectric-fibre deck in the manner of example/cylinder.py against the oracle's dense restatement.
    Both with strict=False: the reference's periodic-wrap quirk makes a periodic
    operator non-self-adjoint and its spectrum complex."""
    text = open(os.path.join(GOLDEN, "vacuum.mx")).read().replace("phase shifts = [0.0, 0.0, 0.0]", "phase shifts = [0.7, 0.0, -0.4]")
    text = text.replace('<LinearSolver type="cg">', '<LinearSolver type="gmres">').replace('<Preconditioner type="amg"/>', '<Preconditioner type="ilu"/>')
    text = text.replace("tol = 1.0e-8\n    shift", "tol = 1.0e-8\n    block size = 2\n    shift")   # every level is a polarisation pair
    deck = mxdeck.Deck(text)
    assert deck.params["eigensolver : block size"] == 2
    prob = mx.Problem(text=text, strict=False)
    assert prob.is_complex
    solver = prob.solver(ctx)
    assert solver.solve() == 5
    res = [solver.result(i) for i in range(5)]
    k2 = np.sort([r["k2"].real for r in res])
    n, h, ph = np.array(deck.n), np.array(deck.size) / np.array(deck.n), np.array(deck.phase)
    exact = np.sort([float(np.sum((2.0 / h * np.sin((np.pi * np.array(m) + ph / 2) / n)) ** 2)) for m in np.ndindex(*n)] * 2)[:5]
    assert np.abs(k2 - exact).max() / exact.max() < 50 * deck.params["eigensolver : tol"], (k2, exact)
    assert max(abs(r["k2"].imag) for r in res) < 1e-6 * exact.max()

    name = os.path.join(GOLDEN, "fiber.mx")
    deck = mxdeck.load(name)
    solver = mx.Problem(path=name, strict=False).solver(ctx)
    assert solver.solve() == 4
    k2 = np.sort([solver.result(i)["k2"].real for i in range(4)])
    want = yee.mag_wave_modes(deck.sim(strict=False).wave_ops(), deck.params["eigensolver : shift"], 4)
    assert np.isrealobj(want)
    assert np.abs(k2 - want).max() / np.abs(want).max() < 50 * deck.params["eigensolver : tol"], (k2, want)


def test_lossy_dielectric_sphere_complex_frequencies(ctx):
    """Config 5's complex variant: loss tangent 1e-3 on the eps = 10 sphere - complex k^2 (damped modes) from the
    complex non-Hermitian Krylov-Schur against the oracle's dense restatement."""
    text = open(os.path.join(GOLDEN, "dsphmsph.mx")).read().replace(
        "eps off diag = [0.0, 0.0, 0.0]", "eps off diag = [0.0, 0.0, 0.0]\n    loss tangent = 0.001").replace(
        "resolution = [12, 12, 12]", "resolution = [10, 10, 10]").replace("type = amg", "type = ilu")
    deck = mxdeck.Deck(text)
    solver = mx.Problem(text=text).solver(ctx)
    assert solver.solve() == 3
    got = np.array(sorted((solver.result(i)["k2"] for i in range(3)), key=lambda z: z.real))
    want = yee.mag_wave_modes(deck.sim().wave_ops(), deck.params["eigensolver : shift"], 3)
    assert np.iscomplexobj(want) and np.all(want.imag < 0)
    assert np.abs(got - want).max() / np.abs(want).max() < 50 * deck.params["eigensolver : tol"], (got, want)
    q = got.real / np.abs(got.imag)     # k^2 ~ omega^2 (1 - i/Q): Q = 1 / (loss tangent x electric filling factor) >= 1000
    assert np.all(q > 1000) and np.all(q < 3000)


def test_crab_cell_deck_composite_shapes(ctx):
    """tests/golden/crabcell.mx (geometry after example/crabcav.py:42-66): Cone, Torus, Ellipsoid, ShapeSubtract /
    Union / Mirror / Repeat through MxProblem -> MxSolver; the three-cell TM010-like passband against the sparse
    shift-invert of the oracle-assembled pencil."""
    name = os.path.join(GOLDEN, "crabcell.mx")
    deck = mxdeck.load(name)
    solver = mx.Problem(path=name).solver(ctx)
    assert solver.solve() == 3
    k2 = np.sort([solver.result(i)["k2"].real for i in range(3)])
    want = _sparse_physical_modes(deck, 3)
    assert np.abs(k2 - want).max() / np.abs(want).max() < 50 * deck.params["eigensolver : tol"], (k2, want)
    assert 60.0 < k2[0] < k2[2] < 75.0      # (2.405 / 0.36)^2 = 44.6 for the bare equator radius, raised by the cone walls


def test_pillbox_with_tubes_deck_phase_advance(ctx):
    """tests/golden/pillwtubes.mx (geometry after example/pillWTubes.py:91-104): ShapeUnion of a pillbox and a
    corrugated tube (inverted tori), 2 pi / 3 Bloch phase advance per cell -> complex solver.  strict=False as for
    every periodic deck."""
    name = os.path.join(GOLDEN, "pillwtubes.mx")
    deck = mxdeck.load(name)
    prob = mx.Problem(path=name, strict=False)
    assert prob.is_complex
    solver = prob.solver(ctx)
    assert solver.solve() == 2
    res = [solver.result(i) for i in range(2)]
    k2 = np.sort([r["k2"].real for r in res])
    want = yee.mag_wave_modes(deck.sim(strict=False).wave_ops(), deck.params["eigensolver : shift"], 2)
    assert np.abs(k2 - want).max() / np.abs(want).max() < 50 * deck.params["eigensolver : tol"], (k2, want)
    assert max(abs(r["k2"].imag) for r in res) < 1e-6 * np.abs(want).max()


def test_multigrid_of_the_real_part_preconditions_complex_solves(ctx):
    """`type = amg` on complex decks: the hierarchy is built from the real part of the operator and its V-cycle is
    applied to the real and the imaginary part of a residual (mxg_mg_apply) - same eigenvalues as with diagonal
    scaling, far fewer GMRES iterations.  Lossy sphere (complex FIT inverse permittivity) and Bloch-phase vacuum."""
    lossy = open(os.path.join(GOLDEN, "dsphmsph.mx")).read().replace(
        "eps off diag = [0.0, 0.0, 0.0]", "eps off diag = [0.0, 0.0, 0.0]\n    loss tangent = 0.001")
    bloch = open(os.path.join(GOLDEN, "fiber.mx")).read()
    for text, strict, want in ((lossy, True, 3), (bloch, False, 4)):
        assert ("type = amg" in text) != ("type = ilu" in text)
        res = {}
        for prec in ("amg", "jacobi", "ilu"):       # `ilu` is served by the multigrid (include/mx/MxMagWaveOp.h)
            t = text.replace("type = amg", "type = " + prec).replace("type = ilu", "type = " + prec)
            solver = mx.Problem(text=t, strict=strict).solver(ctx)
            assert solver.solve() == want
            res[prec] = (np.array(sorted((solver.result(i)["k2"] for i in range(want)), key=lambda z: (round(z.real, 6), z.imag))),
                         solver.stats())
        tol = 50 * float(mx.Problem(text=text, strict=strict).param("eigensolver : tol"))
        assert np.abs(res["amg"][0] - res["jacobi"][0]).max() / np.abs(res["jacobi"][0]).max() < tol
        assert np.abs(res["ilu"][0] - res["jacobi"][0]).max() / np.abs(res["jacobi"][0]).max() < tol
        assert res["amg"][1]["vec_iters"] < 0.5 * res["jacobi"][1]["vec_iters"], (res["amg"][1], res["jacobi"][1])
        assert res["ilu"][1]["vec_iters"] < 0.5 * res["jacobi"][1]["vec_iters"], (res["ilu"][1], res["jacobi"][1])
