"""Row (e) end to end: the eigen-solve on a decomposed grid - one rank per MxGridDomain box (src/MxGrid.cpp:171-193),
row-partitioned operators with ghost import (MxCrsMatrix, plan agreed like Epetra's FillComplete), all-reduced
inner products, rank-local multigrid blocks - against the single-rank solve and the oracle.

RCCL refuses two ranks on one device, so on the one-GPU test box the ranks are threads of an in-process world
(mxg_local_world, include/mxg.h): the same collectives, the same call sequence, staged through the host.  The
RCCL world runs the identical code above mxg_comm_* (maxwell_amd/csrc/device/mxg_comm.hip)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import maxwell_amd as mx
from maxwell_amd import distributed as mxd
from maxwell_amd.api import PILLBOX_SHAPE
from oracle import yee

pytestmark = pytest.mark.gpu
BOX = dict(origin=(-0.5,) * 3, size=(1.0,) * 3)


def _run_ranks(nranks, fn):
    world = mx.LocalWorld(nranks)
    try:
        return world.run(fn)
    finally:
        world.close()


@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_row_partitioned_product_matches_global(built, nranks, monkeypatch):
    """y = (vecLapl - shift dmA) x and the rectangular divB / gradPsi with real neighbour exchanges on 2x1x1, 2x2x1 and
    2x2x2 boxes (src/MxGrid.cpp:171-193; 2x2x2 is the decomposition of BASELINE config 4): every rank's owned rows
    equal the rows of the oracle's global product bit for bit - CSR kernel, compressed stream and the tile form with
    its ghost-reading group rows, one column and a block of three; norms are all-reduced."""
    monkeypatch.setenv("MXG_TILE_MIN_COVERAGE", "0")          # small boxes: take the tile form whatever its coverage
    n = [12, 10, 14]
    shift = 1.7
    from oracle import geometry as geo
    osim = yee.YeeSim(n, pec=geo.pillbox(), **BOX)
    o = osim.wave_ops()
    A = yee.shifted(o["vecLapl"], o["mRhs"], shift).tocsr()
    fields = {"bfield": osim.B.gids, "psifield": osim.P.gids}
    rng = np.random.default_rng(5)
    xB = rng.uniform(-1, 1, len(fields["bfield"]))
    xP = rng.uniform(-1, 1, len(fields["psifield"]))
    want = {"shifted": A @ xB, "divB": o["divB"] @ xB, "gradPsi": o["gradPsi"] @ xP}

    def rank(ctx):
        sim = mxd.BoxSim(n, ctx.rank, nranks, pec_shape=PILLBOX_SHAPE, **BOX)
        ops = mxd.DistOps(ctx, sim)
        out = {}
        for name, colfield, xg in (("shifted", "bfield", xB), ("divB", "bfield", xB), ("gradPsi", "psifield", xP)):
            M = ops.matrix(name, shift)
            x = mx.MultiVec(ctx, M.ncols_owned, 1)
            x.set(xg[np.searchsorted(fields[colfield], M.col_gids)].reshape(-1, 1))
            y = mx.MultiVec(ctx, M.nrows, 1)
            M.apply(x, y)
            out[name] = (M.row_gids.copy(), y.to_numpy()[:, 0].copy(), y.norm2()[0])
            x3 = mx.MultiVec(ctx, M.ncols_owned, 3); x3.random(11 + ctx.rank)
            y3 = mx.MultiVec(ctx, M.nrows, 3)
            M.apply(x3, y3)
            want3 = y3.to_numpy().copy()
            # the same products from the re-encoded forms, bit-identical to the CSR kernel.  Compressed stream (3 B per
            # entry): interior blocks first, boundary blocks read the ghost buffer after the exchange.  Tile form: tiles
            # and ghost-free group rows run while the halo is in flight, ghost-reading group rows behind the halo event.
            for form in (1, 3):
                os.environ["MXG_PAT"] = "0" if form == 1 else "1"      # (every rank thread writes the same value)
                nb, nfall, _ = M.compress()
                assert nb > 0 and M.product_form()[0] == form, (name, form, M.product_form())
                y2 = mx.MultiVec(ctx, M.nrows, 1); y2.fill(float("nan"))
                M.apply(x, y2)
                assert np.array_equal(y2.to_numpy(), y.to_numpy()), (name, form)
                y3.fill(float("nan"))
                M.apply(x3, y3)
                assert np.array_equal(y3.to_numpy(), want3), (name, form, "block of 3")
                ctx.barrier()                                            # nobody re-encodes while another rank still reads the switch
        ops.close()
        return out

    try:
        res = _run_ranks(nranks, rank)
    finally:
        os.environ.pop("MXG_PAT", None)
    for name, rowfield in (("shifted", "bfield"), ("divB", "psifield"), ("gradPsi", "bfield")):
        seen = 0
        for r in range(nranks):
            gids, y, nrm = res[r][name]
            pos = np.searchsorted(fields[rowfield], gids)
            assert np.array_equal(fields[rowfield][pos], gids)
            assert np.array_equal(y, want[name][pos]), (name, r)
            assert abs(nrm - np.linalg.norm(want[name])) < 1e-12 * np.linalg.norm(want[name])
            seen += len(gids)
        assert seen == len(want[name])


@pytest.mark.parametrize("nranks,lin_type,prec", [(2, "cg", "amg"), (4, "cg", "amg"), (8, "cg", "amg"), (2, "gmres", "jacobi")])
def test_decomposed_eigensolve_matches_single_rank(ctx, oracle_pillbox16, nranks, lin_type, prec, monkeypatch):
    """MxSolver on 2 / 4 / 8 boxes (8 = 2x2x2, the decomposition of BASELINE config 4): same eigenvalues as on one rank
    and as the oracle's shift-invert of the global pencil, to north_star's 1e-8; residuals and divergences (all-reduced
    norms) as the reference prints them.  The operators and multigrid levels run the tile form with ghost-reading group
    rows (forced: the boxes of a 16^3 grid are too small for its coverage threshold)."""
    monkeypatch.setenv("MXG_TILE_MIN_COVERAGE", "0")
    from tests.test_eigen_gpu import _oracle_physical_modes, REL_TOL
    _, o = oracle_pillbox16
    shift = yee.default_shift([1, 1, 1])
    want = _oracle_physical_modes(o, shift, 4)
    kw = dict(nev=4, basis=30, tol=1e-10, shift=shift, lin_type=lin_type, lin_basis=40, lin_tol=1e-11, prec_type=prec,
              block_size=1)

    def rank(c):
        sim = mxd.BoxSim(16, c.rank, nranks, pec_shape=PILLBOX_SHAPE, **BOX)
        solver = mx.Solver(c, sim, mx.SolverParams(**kw))
        nconv = solver.solve()
        res = [solver.result(i) for i in range(nconv)]
        st = solver.stats()
        solver.close()
        return res, st

    out = _run_ranks(nranks, rank)
    ref = out[0][0]
    assert len(ref) == 4
    k2 = np.sort([r["k2"].real for r in ref])
    assert np.abs(k2 - want).max() / np.abs(want).max() < REL_TOL, (k2, want)
    for res, st in out:                       # every rank holds the same answer
        assert [r["k2"] for r in res] == [r["k2"] for r in ref]
        assert [r["residual"] for r in res] == [r["residual"] for r in ref]
        assert st["applies"] == out[0][1]["applies"] and st["vec_iters"] == out[0][1]["vec_iters"]
    for r in ref:
        assert r["residual"] < 1e-7 and r["divergence"] < 1e-6
    # single rank on the same device
    solver = mx.Solver(ctx, mx.Sim.pillbox(16), mx.SolverParams(**kw))
    assert solver.solve() == 4
    k1 = np.sort([solver.result(i)["k2"].real for i in range(4)])
    assert np.abs(k2 - k1).max() / np.abs(k1).max() < REL_TOL
    if prec == "amg":
        # the multigrid levels are row-partitioned like the operator (include/mx/MxDistMultigrid.h): the preconditioner
        # is the single-rank one up to summation order, so the inner iteration counts do not grow with the boxes
        # (per operator application: on another summation order the eigensolver may stop an application earlier or later)
        st1, stn = solver.stats(), out[0][1]
        assert abs(stn["applies"] - st1["applies"]) <= 4
        for key in ("vec_iters", "sca_iters"):
            per1, pern = st1[key] / st1["applies"], stn[key] / stn["applies"]
            assert abs(pern - per1) <= 0.05 * per1 + 0.1, (key, per1, pern)



def test_decomposed_dielectric_sphere_non_hermitian(ctx):
    """Config 5 on 2 and 4 boxes: the FIT inverse permittivity makes the operator non-symmetric, so the general
    Krylov-Schur path with GMRES + multigrid inner solves runs on row-partitioned operators; eigenvalues against the
    single-rank solve of the same deck-like setup."""
    kw = dict(origin=(0.0,) * 3, size=(0.5,) * 3, lower=("pmc", "pmc", "pec"), upper=("pec",) * 3,
              pec_shape="sphere(0.49;0,0,0)", dielectrics=[(10.0, "sphere(0.37;0,0,0)")])
    par = dict(nev=3, basis=30, tol=1e-10, shift=0.5, lin_type="gmres", lin_basis=60, lin_tol=1e-11, prec_type="amg", block_size=1)
    solver = mx.Solver(ctx, mx.Sim(16, **kw), mx.SolverParams(**par))
    assert solver.solve() == 3
    k1 = np.sort([solver.result(i)["k2"].real for i in range(3)])
    for nranks in (2, 4):
        def rank(c):
            s = mx.Solver(c, mxd.BoxSim(16, c.rank, nranks, **kw), mx.SolverParams(**par))
            nconv = s.solve()
            res = [s.result(i) for i in range(nconv)]
            s.close()
            return res
        out = _run_ranks(nranks, rank)
        assert len(out[0]) == 3
        k2 = np.sort([r["k2"].real for r in out[0]])
        assert np.abs(k2 - k1).max() / np.abs(k1).max() < 1e-8, (nranks, k2, k1)
        assert max(abs(r["k2"].imag) for r in out[0]) < 1e-9
        for res in out[1:]:
            assert [r["k2"] for r in res] == [r["k2"] for r in out[0]]


def test_deck_on_a_decomposed_grid(ctx, tmp_path):
    """A `.mx` deck on 2 boxes through both entries: mxh_problem_solver (Problem.solver on rank contexts) and the C++
    MxProblem<3> as the `maxwell` command line drives it per process (tests/cpp/test_problem_ranks.cpp: rank
    threads of an in-process world); same eigenvalues as the single-rank run of the deck."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    deck = os.path.join(root, "tests", "golden", "pillbox.mx")
    one = mx.Problem(path=deck).solver(ctx)
    n1 = one.solve()
    k1 = np.sort([one.result(i)["k2"].real for i in range(n1)])
    shift = float(mx.Problem(path=deck).param("eigensolver : shift"))

    def rank(c):
        s = mx.Problem(path=deck).solver(c)
        n = s.solve()
        return [s.result(i)["k2"].real for i in range(n)]

    out = _run_ranks(2, rank)
    assert out[0] == out[1] and len(out[0]) == n1
    tol = 50 * float(mx.Problem(path=deck).param("eigensolver : tol"))
    assert np.abs(np.sort(out[0]) - k1).max() / np.abs(k1).max() < tol

    exe = str(tmp_path / "test_problem_ranks")
    lib = os.path.join(root, "maxwell_amd", "lib")
    subprocess.run(["g++", "-O1", "-fopenmp", "-std=c++17", "-I" + os.path.join(root, "include"),
                    "-I" + os.path.join(root, "include", "mx"), "-I" + os.path.join(root, "maxwell_amd", "csrc", "host"),
                    os.path.join(root, "tests", "cpp", "test_problem_ranks.cpp"), "-o", exe, "-L" + lib, "-lmxhost", "-lmxg",
                    "-lpthread", "-Wl,-rpath," + lib], check=True)
    r = subprocess.run([exe, deck, "2", "save"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    theta = [float(line.split()[2]) for line in r.stdout.splitlines() if line.startswith("THETA")]
    k2 = np.sort([1.0 / t + shift for t in theta])
    assert len(k2) == n1 and np.abs(k2 - k1).max() / np.abs(k1).max() < tol, (k2, k1)
    # rank 0 wrote whole-lattice eigenvector files (MxIO::save gathers the boxes): same layout as the single-rank
    # file, the lowest mode equal to the single-rank eigenvector up to sign

    def read_field(path):
        if path.endswith(".npy"):
            return np.load(path)
        import ctypes as C
        from maxwell_amd import _lib
        lib = _lib.load_mxhost()
        dims = (C.c_int64 * 4)()
        n = mx.Problem(path=deck).grid()["n"]
        arr = np.empty((n[0] + 1) * (n[1] + 1) * (n[2] + 1) * 3)
        assert lib.mxh_field_read_h5(path.encode(), b"real", dims, arr.ctypes.data_as(_lib.c_dp), arr.size) == 0
        return arr.reshape(tuple(dims))

    v1 = one.eigvec(int(np.argmin([one.result(i)["k2"].real for i in range(n1)])), mx.Problem(path=deck).sim().map("bfield").size)
    gids = mx.Problem(path=deck).sim().map("bfield")
    files = sorted(f for f in os.listdir(str(tmp_path)) if f.startswith("mxYeeFitBField_eigVecs_vec"))
    assert len(files) == n1
    order = np.argsort([1.0 / t + shift for t in theta])
    arr = read_field(os.path.join(str(tmp_path), files[int(order[0])]))
    flat = arr.reshape(-1)
    assert np.count_nonzero(flat) <= len(gids) and np.abs(np.delete(flat, gids)).max() == 0.0
    got = flat[gids]
    got, v1 = got / np.linalg.norm(got), v1 / np.linalg.norm(v1)
    assert min(np.abs(got - v1).max(), np.abs(got + v1).max()) < 1e-4


def test_maxwell_cli_with_rccl_id_file(built, tmp_path):
    """`maxwell` under a launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK + --id-file): the unique-id hand-over
    and the RCCL communicator, here with a world of one process (RCCL refuses two ranks on the one test GPU)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    idf = tmp_path / "rccl.id"
    r = subprocess.run([os.path.join(root, "maxwell_amd", "lib", "maxwell"), "--infile=" + os.path.join(root, "tests", "golden", "pillbox.mx"),
                        "--no-fields", "--id-file=" + str(idf)], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "========= Eigenfrequencies (Hz) =========" in r.stdout and not idf.exists()


def test_complex_decks_on_a_decomposed_grid(ctx):
    """Complex arithmetic on 2 boxes: (a) the Bloch-phase fibre deck (strict=False) - ghost
    columns across the periodic seam carry e^{+-i phi}; (b) the lossy dielectric sphere (complex FIT inverse
    permittivity, complex eigenvalues).  Same eigenvalues as the single-rank complex solve."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fiber = open(os.path.join(root, "tests", "golden", "fiber.mx")).read()
    lossy = open(os.path.join(root, "tests", "golden", "dsphmsph.mx")).read().replace(
        "eps off diag = [0.0, 0.0, 0.0]", "eps off diag = [0.0, 0.0, 0.0]\n    loss tangent = 0.001").replace(
        "resolution = [12, 12, 12]", "resolution = [10, 10, 10]")     # `type = amg`: multigrid of the real part, row-partitioned
    for text, strict, want in ((fiber, False, 4), (lossy, True, 3)):
        one = mx.Problem(text=text, strict=strict)
        assert one.is_complex
        s1 = one.solver(ctx)
        assert s1.solve() == want
        k1 = np.array(sorted((s1.result(i)["k2"] for i in range(want)), key=lambda z: (round(z.real, 6), z.imag)))

        def rank(c):
            s = mx.Problem(text=text, strict=strict).solver(c)
            n = s.solve()
            return [s.result(i)["k2"] for i in range(n)]

        out = _run_ranks(2, rank)
        assert out[0] == out[1] and len(out[0]) == want
        k2 = np.array(sorted(out[0], key=lambda z: (round(z.real, 6), z.imag)))
        tol = 50 * float(one.param("eigensolver : tol"))
        assert np.abs(k2 - k1).max() / np.abs(k1).max() < tol, (k2, k1)


@pytest.mark.parametrize("n,nranks", [([14, 11, 9], 3), ([20, 12, 10], 6)])
def test_odd_grids_and_rank_counts(ctx, n, nranks):
    """Boxes whose edges are not multiples of the aggregate size (the row-partitioned coarsening stops at once and the
    whole hierarchy is replicated), rank counts that are not powers of two, a rank that owns mostly metal."""
    shift = yee.default_shift([1, 1, 1])
    kw = dict(nev=3, basis=30, tol=1e-10, shift=shift, lin_type="cg", lin_tol=1e-11, prec_type="amg", block_size=1)
    solver = mx.Solver(ctx, mx.Sim(n, pec_shape=PILLBOX_SHAPE, **BOX), mx.SolverParams(**kw))
    assert solver.solve() == 3
    k1 = np.sort([solver.result(i)["k2"].real for i in range(3)])

    def rank(c):
        s = mx.Solver(c, mxd.BoxSim(n, c.rank, nranks, pec_shape=PILLBOX_SHAPE, **BOX), mx.SolverParams(**kw))
        nconv = s.solve()
        res = [s.result(i)["k2"].real for i in range(nconv)]
        s.close()
        return res

    out = _run_ranks(nranks, rank)
    assert all(o == out[0] for o in out) and len(out[0]) == 3
    assert np.abs(np.sort(out[0]) - k1).max() / np.abs(k1).max() < 1e-8


def test_ranks_that_own_nothing(ctx):
    """A small cavity in one corner of the grid: three of the four MxGridDomain boxes lie in the metal and own no
    unknown at all (empty maps, empty matrices, empty multigrid levels) - they still take part in every collective."""
    kw = dict(origin=(-0.5,) * 3, size=(1.0,) * 3, lower=("pec",) * 3, upper=("pec",) * 3,
              pec_shape="sphere(0.21;-0.25,-0.25,-0.25)")
    par = dict(nev=2, basis=20, tol=1e-9, shift=30.0, lin_type="cg", lin_tol=1e-10, prec_type="amg")
    s1 = mx.Solver(ctx, mx.Sim(16, **kw), mx.SolverParams(**par))
    assert s1.solve() == 2
    k1 = np.sort([s1.result(i)["k2"].real for i in range(2)])

    def rank(c):
        sim = mxd.BoxSim(16, c.rank, 4, **kw)
        rows = len(sim.dist_op("shifted", 30.0)["rowgid"])
        s = mx.Solver(c, sim, mx.SolverParams(**par))
        n = s.solve()
        res = [s.result(i)["k2"].real for i in range(n)]
        s.close()
        return rows, res

    out = _run_ranks(4, rank)
    assert sorted(o[0] for o in out)[:3] == [0, 0, 0] and max(o[0] for o in out) > 0
    assert all(o[1] == out[0][1] for o in out)
    assert np.abs(np.sort(out[0][1]) - k1).max() / k1.max() < 1e-7


@pytest.mark.parametrize("lin_type,block", [("cg", 2), ("bicgstab", 1)])
def test_decomposed_block_and_bicgstab(ctx, oracle_pillbox16, lin_type, block):
    """Block size 2 (two-column products with ghost import, block orthogonalisation with all-reduced Gram matrices)
    and the BiCGStab inner solver on 2 boxes."""
    from tests.test_eigen_gpu import _oracle_physical_modes, REL_TOL
    _, o = oracle_pillbox16
    shift = yee.default_shift([1, 1, 1])
    want = _oracle_physical_modes(o, shift, 4)
    kw = dict(nev=4, basis=20, tol=1e-10, shift=shift, lin_type=lin_type, lin_tol=1e-11, prec_type="amg", block_size=block)

    def rank(c):
        s = mx.Solver(c, mxd.BoxSim(16, c.rank, 2, pec_shape=PILLBOX_SHAPE, **BOX), mx.SolverParams(**kw))
        n = s.solve()
        res = [s.result(i)["k2"].real for i in range(n)]
        s.close()
        return res

    out = _run_ranks(2, rank)
    assert out[0] == out[1] and len(out[0]) >= 4
    k2 = np.sort(out[0])[:4]
    assert np.abs(k2 - want).max() / np.abs(want).max() < REL_TOL, (k2, want)


def test_setup_collectives_over_rccl_one_rank(built):
    """The setup-time collectives of include/mxg.h on a real RCCL communicator (a world of one rank - RCCL refuses two
    ranks on the one test GPU): all-gather of an int64, of variable-length byte strings, all-reduce of host doubles."""
    import ctypes as C
    from maxwell_amd import _lib
    c = mx.Ctx(0, 0, 1, mx.Ctx.unique_id())
    try:
        lib = c.lib
        out = (C.c_int64 * 1)()
        assert lib.mxg_comm_allgather_i64(c.h, 41, out) == 0 and out[0] == 41
        data = np.arange(1000, dtype=np.int64) * 3 + 1
        counts = (C.c_int64 * 1)(data.nbytes)
        recv = np.zeros_like(data)
        assert lib.mxg_comm_allgatherv(c.h, data.ctypes.data_as(C.c_void_p), data.nbytes, counts, recv.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(recv, data)
        v = np.array([1.5, -2.25, 1e300])
        assert lib.mxg_comm_allreduce_host(c.h, v.ctypes.data_as(_lib.c_dp), 3) == 0
        assert np.array_equal(v, [1.5, -2.25, 1e300])
        c.barrier()
    finally:
        c.close()


def test_partitioned_multigrid_class_on_one_rccl_rank(built, tmp_path):
    """MXG_DIST_MG=force: the row-partitioned multigrid class (coarsening by lattice ids, plan agreement per level,
    gathered levels with their all-gather per V-cycle) on ONE rank over a real RCCL communicator.  With one rank it must
    reproduce the single-rank hierarchy: same inner iteration counts and eigenvalues as the default path."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json; sys.path.insert(0, %r)\n"
        "import numpy as np, maxwell_amd as mx\n"
        "c = mx.Ctx(0, 0, 1, mx.Ctx.unique_id())\n"
        "s = mx.Solver(c, mx.Sim.pillbox(32), mx.SolverParams(nev=3, basis=20, tol=1e-9, shift=0.05 * (2 * np.pi) ** 2, "
        "lin_type='cg', lin_tol=1e-10, prec_type='amg'))\n"
        "n = s.solve(); st = s.stats()\n"
        "print('RESULT ' + json.dumps(dict(k2=[s.result(i)['k2'].real for i in range(n)], vec=st['vec_iters'], sca=st['sca_iters'])))\n"
        % root)
    out = {}
    for mode in ("default", "force"):
        env = dict(os.environ)
        env.pop("MXG_DIST_MG", None)
        if mode == "force":
            env["MXG_DIST_MG"] = "force"
            env["MXG_DIST_MG_REPLICATE"] = "20000"      # two partitioned levels above the gathered ones at 32^3
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-2000:]
        out[mode] = json.loads(line[0][7:])
    assert len(out["force"]["k2"]) == 3
    assert np.allclose(out["force"]["k2"], out["default"]["k2"], rtol=1e-9)
    assert abs(out["force"]["vec"] - out["default"]["vec"]) <= 0.03 * out["default"]["vec"] + 2
    assert abs(out["force"]["sca"] - out["default"]["sca"]) <= 0.03 * out["default"]["sca"] + 2
