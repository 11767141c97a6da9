"""BASELINE config 2 at its full size (pillbox 256^3: 20.6 M rows, 146 M non-zeros) on the GPU:
bit parity with the oracle's serial row loop on the whole matrix, the size-independent properties of the
operators (linearity, symmetry, div curl = 0), CSR vs compressed-stream kernel, and the eigenpairs against
the analytic cavity modes at the O(h^2) discretisation error."""
import numpy as np
import pytest

import maxwell_amd as mx
from oracle import cref

pytestmark = pytest.mark.gpu
N = 256
SHIFT = 0.05 * (2.0 * np.pi) ** 2


@pytest.fixture(scope="module")
def sim256(built):
    sim = mx.Sim.pillbox(N)
    yield sim
    sim.close()


def test_full_size_spmv_bit_exact_and_properties(ctx, sim256):
    sim = sim256
    A = sim.to_device(ctx, "vecLapl", shift=SHIFT)
    assert A.nrows == 20608881 and A.nnz == 145733441
    host = sim.op("vecLapl", shift=SHIFT)                       # scipy copy of the same arrays
    rng = np.random.default_rng(256)
    x = rng.uniform(-1, 1, A.ncols); y = rng.uniform(-1, 1, A.ncols)
    X = mx.MultiVec.from_numpy(ctx, x); Y = mx.MultiVec.from_numpy(ctx, y)
    AX = mx.MultiVec(ctx, A.nrows, 1); AY = mx.MultiVec(ctx, A.nrows, 1)
    A.apply(X, AX); A.apply(Y, AY)
    ax = AX.to_numpy()[:, 0]
    assert np.array_equal(ax, cref.spmv(host, x))               # every one of the 20.6 M row sums, bit for bit
    # symmetry of the pencil: y.(A x) = x.(A y)
    ay = AY.to_numpy()[:, 0]
    assert abs(y @ ax - x @ ay) <= 1e-11 * (np.linalg.norm(y) * np.linalg.norm(ax))
    # linearity: A(2x - 3y) = 2 Ax - 3 Ay up to rounding
    Z = mx.MultiVec.from_numpy(ctx, 2.0 * x - 3.0 * y); AZ = mx.MultiVec(ctx, A.nrows, 1)
    A.apply(Z, AZ)
    az = AZ.to_numpy()[:, 0]
    assert np.abs(az - (2.0 * ax - 3.0 * ay)).max() <= 1e-12 * np.abs(az).max()
    # the compressed-stream encoding of the same matrix gives the same bits
    nb, nf, nbytes = A.compress()
    assert nf < nb and nbytes < A.nnz * 5
    AC = mx.MultiVec(ctx, A.nrows, 1)
    A.apply(X, AC)
    assert np.array_equal(AC.to_numpy()[:, 0], ax)
    # div curl = 0: the discrete structure the projection of MxMagWaveOp::Apply relies on
    curlE = sim.to_device(ctx, "curlE"); divB = sim.to_device(ctx, "divB")
    e = mx.MultiVec(ctx, curlE.ncols, 1); e.random(7)
    b = mx.MultiVec(ctx, curlE.nrows, 1); p = mx.MultiVec(ctx, divB.nrows, 1)
    curlE.apply(e, b); divB.apply(b, p)
    assert p.norm2()[0] <= 1e-9 * b.norm2()[0]


def test_full_size_eigenpairs_against_the_analytic_cavity(ctx, sim256):
    params = mx.SolverParams(nev=3, basis=30, tol=1e-8, shift=SHIFT, lin_type="cg", lin_tol=1e-9, prec_type="amg")
    solver = mx.Solver(ctx, sim256, params)
    assert solver.solve() == 3
    res = [solver.result(i) for i in range(3)]
    k2 = np.sort([r["k2"].real for r in res])
    tm010 = (2.404825557695773 / 0.4) ** 2
    te111 = (1.8411837813406593 / 0.4) ** 2 + (np.pi / 0.8) ** 2
    assert abs(k2[0] - tm010) / tm010 < 5e-5                    # 36.14458 vs 36.14491: second-order error at h = 1/256
    assert abs(k2[1] - te111) / te111 < 5e-5 and abs(k2[2] - te111) / te111 < 5e-5
    assert abs(k2[1] - k2[2]) / te111 < 1e-8                    # the degenerate pair
    assert max(r["residual"] for r in res) < 1e-4 and max(r["divergence"] for r in res) < 1e-6   # as printed by checkEigensolution
    solver.close()


def test_config3_cylinder_full_size_bit_exact(ctx, built):
    """BASELINE config 3 (curved PEC cylinder, periodic z, 256^3: 25.6 M rows with irregular wall rows):
    the whole product against the oracle's serial row loop, CSR and compressed stream."""
    sim = mx.Sim(N, origin=(-0.5,) * 3, size=(1.0,) * 3, pec_shape="cylinder(0.4;0,0,1;0,0,0)")
    A = sim.to_device(ctx, "vecLapl", shift=SHIFT)
    host = sim.op("vecLapl", shift=SHIFT)
    assert int(np.diff(host.indptr).max()) > 7                   # wall rows are wider than the 7-point bulk
    x = np.random.default_rng(3).uniform(-1, 1, A.ncols)
    X = mx.MultiVec.from_numpy(ctx, x); Y = mx.MultiVec(ctx, A.nrows, 1)
    A.apply(X, Y)
    want = cref.spmv(host, x)
    assert np.array_equal(Y.to_numpy()[:, 0], want)
    A.compress()
    A.apply(X, Y)
    assert np.array_equal(Y.to_numpy()[:, 0], want)
    sim.close()


def test_config5_dielectric_block_of_8_bit_exact(ctx, built):
    """BASELINE config 5 operator (eps = 10 sphere in a PEC sphere, octant box, FIT inverse permittivity) at
    128^3 with a block of 8 right-hand sides: SpMM bit for bit against the serial row loop."""
    sim = mx.Sim(128, origin=(0.0,) * 3, size=(0.5,) * 3, lower=("pmc", "pmc", "pec"), upper=("pec",) * 3,
                 pec_shape="sphere(0.49;0,0,0)", dielectrics=[(10.0, "sphere(0.37;0,0,0)")])
    A = sim.to_device(ctx, "vecLapl", shift=0.5)
    host = sim.op("vecLapl", shift=0.5)
    assert abs(host - host.T).max() > 1e-3                        # the FIT inverse permittivity is not symmetric
    x = np.random.default_rng(5).uniform(-1, 1, (A.ncols, 8))
    X = mx.MultiVec.from_numpy(ctx, x); Y = mx.MultiVec(ctx, A.nrows, 8)
    A.apply(X, Y)
    got = Y.to_numpy()
    for j in range(8):
        assert np.array_equal(got[:, j], cref.spmv(host, x[:, j])), j
    A.compress()
    A.apply(X, Y)
    got = Y.to_numpy()
    for j in (0, 7):
        assert np.array_equal(got[:, j], cref.spmv(host, x[:, j])), j
    sim.close()
