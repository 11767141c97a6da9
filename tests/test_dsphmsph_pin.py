"""CPU twin of tests/test_dsphmsph_pin_gpu.py: the ORACLE's restatement of the FIT inverse permittivity
(oracle/yee.py inv_eps, after src/MxYeeFitInvEps.cpp) against answers that do not come from this repository:

  1. the reference-held analytic table of example/dsphmsph.py:512-561 (tests/golden/dsphmsph_ktable.json) for
     three symmetry classes at 8^3 and 16^3 (oracle-assembled pencil, scipy shift-invert);
  2. a physics patch test: for a PLANAR interface of any tilt the piecewise-constant field
     E_out = F, E_in = Gamma F is an exact solution of the interface conditions, so the stencil must map the
     exact dual-face averages of D to the exact edge averages of E - to rounding.  This checks the tuple
     mechanics (which neighbours, which fractions weigh which rows, the 1/numTuples average) independently of
     how the reference's text was read;
  3. an axis-aligned dielectric slab in a PEC box against the closed-form dispersion relation of its TM_z(1,1)
     mode: second-order convergence (product assembler, bit-identical to the oracle: tests/test_assembly.py).
"""
import cmath

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.optimize import brentq

import maxwell_amd as mx
from dsph_modes import TABLE, class_modes, expected_k, oct_lower_bcs
from oracle import geometry as geo
from oracle import yee


def _physical_modes(L, M, D, keep, sigma, want):
    d = M.diagonal()
    vals, vecs = spla.eigs(L[keep][:, keep].tocsc(), k=2 * want + 4, M=sp.diags(d[keep]).tocsc(), sigma=sigma,
                           which="LM", tol=1e-12)
    out = []
    for lam, v in zip(vals, vecs.T):
        full = np.zeros(L.shape[0], dtype=complex); full[keep] = v
        if np.linalg.norm(D @ (M @ full)) / np.linalg.norm(full) < 1e-6:
            assert abs(lam.imag) < 1e-8 * abs(lam)
            out.append(lam.real)
    return np.sqrt(np.sort(out))[:want]


def test_class_rule_reproduces_recorded_octmodes_output():
    """The symmetry-class rule as restated in tests/dsph_modes.py + the table give the frequencies the
    reference's octModes(['pec']*3, 5) returned when SURVEY.md section 8c was written."""
    f = expected_k(("pec", "pec", "pec"), 6) * TABLE["lightspeed"] / (2.0 * np.pi)
    want = TABLE["octModes_pec_pec_pec_5_Hz"]
    for g, w in zip([f[0:2], f[2:3], f[3:6]], want):
        assert np.all(np.abs(g - w) / w < 5e-5), (g, w)
    # every (l, m, phase) lands in exactly one class; a class and its all-flipped partner swap TE and TM
    for l in range(1, 5):
        for m in range(l + 1):
            for ph in ("re", "im")[: 1 if m == 0 else 2]:
                te, tm = oct_lower_bcs("TE", l, m, ph), oct_lower_bcs("TM", l, m, ph)
                assert all(a != b for a, b in zip(te, tm))


@pytest.mark.parametrize("lbcs,want", [(("pmc", "pmc", "pec"), 4), (("pec", "pec", "pmc"), 4), (("pec", "pec", "pec"), 3)])
def test_oracle_spectrum_converges_to_the_analytic_table(lbcs, want):
    exact = expected_k(lbcs, want)
    errs = []
    for n in (8, 16):
        sim = yee.YeeSim(n, origin=(0, 0, 0), size=(0.5,) * 3, lower=lbcs, upper=("pec",) * 3,
                         pec=geo.Sphere(TABLE["b"], (0, 0, 0)),
                         dielectrics=[(geo.Sphere(TABLE["a"], (0, 0, 0)), TABLE["eps_in"] * np.eye(3), "sphere")])
        o = sim.wave_ops()
        L, M, D = o["vecLapl"], o["mRhs"], o["divB"]
        keep = (M.diagonal() > 0) & (np.asarray(abs(L).sum(axis=1)).ravel() > 0)
        k = _physical_modes(L, M, D, keep, 0.9 * exact[0] ** 2, want)
        errs.append(np.abs(k - exact) / exact)
    e8, e16 = errs
    assert np.all(e8 < 2.5e-2) and np.all(e16 < 6e-3), (lbcs, e8, e16)
    assert np.all(e16 < 0.62 * e8), (lbcs, e8, e16)          # between first and second order at these sizes


@pytest.mark.parametrize("normal", [(0.48, 0.6, 0.64), (1.0, 0.0, 0.0), (-0.3, 0.9, 0.1), (0.0, 0.7071, 0.7071)])
@pytest.mark.parametrize("eps", ["iso", "aniso"])
def test_stencil_is_exact_on_a_tilted_planar_interface(normal, eps):
    n = 8
    nrm = np.array(normal, dtype=float); nrm /= np.linalg.norm(nrm)
    hs = geo.HalfSpace((0.503, 0.52, 0.47), tuple(nrm))
    if eps == "iso":
        e = 10.0 * np.eye(3)
    else:
        e = np.array([[9.0, 0.7, -0.4], [0.7, 4.0, 1.1], [-0.4, 1.1, 6.5]])     # symmetric, off-diagonal
    sim = yee.YeeSim(n, origin=(0, 0, 0), size=(1, 1, 1), dielectrics=[(hs, e, "d0")], strict=False)
    A = sim.inv_eps()
    E, Dm = sim.E, sim.D
    F = np.array([0.3, -1.1, 0.7])
    g_in = np.eye(3) + np.outer(nrm, nrm) @ (np.eye(3) - e) / (nrm @ e @ nrm)      # src/MxYeeFitInvEps.cpp:345
    p_in = e @ g_in
    e_in, d_in = g_in @ F, p_in @ F
    # the model field really satisfies the interface conditions: D normal and E tangential continuous
    assert abs(nrm @ d_in - nrm @ F) < 1e-12 and np.linalg.norm(np.cross(nrm, e_in - F)) < 1e-12
    lf = np.empty(len(E.gids)); af = np.empty(len(E.gids))
    for c in range(3):
        s = E.comp == c
        lf[s] = sim.frac_at(E, c, E.cells[s], "d0")
        af[s] = sim.frac_at(Dm, c, E.cells[s], "d0")
    comp = E.comp
    d_avg = af * d_in[comp] + (1 - af) * F[comp]          # exact dual-face averages of D
    e_avg = lf * e_in[comp] + (1 - lf) * F[comp]          # exact edge averages of E
    got = A @ d_avg
    interior = np.all((E.cells >= 2) & (E.cells <= n - 3), axis=1)
    cut = ((lf != 0) & (lf != 1)) | ((af != 0) & (af != 1))
    assert (interior & cut).sum() >= 20
    assert np.abs(got - e_avg)[interior].max() < 1e-13


def _slab_tm11_root(z0, eps1):
    """TM_z(1,1) of a PEC unit box filled with eps1 below z0: (kz1/eps1) tan(kz1 z0) + kz2 tan(kz2 (1 - z0)) = 0,
    kz1^2 = eps1 k^2 - 2 pi^2, kz2^2 = k^2 - 2 pi^2 (D_z and d/dz E_z continuous, E_z' = 0 on the z walls)."""
    kt2 = 2.0 * np.pi ** 2

    def f(k):
        kz1, kz2 = cmath.sqrt(eps1 * k * k - kt2), cmath.sqrt(k * k - kt2)
        return ((kz1 / eps1) * cmath.tan(kz1 * z0) + kz2 * cmath.tan(kz2 * (1 - z0))).real

    ks = np.linspace(1.45, 3.0, 800)
    vs = np.array([f(k) for k in ks])
    for i in range(len(ks) - 1):
        if vs[i] * vs[i + 1] < 0 and max(abs(vs[i]), abs(vs[i + 1])) < 50:
            return brentq(f, ks[i], ks[i + 1], xtol=1e-14)
    raise AssertionError("no root")


def test_axis_aligned_slab_second_order_against_dispersion_relation():
    z0, eps1 = 0.375, 10.0
    exact = _slab_tm11_root(z0, eps1)
    assert abs(exact - 1.874966516519066) < 1e-9
    errs = []
    for n in (8, 16):
        sim = mx.Sim(n, origin=(0, 0, 0), size=(1.0,) * 3, lower=("pec",) * 3, upper=("pec",) * 3,
                     dielectrics=[(eps1, "halfspace(0,0,-1;0,0,%r)" % z0)])
        L, M, D = sim.op("vecLapl"), sim.op("mRhs"), sim.op("divB")
        keep = (sim.used_mask() != 0) & (M.diagonal() > 0)
        k = _physical_modes(L, M, D, keep, exact ** 2, 1)
        errs.append(abs(k[0] - exact))
        sim.close()
    order = np.log2(errs[0] / errs[1])
    assert errs[1] < 5e-3 and abs(order - 2.0) < 0.15, (errs, order)
