"""Magnetic materials (<Mu>): the FIT permeability operator `mu` of
src/MxYeeFitMu.cpp:294-463 and the mu terms of the operator chain (src/MxMagWaveOp.cpp:156-239).

The reference holds no fixture for this operator (no example deck uses <Mu>), so the oracle is pinned on the
physics it must reproduce:
  1. a uniform relative permeability mu_r scales every k^2 by exactly 1/mu_r;
  2. the planar-interface patch test: for a plane of any tilt and any (symmetric) mu tensor the piecewise-constant
     field with continuous tangential H and normal B is mapped exactly (edge averages of H -> face averages of B);
and the product assembler (C++) is compared with the oracle bit for bit on the union pattern.
"""
import numpy as np
import pytest

import maxwell_amd as mx
from oracle import geometry as geo
from oracle import yee

ANISO = np.array([[3.0, 0.3, -0.2], [0.3, 2.0, 0.1], [-0.2, 0.1, 2.5]])


def _same(a, b):
    a = a.tocsr(); b = b.tocsr()
    d = a - b
    return a.shape == b.shape and (d.nnz == 0 or np.abs(d.data).max() == 0.0)


def test_uniform_permeability_scales_the_spectrum():
    kw = dict(origin=(0, 0, 0), size=(1.0, 0.8, 0.6), lower=("pec",) * 3, upper=("pec",) * 3)
    w0 = yee.mag_wave_modes(yee.YeeSim([6, 6, 6], **kw).wave_ops(), 1.0, 4)
    everywhere = geo.Sphere(10.0, [0, 0, 0])
    w1 = yee.mag_wave_modes(yee.YeeSim([6, 6, 6], mus=[(everywhere, 2.5 * np.eye(3), "m")], **kw).wave_ops(), 1.0, 4)
    assert np.allclose(w0 / w1, 2.5, rtol=1e-10)


def test_uniform_lossy_permeability_scales_the_spectrum_by_a_complex_factor():
    """mu_r (1 + i t) everywhere: every k^2 divided by exactly that number (damped modes, Q = 1 / t)."""
    kw = dict(origin=(0, 0, 0), size=(1.0, 0.8, 0.6), lower=("pec",) * 3, upper=("pec",) * 3)
    w0 = yee.mag_wave_modes(yee.YeeSim([6, 6, 6], **kw).wave_ops(), 1.0, 4)
    everywhere = geo.Sphere(10.0, [0, 0, 0])
    mu = 2.5 * (1.0 + 0.05j)
    w1 = yee.mag_wave_modes(yee.YeeSim([6, 6, 6], mus=[(everywhere, mu * np.eye(3), "m")], **kw).wave_ops(), 1.0, 4)
    assert np.iscomplexobj(w1) and np.allclose(np.sort_complex(w0 / w1), mu, rtol=1e-10)


@pytest.mark.parametrize("normal", [(0, 0, 1), (1, 1, 0), (1, 2, 3), (-2, 1, 0.5)])
@pytest.mark.parametrize("kind", ["iso", "aniso"])
def test_mu_stencil_is_exact_on_a_tilted_planar_interface(normal, kind):
    n = 8
    nrm = np.array(normal, dtype=float); nrm /= np.linalg.norm(nrm)
    hs = geo.HalfSpace((0.503, 0.52, 0.47), tuple(nrm))
    mu = 4.0 * np.eye(3) if kind == "iso" else ANISO
    sim = yee.YeeSim(n, origin=(0, 0, 0), size=(1, 1, 1), mus=[(hs, mu, "m0")], strict=False)
    A = sim.mu_op()
    B, H = sim.B, sim.H
    F = np.array([0.3, -1.1, 0.7])
    g_in = np.eye(3) + np.outer(nrm, nrm) @ (np.eye(3) - mu) / (nrm @ mu @ nrm)
    h_in, b_in = g_in @ F, mu @ g_in @ F
    assert abs(nrm @ b_in - nrm @ F) < 1e-12 and np.linalg.norm(np.cross(nrm, h_in - F)) < 1e-12
    lf = np.empty(len(B.gids)); af = np.empty(len(B.gids))
    for c in range(3):
        s = B.comp == c
        lf[s] = sim.frac_at(H, c, B.cells[s], "m0")
        af[s] = sim.frac_at(B, c, B.cells[s], "m0")
    comp = B.comp
    h_avg = lf * h_in[comp] + (1 - lf) * F[comp]          # exact dual-edge averages of H
    b_avg = af * b_in[comp] + (1 - af) * F[comp]          # exact face averages of B
    got = A @ h_avg
    interior = np.all((B.cells >= 2) & (B.cells <= n - 3), axis=1)
    cut = ((lf != 0) & (lf != 1)) | ((af != 0) & (af != 1))
    assert (interior & cut).sum() >= 20
    assert np.abs(got - b_avg)[interior].max() < 1e-13


def test_assembler_matches_oracle_with_magnetic_sphere(built):
    """C++ assembler == oracle for every operator of the chain: anisotropic mu sphere (off-diagonal terms -> the
    9-point tuple stencil on cut faces) + isotropic eps core inside a curved PEC wall."""
    kw = dict(origin=(0.0, 0.0, 0.0), size=(0.5, 0.5, 0.5), lower=("pmc", "pmc", "pec"), upper=("pec", "pec", "pec"))
    o = yee.YeeSim(8, pec=geo.Sphere(0.49, [0, 0, 0]), dielectrics=[(geo.Sphere(0.2, [0, 0, 0]), 4.0 * np.eye(3), "d0")],
                   mus=[(geo.Sphere(0.33, [0, 0, 0]), ANISO, "mu0")], **kw).wave_ops()
    sim = mx.Sim(8, pec_shape="sphere(0.49;0,0,0)", dielectrics=[(4.0, "sphere(0.2;0,0,0)")],
                 mus=[((3.0, 2.0, 2.5, 0.1, -0.2, 0.3), "sphere(0.33;0,0,0)")], **kw)
    assert "mu" in o and "invMuVolAve" in o
    for name, ref in o.items():
        assert _same(sim.op(name), ref), name
    assert np.abs(o["mu"] - sp_diag(o["mu"])).max() > 1e-3          # the stencil really has off-diagonal entries


def test_assembler_matches_oracle_without_pec(built):
    """No PEC: mRhs is the mu operator itself (src/MxMagWaveOp.cpp:237-238), periodic box."""
    o = yee.YeeSim(6, mus=[(geo.Sphere(0.3, [0.5, 0.5, 0.5]), np.diag([2.0, 3.0, 4.0]), "mu0")], strict=False,
                   origin=(0, 0, 0), size=(1, 1, 1)).wave_ops()
    sim = mx.Sim(6, mus=[((2.0, 3.0, 4.0), "sphere(0.3;0.5,0.5,0.5)")], strict=False, origin=(0, 0, 0), size=(1, 1, 1))
    for name, ref in o.items():
        assert _same(sim.op(name), ref), name
    assert _same(sim.op("mRhs"), o["mu"])


def sp_diag(a):
    import scipy.sparse as sp
    return sp.diags(a.diagonal())
