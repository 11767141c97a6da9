"""Where does the first-order term of the TM entries of the dielectric-sphere pin come from?  

tests/test_dsphmsph_pin_gpu.py pins the GPU eigen-solve on the reference-held analytic table; its TE entries converge at
second order, its TM entries (E normal to the curved dielectric surface) carry a small first-order term a h.  Two
experiments with exact physics separate the two curved surfaces of that problem:
  (i)  the EMPTY PEC sphere (no dielectric): closed-form spectrum, k a = zeros of j_l (TE) and of d/dx [x j_l(x)] (TM).
       Only the Dey-Mittra cut-cell treatment of the PEC sphere is in play.
  (ii) an epsilon-contrast sweep at fixed geometry (eps = 1.01, 2, 10 in the r = 0.37 sphere): the coefficient a of the
       lowest mode (TM, l = 1) from the three-resolution fit k(h) = k0 + a h + b h^2 - no analytic k needed.
The measured numbers are written to gpurun_out/tm_term.json (profiles/ keeps a copy) and DESIGN.md section 5 states the
conclusion; the assertions hold what was observed so that a regression of either surface treatment shows up."""
import json
import os

import numpy as np
import pytest
from scipy.optimize import brentq
from scipy.special import spherical_jn

import maxwell_amd as mx
from dsph_modes import oct_lower_bcs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = (32, 64, 128)
A_PEC = 0.49
LBCS = ("pmc", "pmc", "pec")
_report = {}


def _deck(n, nev, eps=None, pec_sphere=True):
    diel = "" if eps is None else """  <Dielectric>
    eps diag = [%r, %r, %r]
    eps off diag = [0.0, 0.0, 0.0]
    <Sphere>
      name = epsSphere
      radius = 0.37
    </Sphere>
  </Dielectric>
""" % (eps, eps, eps)
    pec = "" if not pec_sphere else """  <PEC>
    <Sphere>
      name = pecSphere
      radius = %r
    </Sphere>
  </PEC>
""" % A_PEC
    return """<Simulation>
  name = sphere
  dim = 3
  polarization = TE
  <Grid>
    resolution = [%d, %d, %d]
    origin = [0.0, 0.0, 0.0]
    dimensions = [0.5, 0.5, 0.5]
  </Grid>
  <BoundaryConditions>
    lower bcs = [%s, %s, %s]
    upper bcs = [pec, pec, pec]
  </BoundaryConditions>
%s%s  <Eigensolver>
    nev = %d
    basis = %d
    block size = 3
    shift = 0.5
    tol = 1e-09
    <LinearSolver>
      type = %s
      basis = 40
      tol = 1e-11
      <Preconditioner>
        type = amg
      </Preconditioner>
    </LinearSolver>
  </Eigensolver>
</Simulation>
""" % (n, n, n, LBCS[0], LBCS[1], LBCS[2], pec, diel, nev, max(30, 4 * nev), "gmres")


def _solve(ctx, n, nev, eps, pec_sphere=True):
    prob = mx.Problem(text=_deck(n, nev, eps, pec_sphere))
    solver = prob.solver(ctx)
    nconv = solver.solve()
    assert nconv >= nev, (n, eps, nconv)
    res = [solver.result(i) for i in range(nconv)]
    assert max(r["residual"] for r in res) < 1e-6
    k = np.sqrt(np.sort([r["k2"].real for r in res]))[:nev]
    solver.close(); prob.close()
    return k


def _fit(ks):
    h = np.array([0.5 / n for n in RES])
    V = np.stack([np.ones(3), h, h * h], axis=1)
    k0, a, b = np.linalg.solve(V, ks)
    order = float(np.log2((ks[0] - ks[1]) / (ks[1] - ks[2])))
    return float(k0), float(a), float(b), order


def _empty_sphere_modes(kmax):
    """[(k, pol, l, multiplicity in the class LBCS)] of the PEC sphere of radius A_PEC, closed form."""
    out = []
    for l in range(1, 7):
        te = lambda x: spherical_jn(l, x)
        tm = lambda x: spherical_jn(l, x) + x * spherical_jn(l, x, derivative=True)
        for pol, f in (("TE", te), ("TM", tm)):
            xs = np.linspace(0.5, kmax * A_PEC + 1.0, 4000)
            v = f(xs)
            for i in np.nonzero(v[:-1] * v[1:] < 0)[0]:
                k = brentq(f, xs[i], xs[i + 1], xtol=1e-14) / A_PEC
                mult = sum(1 for m in range(l + 1) for ph in ("re", "im")
                           if not (m == 0 and ph == "im") and list(LBCS) == oct_lower_bcs(pol, l, m, ph))
                if mult and k <= kmax:
                    out.append((k, pol, l, mult))
    return sorted(out)


def test_empty_pec_sphere_is_second_order_for_te_and_tm(ctx):
    """(i) Dey-Mittra alone: every mode of the empty PEC sphere, TM included, converges at second order to the closed
    form - the cut-cell PEC treatment carries no first-order term."""
    modes = []
    for k, pol, l, mult in _empty_sphere_modes(13.0):
        modes += [(k, pol, l)] * mult
    nev = 4
    modes = modes[:nev]
    ks = np.array([_solve(ctx, n, nev, None) for n in RES])
    rep = []
    for i, (kt, pol, l) in enumerate(modes):
        k0, a, b, order = _fit(ks[:, i])
        rich = ks[2, i] + (ks[2, i] - ks[1, i]) / 3.0
        rec = dict(mode="%s l=%d" % (pol, l), k_exact=kt, k=[float(v) for v in ks[:, i]], order=order, a=a, b=b,
                   richardson2_rel=float(abs(rich - kt) / kt), rel_err=[float(abs(v - kt) / kt) for v in ks[:, i]])
        rep.append(rec)
        assert rec["rel_err"][0] < 3e-3 and rec["rel_err"][2] < rec["rel_err"][1] < rec["rel_err"][0], rec
        assert abs(order - 2.0) <= 0.3, rec                       # TE and TM alike
        assert rec["richardson2_rel"] <= 2e-5, rec
    assert any(r["mode"].startswith("TM") for r in rep) and any(r["mode"].startswith("TE") for r in rep)
    _report["empty_pec_sphere"] = rep


def test_first_order_term_sits_where_the_inverse_permittivity_stencil_meets_the_pec_sphere(ctx):
    """(ii) the coefficient a of the lowest mode (TM, l = 1) for eps = 1.01, 2, 10 in the inner sphere does NOT follow the
    permittivity jump: it is the same -0.10 ... -0.12 for all three, although with eps = 1.01 the problem is physically the
    empty sphere of (i), whose a is below 0.01.  What the three runs share is the operator chain with the FIT inverse
    permittivity in it (curlE dmL invEps curlB instead of curlE dmL curlB): its 9-point stencil loses the legs that point at
    edges inside the metal (not in the E map) in the cells cut by the PEC sphere, where a TM mode has its normal E.
    (iii) the same dielectric sphere WITHOUT the PEC sphere (PEC box walls only: no cut cells): a vanishes - the curved
    dielectric interface itself is second order, for E normal to it too."""
    rep = []
    for eps in (1.01, 2.0, 10.0):
        ks = np.array([_solve(ctx, n, 1, eps)[0] for n in RES])
        k0, a, b, order = _fit(ks)
        rep.append(dict(eps=eps, k=[float(v) for v in ks], k0=k0, a=a, b=b, order=order))
    _report["eps_sweep_lowest_mode"] = rep
    box = []
    ks = np.array([_solve(ctx, n, 3, 10.0, pec_sphere=False) for n in RES])
    for i in range(3):
        k0, a, b, order = _fit(ks[:, i])
        box.append(dict(mode=i, k=[float(v) for v in ks[:, i]], k0=k0, a=a, b=b, order=order))
    _report["dielectric_sphere_in_pec_box_no_cut_cells"] = box
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(dict(resolutions=RES, **_report), open(os.path.join(out, "tm_term.json"), "w"), indent=1)
    a = [r["a"] for r in rep]
    assert all(-0.14 < v < -0.08 for v in a), rep                  # independent of the jump
    assert max(a) - min(a) < 0.03, rep
    assert all(abs(r["a"]) < 0.02 and abs(r["order"] - 2.0) < 0.3 for r in box), box
