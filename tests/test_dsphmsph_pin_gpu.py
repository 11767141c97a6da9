"""The parity pin on reference-held numbers: the reference's own analytic table of the
dielectric sphere in a PEC sphere (example/dsphmsph.py:512-561, kept as data in tests/golden/dsphmsph_ktable.json)
against the GPU eigen-solve of the example/run.py problem, for three of the eight symmetry classes of
example/dsphmsph.py:203-210 at three resolutions.

Per table entry (a mode of a class; degenerate members each count):
  * the three-resolution extrapolation  k(h) = k0 + a h + b h^2  (h = 0.5/N, N = 32, 64, 128) must give the
    table value:  |k0 - k_table| / k_table <= 2e-5 for TM entries (<= 4e-5 as a sanity bound for TE entries,
    whose sharp statement is the next item: fitting an absent first-order term only amplifies solver noise);
  * TE entries (E tangential to both spheres) converge at second order: observed order
    log2((k32 - k64) / (k64 - k128)) = 2 +- 0.3, and the plain second-order Richardson value
    k128 + (k128 - k64)/3 is within 2e-5 of the table;
  * TM entries (E normal to the dielectric surface) carry a small FIRST-order term besides the second-order one
    (a ~ 0.1 on TM_01): the scheme's own behaviour on a curved interface - the same stencil reproduces a
    piecewise-constant field across a tilted planar interface exactly and is second order on an axis-aligned
    slab (tests/test_dsphmsph_pin.py), and the term survives exact (arc instead of chord) area fractions - see
    DESIGN.md section 5.  Their observed order is asserted to lie in [0.9, 2.3] and recorded, not forced to 2.
Also checked: the three frequencies the reference's octModes(['pec']*3, 5) returns.
"""
import json
import os

import numpy as np
import pytest

import maxwell_amd as mx
from dsph_modes import TABLE, class_modes, deck_text, expected_k

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = (32, 64, 128)
CLASSES = {                      # lower walls -> number of eigenpairs asked for (whole degenerate groups)
    ("pmc", "pmc", "pec"): 6,    # TM_01, TE_02, TM_11, TM_03 x2, TE_04 (one of its two members)
    ("pec", "pec", "pec"): 6,    # TE_02 x2, TM_03, TE_04 x3  (= octModes(['pec']*3, 5))
    ("pec", "pec", "pmc"): 5,    # TE_01, TM_02, TE_03 x2, TE_11
}
RICHARDSON_TOL = 2e-5
_cache = {}


def _solve(ctx, lbcs, nev, n):
    prob = mx.Problem(text=deck_text(n, lbcs, nev, basis=max(30, 4 * nev)))
    solver = prob.solver(ctx)
    nconv = solver.solve()
    assert nconv >= nev, (lbcs, n, nconv)
    res = [solver.result(i) for i in range(nconv)]
    assert max(abs(r["k2"].imag) for r in res) < 1e-8
    assert max(r["residual"] for r in res) < 1e-6
    k = np.sqrt(np.sort([r["k2"].real for r in res]))[:nev]
    st = solver.stats()
    solver.close(); prob.close()
    return k, st


def _class_data(ctx, lbcs):
    if lbcs not in _cache:
        nev = CLASSES[lbcs]
        ks, secs = [], []
        for n in RES:
            k, st = _solve(ctx, lbcs, nev, n)
            ks.append(k); secs.append(st["eigensolve_seconds"])
        _cache[lbcs] = (np.array(ks), secs)
    return _cache[lbcs]


def _entries(lbcs, nev):
    """(table k, pol, n, l) per computed eigenvalue index."""
    out = []
    for k, pol, n, l, mult in class_modes(lbcs, 8.0):
        out += [(k, pol, n, l)] * mult
    return out[:nev]


@pytest.mark.parametrize("lbcs", list(CLASSES))
def test_analytic_table_by_symmetry_class(ctx, lbcs):
    nev = CLASSES[lbcs]
    ks, secs = _class_data(ctx, lbcs)
    h = np.array([0.5 / n for n in RES])
    V = np.stack([np.ones(3), h, h * h], axis=1)
    report = []
    for i, (kt, pol, nr, l) in enumerate(_entries(lbcs, nev)):
        k32, k64, k128 = ks[:, i]
        k0, a, b = np.linalg.solve(V, ks[:, i])
        order = float(np.log2((k32 - k64) / (k64 - k128)))
        rich2 = k128 + (k128 - k64) / 3.0
        rec = dict(cls="_".join(lbcs), mode="%s n=%d l=%d" % (pol, nr, l), k_table=kt, k=[float(v) for v in ks[:, i]],
                   rel_err=[float(abs(v - kt) / kt) for v in ks[:, i]], order=order, a=float(a), b=float(b),
                   extrap3_rel=float(abs(k0 - kt) / kt), richardson2_rel=float(abs(rich2 - kt) / kt))
        report.append(rec)
        # every resolution inside the discretisation error, error falling monotonically
        assert rec["rel_err"][0] < 3e-3 and rec["rel_err"][0] > rec["rel_err"][1] > rec["rel_err"][2], rec
        if pol == "TE":          # second order: plain Richardson is the sharp statement, the 3-term fit a sanity bound
            assert abs(order - 2.0) <= 0.3, rec
            assert rec["richardson2_rel"] <= RICHARDSON_TOL, rec
            assert rec["extrap3_rel"] <= 2 * RICHARDSON_TOL, rec
        else:                    # mixed order: the 3-term fit removes both terms
            assert 0.9 <= order <= 2.3, rec
            assert rec["extrap3_rel"] <= RICHARDSON_TOL, rec
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(dict(resolutions=RES, eigensolve_seconds=secs, entries=report),
                  open(os.path.join(out, "dsphmsph_pin_%s.json" % "_".join(lbcs)), "w"), indent=1)


def test_at_least_eight_table_entries_in_three_classes():
    """Book-keeping of the pin itself: >= 3 classes, >= 8 distinct entries of kTEs / kTMs."""
    seen = set()
    for lbcs, nev in CLASSES.items():
        seen |= {(pol, n, l) for _, pol, n, l in _entries(lbcs, nev)}
    assert len(CLASSES) >= 3 and len(seen) >= 8, sorted(seen)


def test_octmodes_pec_pec_pec_frequencies(ctx):
    """octModes(['pec']*3, 5) of the reference: 1.9368e8, 2.6663e8, 2.8683e8 Hz, five
    significant digits as recorded; the 128^3 solve reproduces them to that precision."""
    lbcs = ("pec", "pec", "pec")
    ks, _ = _class_data(ctx, lbcs)
    f = ks[-1] * TABLE["lightspeed"] / (2.0 * np.pi)
    want = TABLE["octModes_pec_pec_pec_5_Hz"]
    groups = [f[0:2], f[2:3], f[3:6]]                 # TE_02 x2, TM_03, TE_04 x3
    for g, w in zip(groups, want):
        assert np.all(np.abs(g - w) / w < 1.5e-4), (g, w)     # 5 recorded digits (5e-5) + the 128^3 error
    # and the analytic table agrees with the recorded output of the reference's function
    fk = expected_k(lbcs, 6) * TABLE["lightspeed"] / (2.0 * np.pi)
    for g, w in zip([fk[0:2], fk[2:3], fk[3:6]], want):
        assert np.all(np.abs(g - w) / w < 5e-5), (g, w)
