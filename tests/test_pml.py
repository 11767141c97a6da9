"""Perfectly matched layers (<PML>): src/MxPML.h and the PML terms of the FIT material
operators (src/MxYeeFitInvEps.cpp:514-567,779-791, src/MxYeeFitMu.cpp:385-434,765-777).

The reference holds no fixture with a PML, so the oracle is pinned on an exact property: a layer that covers the whole
box with a constant stretch (exponent 0 -> alpha = max alpha everywhere) is the coordinate change x -> s x, s = 1 + i
alpha, and the Yee spectrum of a PEC box then is sum_i (2 / (s_i h_i) sin(pi m_i / (2 N_i)))^2 exactly.  The product
assembler (C++, complex) is compared with the oracle bit for bit in tests/test_mx_deck.py (pmlbox.mx) and here."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import maxwell_amd as mx
from oracle import geometry as geo
from oracle import mxdeck, yee

UNIFORM = """<S>
 polarization = TE
 <Grid>
  resolution = [6, 5, 4]
  dimensions = [1.0, 0.8, 0.6]
 </Grid>
 <BoundaryConditions>
  lower bcs = [pec, pec, pec]
  upper bcs = [pec, pec, pec]
 </BoundaryConditions>
 <Dielectric>
  <Sphere>
   name = a
   radius = 10
  </Sphere>
 </Dielectric>
 <Mu>
  <Sphere>
   name = b
   radius = 10
  </Sphere>
 </Mu>
 <PML>
  max alpha = 0.3
  exponent = 0.0
  profile begin = [0, 0, 0]
  profile end = DIR
  <Sphere>
   name = c
   radius = 10
  </Sphere>
 </PML>
 <Eigensolver>
  nev = 4
  shift = 1.0
  tol = 1e-9
  basis = 30
  <LinearSolver>
   tol = 1e-11
   <Preconditioner>
    type = amg
   </Preconditioner>
  </LinearSolver>
 </Eigensolver>
</S>
"""


def stretched_box_spectrum(n, size, s, shift, want):
    """k^2 of a PEC box on the Yee grid with complex cell sizes s_i h_i, ordered as shift-invert finds them."""
    h = np.array(size) / np.array(n)
    out = []
    for m in np.ndindex(*(np.array(n) + 1)):
        nz = sum(1 for v in m if v > 0)
        if nz >= 2:
            k2 = sum((2.0 / (s[i] * h[i]) * np.sin(np.pi * m[i] / (2 * n[i]))) ** 2 for i in range(3))
            out += [k2] * (2 if nz == 3 else 1)
    out = np.array(out)
    return out[np.argsort(-np.abs(1.0 / (out - shift)))][:want]


@pytest.mark.parametrize("direction", [(1.0, 0.0, 0.0), (0.0, 0.0, 2.0)])
def test_constant_stretch_gives_the_exact_complex_spectrum(direction):
    d = np.array(direction) / np.linalg.norm(direction)
    deck = mxdeck.Deck(UNIFORM.replace("DIR", str(list(direction))))
    assert deck.params["is complex"]
    got = yee.mag_wave_modes(deck.sim().wave_ops(), 1.0, 4)
    s = [1.0 + 0.3j * d[i] for i in range(3)]
    want = stretched_box_spectrum(deck.n, deck.size, s, 1.0, 4)
    assert np.allclose(np.sort_complex(got), np.sort_complex(want), rtol=1e-10)
    assert got.imag.min() < -1.0 and got.imag.max() < 1e-9      # damped, except modes that do not vary along the stretch


def test_oblique_layer_direction_takes_the_full_tensor(built):
    """dir = (1, 1, 0) / sqrt 2: S is not diagonal - the 9-point stencil couples the components.  C++ == oracle bit for
    bit, and the inverse permittivity really has off-diagonal entries."""
    text = UNIFORM.replace("DIR", "[1.0, 1.0, 0.0]")
    deck = mxdeck.Deck(text)
    ops = deck.sim().wave_ops()
    prob = mx.Problem(text=text)
    for name, ref in ops.items():
        d = prob.op(name).tocsr() - ref.tocsr()
        assert d.nnz == 0 or np.abs(d.data).max() == 0.0, name
    ie = ops["invEps"].tocsr()
    assert np.abs(ie - sp.diags(ie.diagonal())).max() > 1e-3


def test_layer_outside_changes_nothing_inside(built):
    """Rows whose coordinate is outside every layer keep the plain material entries (pmlTransf = I)."""
    with open(os.path.join(os.path.dirname(__file__), "golden", "pmlbox.mx")) as f:
        text = f.read()
    with_pml = mxdeck.Deck(text).sim()
    without = mxdeck.Deck(re.sub(r"\n  <PML>\n.*?\n  </PML>\n", "\n", text, flags=re.S)).sim()
    a, b = with_pml.inv_eps().tocsr(), without.inv_eps().tocsr()
    E = with_pml.E
    x = np.array([with_pml.comp_coord(E, int(E.comp[r]), E.cells[r])[0] for r in range(len(E.gids))])
    clear = x <= 0.6
    assert np.abs((a - b)[clear]).max() < 1e-14 and np.abs((a - b)[~clear]).max() > 0.1    # complex vs real arithmetic: 1 ulp


def test_pml_deck_errors():
    with pytest.raises(mx.MxError, match="PML object needs"):
        mx.Problem(text="<S>\n polarization = TE\n <PML>\n </PML>\n</S>")
    with pytest.raises(mx.MxError, match="profile begin"):
        mx.Problem(text="<S>\n polarization = TE\n <PML>\n <Sphere>\n name = a\n </Sphere>\n </PML>\n</S>")
