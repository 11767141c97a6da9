"""Expected spectrum of the reference's config-5 physics (example/run.py:13-28,160-182: octant box [0, 0.5]^3,
eps = 10 sphere r 0.37 in a PEC sphere r 0.49, upper walls PEC, lower walls one of the 8 symmetry classes of
example/dsphmsph.py:203-210) from the reference-held analytic table tests/golden/dsphmsph_ktable.json.

Which multipoles live in which symmetry class is the rule of example/dsphmsph.py:438-498, restated:
  TM, real part:  x = 0 wall pec if m is odd else pmc;  y = 0 wall pmc;  z = 0 wall pec if l - m is odd else pmc
  TM, imag part:  the x and y walls flipped, z as for the real part   (m = 0 has no imaginary part)
  TE:             every wall the opposite of the same-phase TM mode
Each (pol, n, l) of the table is (2l + 1)-fold degenerate in the full sphere; one octant class sees the
(m, phase) members that satisfy its three walls."""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TABLE = json.load(open(os.path.join(_HERE, "golden", "dsphmsph_ktable.json")))
FLIP = {"pec": "pmc", "pmc": "pec"}


def oct_lower_bcs(pol, l, m, phase):
    """example/dsphmsph.py:438-498."""
    bcs = ["pec" if m % 2 else "pmc", "pmc", "pec" if (l - m) % 2 else "pmc"]        # TM, real
    if phase == "im":
        bcs = [FLIP[bcs[0]], FLIP[bcs[1]], bcs[2]]
    if pol == "TE":
        bcs = [FLIP[b] for b in bcs]
    return bcs


def class_modes(lbcs, kmax):
    """[(k, pol, n, l, multiplicity)] of the table's modes with k <= kmax that the class `lbcs` contains,
    ascending in k (the selection loop of octModes, example/dsphmsph.py:236-246)."""
    out = []
    for pol, tab in (("TE", TABLE["kTE"]), ("TM", TABLE["kTM"])):
        for key, k in tab.items():
            n, l = (int(v) for v in key.split(","))
            mult = 0
            for m in range(l + 1):
                for phase in ("re", "im"):
                    if m == 0 and phase == "im":
                        continue
                    if list(lbcs) == oct_lower_bcs(pol, l, m, phase):
                        mult += 1
            if mult and k <= kmax:
                out.append((k, pol, n, l, mult))
    return sorted(out)


def expected_k(lbcs, count):
    """The lowest `count` wave numbers of the class, degenerate members repeated; only complete below the
    smallest k the table leaves out (l = 8, n = 0: above 9.7; radial index 3 of l = 1: above 8.16)."""
    ks = []
    for k, _, _, _, mult in class_modes(lbcs, 8.0):
        ks += [k] * mult
    return np.array(ks[:count])


def deck_text(n, lbcs, nev, basis=30, tol=1e-9, lin_tol=1e-11, block=3):
    """Hand-written deck of the run.py problem (the same blocks as tests/golden/dsphmsph.mx)."""
    return """<Simulation>
  name = dsphmsph
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
  <PEC>
    <Sphere>
      name = pecSphere
      radius = 0.49
    </Sphere>
  </PEC>
  <Dielectric>
    eps diag = [10.0, 10.0, 10.0]
    eps off diag = [0.0, 0.0, 0.0]
    <Sphere>
      name = epsSphere
      radius = 0.37
    </Sphere>
  </Dielectric>
  <Eigensolver>
    nev = %d
    basis = %d
    block size = %d
    shift = 0.5
    tol = %r
    <LinearSolver>
      type = gmres
      basis = 40
      tol = %r
      <Preconditioner>
        type = amg
      </Preconditioner>
    </LinearSolver>
  </Eigensolver>
</Simulation>
""" % (n, n, n, lbcs[0], lbcs[1], lbcs[2], nev, basis, block, tol, lin_tol)
