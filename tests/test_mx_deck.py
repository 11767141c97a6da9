"""`.mx` input decks: the C++ reader (include/mx/MxProblem.h via mxh_problem_*) against the
independent Python restatement oracle/mxdeck.py, on hand-written decks (tests/golden/*.mx; see the
comment at the top of each deck and DESIGN.md §5: not output of the reference's writer).

CPU-only: parsing and assembly need no GPU.  The GPU end-to-end run is tests/test_mx_cli_gpu.py.
"""
import os

import numpy as np
import pytest

import maxwell_amd as mx
from oracle import mxdeck

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
pytestmark = pytest.mark.usefixtures("built")
DECKS = ["pillbox.mx", "pillbox_mxpy.mx", "vacuum.mx", "tilted_cavity.mx", "dsphmsph.mx", "crabcell.mx", "magsph.mx"]
COMPLEX_DECKS = ["fiber.mx", "pillwtubes.mx", "pmlbox.mx"]


def _text(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return f.read()


def _same_number(a, b):
    return float(a) == float(b)


@pytest.mark.parametrize("name", DECKS)
def test_parameters_match_restatement(built, name):
    """Every parameter-list entry (src/MxProblem.cpp:82-105 keys), grid and BCs agree."""
    deck = mxdeck.Deck(_text(name))
    prob = mx.Problem(path=os.path.join(GOLDEN, name))
    g = prob.grid()
    assert g["n"] == deck.n and g["origin"] == deck.origin and g["size"] == deck.size
    assert g["lower"] == deck.lower and g["upper"] == deck.upper
    assert g["dm_frac"] == deck.dm_frac and g["has_pec"] == (deck.pec is not None)
    for key, val in deck.params.items():
        got = prob.param(key)
        if isinstance(val, bool):
            assert got == ("1" if val else "0"), key
        elif isinstance(val, (int, float)):
            assert _same_number(got, val), (key, got, val)
        else:
            assert got == val, key


def test_defaults_of_the_reference_tables():
    """A deck with nothing but the polarization takes the defaults of src/MxProblem.cpp:26-66,445-459,604-612."""
    prob = mx.Problem(text="<Sim>\n polarization = TE\n</Sim>")
    assert prob.grid() == dict(n=[16] * 3, origin=[0.0] * 3, size=[1.0] * 3, lower=["periodic"] * 3,
                               upper=["periodic"] * 3, dm_frac=0.0, has_pec=False)
    expect = {"eigensolver : type": "krylov-schur", "eigensolver : basis": "20", "eigensolver : max restarts": "100",
              "eigensolver : invert": "1", "eigensolver : block size": "1", "eigensolver : nev": "10",
              "eigensolver : spectrum": "LM", "eigensolver : output": "2", "linear solver : type": "gmres",
              "linear solver : basis": "40"}
    for k, v in expect.items():
        assert prob.param(k) == v, k
    assert float(prob.param("eigensolver : tol")) == 1.0e-6
    assert float(prob.param("linear solver : tol")) == 1.0e-6
    with pytest.raises(KeyError):
        prob.param("linear solver : prec type")


def test_body_line_wins_over_tag_attribute():
    """MxUtil::XML::getAttr: body first, then tag attribute (src/MxUtil.cpp:266-280)."""
    prob = mx.Problem(text='<S polarization="TM">\n <Eigensolver nev="4">\n  nev = 7\n  <LinearSolver/>\n'
                           ' </Eigensolver>\n</S>')
    assert prob.param("eigensolver : nev") == "7"
    assert prob.param("polarization") == "TM"


@pytest.mark.parametrize("text,needle", [
    ("<S>\n</S>", "polarization"),                                              # required attribute
    ("<S>\n polarization = XX\n</S>", "Invalid polarization"),
    ("<S>\n polarization = TE\n <Eigensolver>\n nev = 2\n </Eigensolver>\n</S>", "LinearSolver"),
    ("<S>\n polarization = TE\n <BoundaryConditions>\n lower bcs = [pec, foo, pec]\n </BoundaryConditions>\n</S>",
     "Invalid lower boundary condition"),
    ("<S>\n polarization = TE\n <Dielectric>\n </Dielectric>\n</S>", "Dielectric"),
    ("<S>\n polarization = TE\n <Mu>\n </Mu>\n</S>", "Mu object"),
    ("<S>\n polarization = TE\n <PML>\n </PML>\n</S>", "PML object"),
    ("<S>\n polarization = TE\n <PEC>\n <Torus>\n name = t\n </Torus>\n </PEC>\n</S>", "major radius"),
    ("<S>\n polarization = TE\n <PEC>\n <ShapeSubtract>\n <Sphere>\n name = a\n </Sphere>\n </ShapeSubtract>\n </PEC>\n</S>",
     "exactly 2"),
    ("<S>\n polarization = TE\n <PEC>\n <ShapeMirror>\n normal = [0,0,1]\n point in plane = [0,0,0]\n </ShapeMirror>\n </PEC>\n</S>",
     "one shape block"),
    ("<S>\n polarization = TE\n <Eigensolver>\n <LinearSolver>\n <Preconditioner>\n type = foo\n </Preconditioner>\n"
     " </LinearSolver>\n </Eigensolver>\n</S>", "Preconditioner type"),
    ("<S>\n polarization = TE\n <Grid>\n</S>", "parse error"),
    ("<S>\n polarization = TE\n", "parse error"),
])
def test_errors_are_reported(text, needle):
    with pytest.raises(mx.MxError) as e:
        mx.Problem(text=text)
    assert needle in str(e.value)


@pytest.mark.parametrize("name", ["pillbox.mx", "tilted_cavity.mx", "crabcell.mx", "pillwtubes.mx"])
def test_pec_level_set_bit_identical(name):
    """Shape factory + transforms in document order: the level set equals the restatement's bit for bit.
    crabcell / pillwtubes cover every remaining shape tag of src/MxProblem.cpp:8-17 (Ellipsoid, Torus, Cone,
    ShapeSubtract / Union / Mirror / Repeat)."""
    deck = mxdeck.Deck(_text(name))
    prob = mx.Problem(text=_text(name))
    rng = np.random.default_rng(7)
    lo = np.array(deck.origin)
    pts = lo + rng.random((400, 3)) * np.array(deck.size)
    want = deck.pec.func(pts)
    got = np.array([prob.pec_func(p) for p in pts])
    assert np.array_equal(got, want)
    assert (want > 0).any() and (want < 0).any()


@pytest.mark.parametrize("name", DECKS)
def test_assembled_operators_bit_identical(name):
    """Deck -> assembled operators: C++ (mxh_problem_sim) == oracle (Deck.sim), values and pattern."""
    deck = mxdeck.Deck(_text(name))
    osim = deck.sim()
    ops = osim.wave_ops()
    sim = mx.Problem(text=_text(name)).sim()
    for field in ("bfield", "efield", "psifield"):
        want = {"bfield": osim.B, "efield": osim.E, "psifield": osim.P}[field].gids
        assert np.array_equal(sim.map(field), want), field
    for opname, ref in ops.items():
        a, b = sim.op(opname).tocsr(), ref.tocsr()
        a.sort_indices(); b.sort_indices()
        d = a - b            # the product keeps explicit zeros (as Epetra does): compare on the union pattern
        assert a.shape == b.shape and (d.nnz == 0 or np.abs(d.data).max() == 0.0), opname


def test_mu_stencil_rows_on_wall_faces_keep_their_entries():
    """The reference multiplies the 9-point mu stencil by the factors of its COLUMNS (src/MxYeeFitMu.cpp:441-442),
    not of its row: where a non-diagonal <Mu> reaches a PMC/PEC wall, rows of wall faces whose own factor is 0 keep
    off-diagonal entries (their columns stay empty).  Reproduced bit for bit - it is the reference's matrix - and
    recorded in DESIGN.md section 7; tests/golden/magsph.mx keeps the magnetic sphere clear of the walls."""
    text = _text("magsph.mx").replace("origin = [-0.5, -0.5, -0.5]", "origin = [0.0, 0.0, 0.0]").replace(
        "dimensions = [1.0, 1.0, 1.0]", "dimensions = [0.5, 0.5, 0.5]").replace(
        "lower bcs = [pec, pec, pec]", "lower bcs = [pmc, pmc, pec]").replace("vector = [0.06, -0.04, 0.03]", "vector = [0.0, 0.0, 0.0]")
    ops = mxdeck.Deck(text).sim().wave_ops()
    sim = mx.Problem(text=text).sim()
    for opname, ref in ops.items():
        d = sim.op(opname).tocsr() - ref.tocsr()
        assert d.nnz == 0 or np.abs(d.data).max() == 0.0, opname
    m = sim.op("mRhs").tocsr()
    unused = sim.used_mask() == 0
    assert (np.abs(m).sum(axis=1).A1[unused] > 0).sum() > 10 and np.abs(m).sum(axis=0).A1[unused].max() == 0.0


def test_dm_frac_zeroes_small_faces():
    """<PEC> 'D-M frac' (src/MxYeeFitBField.cpp:83-90): B faces below the threshold get factor 0 -
    the map is unchanged, curlE rows of those faces vanish."""
    base = _text("tilted_cavity.mx")
    a = mx.Problem(text=base.replace("D-M frac = 0.05", "D-M frac = 0.0")).sim()
    b = mx.Problem(text=base).sim()
    assert np.array_equal(a.map("bfield"), b.map("bfield"))
    ca, cb = a.op("curlE"), b.op("curlE")
    rows_a = np.diff(ca.indptr) > 0
    rows_b = np.abs(cb).sum(axis=1).A1 > 0
    assert rows_b.sum() < (np.abs(ca).sum(axis=1).A1 > 0).sum()
    assert not (rows_b & ~rows_a).any()


@pytest.mark.parametrize("name", COMPLEX_DECKS)
def test_complex_deck_operators_bit_identical(name):
    """Bloch phase shifts: the complex assembler (YeeAssembler<complex>) against the complex oracle, incl. the FIT
    inverse permittivity and a two-cell periodic direction (repeated (row, col) entries are summed)."""
    deck = mxdeck.Deck(_text(name))
    assert deck.params["is complex"]
    prob = mx.Problem(text=_text(name))
    assert prob.is_complex and prob.param("is complex") == "1"
    with pytest.raises(mx.MxError, match="complex"):
        prob.sim()
    ops = deck.sim().wave_ops()
    for opname, ref in ops.items():
        a, b = prob.op(opname).tocsr(), ref.tocsr()
        assert np.iscomplexobj(a.data) and a.shape == b.shape
        d = a - b
        assert d.nnz == 0 or np.abs(d.data).max() == 0.0, opname


def _complex_ops_same(text, strict=True):
    deck = mxdeck.Deck(text)
    assert deck.params["is complex"]
    prob = mx.Problem(text=text, strict=strict)
    assert prob.is_complex
    for opname, ref in deck.sim(strict=strict).wave_ops().items():
        a, b = prob.op(opname).tocsr(), ref.tocsr()
        d = a - b
        assert a.shape == b.shape and (d.nnz == 0 or np.abs(d.data).max() == 0.0), opname
    return deck, prob


def test_loss_tangent_makes_the_problem_complex():
    """<Dielectric> `loss tangent` t: eps_ii (1 + i t) (src/MxProblem.cpp:505-513) - complex FIT inverse
    permittivity (tupleUpdate in complex arithmetic), complex operators, bit-identical to the oracle."""
    text = _text("dsphmsph.mx").replace("eps off diag = [0.0, 0.0, 0.0]", "eps off diag = [0.0, 0.0, 0.0]\n    loss tangent = 0.001")
    text = text.replace("resolution = [12, 12, 12]", "resolution = [8, 8, 8]")
    deck, prob = _complex_ops_same(text)
    ie = prob.op("invEps")
    assert np.abs(ie.data.imag).max() > 1e-5


def test_lossy_mu_makes_the_problem_complex():
    """<Mu> `loss tangent` t: mu_ii (1 + i t) (src/MxProblem.cpp:536-547) - complex FIT permeability (diagonal path and
    the 9-point tuple stencil in complex arithmetic), complex invMuVolAve, complex chain; bit-identical to the oracle."""
    text = _text("magsph.mx").replace("mu off diag = [0.1, -0.2, 0.3]", "mu off diag = [0.1, -0.2, 0.3]\n    loss tangent = 0.02")
    text = text.replace("resolution = [10, 10, 10]", "resolution = [8, 8, 8]")
    deck, prob = _complex_ops_same(text)
    assert np.abs(prob.op("mu").data.imag).max() > 1e-3 and np.abs(prob.op("invMuVolAve").data.imag).max() > 1e-4


def test_complex_inverse_permittivity_with_bloch_phases():
    """A tilted fibre in a periodic box with Bloch phases (strict=False keeps the wrap-around entries): stencil
    members across the periodic seam carry e^{+-i phi} into the FIT inverse permittivity."""
    text = _text("fiber.mx").replace("axis = [1.0, 0.0, 0.0]", "axis = [1.0, 0.4, 0.2]").replace("resolution = [2, 12, 12]", "resolution = [4, 10, 10]")
    text = text.replace("lower bcs = [periodic, pec, pec]", "lower bcs = [periodic, periodic, pec]").replace(
        "upper bcs = [periodic, pec, pec]", "upper bcs = [periodic, periodic, pec]").replace("phase shifts = [0.9, 0.0, 0.0]", "phase shifts = [0.9, -0.5, 0.0]")
    deck, prob = _complex_ops_same(text, strict=False)
    ie = prob.op("invEps")
    assert np.abs(ie.data.imag).max() > 1e-3


def _random_shape_xml(rng, depth, counter):
    """A random CSG tree in deck syntax over every shape tag of src/MxProblem.cpp:8-17, with random transforms."""
    def vec(lo, hi, n=3):
        return "[" + ", ".join("%.4f" % v for v in rng.uniform(lo, hi, n)) + "]"

    def transforms():
        out = []
        for _ in range(int(rng.integers(0, 3))):
            kind = rng.choice(["Translate", "Rotate", "Scale", "Reflect"])
            if kind == "Translate":
                out.append("<Translate>\n vector = %s\n</Translate>" % vec(-0.15, 0.15))
            elif kind == "Rotate":
                out.append("<Rotate>\n axis = %s\n angle = %.4f\n</Rotate>" % (vec(-1, 1), rng.uniform(-1, 1)))
            elif kind == "Scale":
                out.append("<Scale>\n magnitudes = %s\n</Scale>" % vec(0.7, 1.3))
            else:
                out.append("<Reflect>\n normal = %s\n point in plane = %s\n</Reflect>" % (vec(-1, 1), vec(-0.1, 0.1)))
        return "\n".join(out)

    counter[0] += 1
    name = "s%d" % counter[0]
    leaf = depth == 0 or rng.random() < 0.35
    inv = "\n invert = true" if rng.random() < 0.15 else ""
    if leaf:
        kind = rng.choice(["Sphere", "Cylinder", "Slab", "HalfSpace", "Ellipsoid", "Torus", "Cone"])
        body = {"Sphere": "radius = %.4f" % rng.uniform(0.15, 0.45),
                "Cylinder": "radius = %.4f\n axis = %s" % (rng.uniform(0.1, 0.4), vec(-1, 1)),
                "Slab": "thickness = %.4f\n normal = %s" % (rng.uniform(0.1, 0.6), vec(-1, 1)),
                "HalfSpace": "normal = %s" % vec(-1, 1),
                "Ellipsoid": "axes = %s" % vec(0.15, 0.45),
                "Torus": "major radius = %.4f\n minor radius = %.4f\n axis = %s" % (rng.uniform(0.2, 0.35), rng.uniform(0.05, 0.12), vec(-1, 1)),
                "Cone": "angle = %.4f\n axis = %s" % (rng.uniform(0.3, 1.2), vec(-1, 1))}[kind]
        return "<%s>\n name = %s%s\n %s\n%s\n</%s>" % (kind, name, inv, body, transforms(), kind)
    kind = rng.choice(["ShapeIntersection", "ShapeUnion", "ShapeSubtract", "ShapeMirror", "ShapeRepeat"])
    if kind in ("ShapeIntersection", "ShapeUnion"):
        kids = "\n".join(_random_shape_xml(rng, depth - 1, counter) for _ in range(int(rng.integers(2, 4))))
        return "<%s>\n name = %s%s\n%s\n%s\n</%s>" % (kind, name, inv, kids, transforms(), kind)
    if kind == "ShapeSubtract":
        kids = "\n".join(_random_shape_xml(rng, depth - 1, counter) for _ in range(2))
        return "<ShapeSubtract>\n name = %s%s\n%s\n%s\n</ShapeSubtract>" % (name, inv, kids, transforms())
    kid = _random_shape_xml(rng, depth - 1, counter)
    if kind == "ShapeMirror":
        return "<ShapeMirror>\n name = %s%s\n normal = %s\n point in plane = %s\n%s\n%s\n</ShapeMirror>" % (
            name, inv, vec(-1, 1), vec(-0.1, 0.1), kid, transforms())
    return ("<ShapeRepeat>\n name = %s%s\n origin = %s\n direction = %s\n step = %.4f\n pos steps = %d\n neg steps = %d\n%s\n%s\n"
            "</ShapeRepeat>" % (name, inv, vec(-0.1, 0.1), vec(-1, 1), rng.uniform(0.2, 0.5), rng.integers(0, 3),
                                rng.integers(0, 3), kid, transforms()))


@pytest.mark.parametrize("seed", range(12))
def test_random_csg_trees_bit_identical(seed):
    """Random trees over all twelve shape tags with random transforms on primitives and composites: the C++ factory
    (include/mx/MxProblem.h + mx_geometry.cpp) and the oracle's (oracle/mxdeck.py + geometry.py) give the same level
    set bit for bit at random points - the composition rules of src/MxShape{Union,Subtract,Mirror,Repeat}.hpp and the
    transform order of src/MxShape.cpp:30-60 are exercised far beyond the hand-written decks."""
    rng = np.random.default_rng(1000 + seed)
    shape = _random_shape_xml(rng, 3, [0])
    text = ("<S>\n polarization = TE\n <Grid>\n resolution = [4, 4, 4]\n origin = [-0.5, -0.5, -0.5]\n </Grid>\n"
            " <PEC>\n%s\n </PEC>\n</S>" % shape)
    deck = mxdeck.Deck(text)
    prob = mx.Problem(text=text)
    pts = rng.uniform(-0.6, 0.6, (600, 3))
    want = deck.pec.func(pts)
    got = np.array([prob.pec_func(p) for p in pts])
    ok = np.isfinite(want)
    assert np.array_equal(got[ok], want[ok]), shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    scalar = np.array([deck.pec.func1(tuple(p)) for p in pts[:100].tolist()])
    assert np.array_equal(scalar[ok[:100]], want[:100][ok[:100]])


@pytest.mark.parametrize("seed", range(6))
def test_random_csg_assembly_bit_identical(seed):
    """The same random trees as PEC surfaces of a small grid: cut-cell fractions need the composite shapes' gradients
    (Newton steps of the root finder) - every assembled operator equals the oracle's bit for bit."""
    rng = np.random.default_rng(5000 + seed)
    shape = _random_shape_xml(rng, 2, [0])
    text = ("<S>\n polarization = TE\n <Grid>\n resolution = [6, 5, 7]\n origin = [-0.52, -0.5, -0.49]\n </Grid>\n"
            " <BoundaryConditions>\n lower bcs = [pec, pec, pec]\n upper bcs = [pec, pec, pec]\n </BoundaryConditions>\n"
            " <PEC>\n%s\n </PEC>\n</S>" % shape)
    deck = mxdeck.Deck(text)
    ops = deck.sim().wave_ops()
    sim = mx.Problem(text=text).sim()
    for opname, ref in ops.items():
        a, b = sim.op(opname).tocsr(), ref.tocsr()
        assert a.shape == b.shape, opname
        d = a - b
        if d.nnz:
            bad = np.abs(d.data)
            assert np.all((bad == 0.0) | np.isnan(bad)), (opname, shape)      # NaN fractions (degenerate cuts) on both sides
            assert np.array_equal(np.isnan(a.toarray()), np.isnan(b.toarray())), opname


@pytest.mark.parametrize("seed", range(5))
def test_random_dielectric_shapes_bit_identical(seed):
    """Random composite shapes as (anisotropic) dielectrics: the FIT inverse permittivity takes the interface normal from
    the composite's gradient (src/MxYeeFitInvEps.cpp stencil normal) - invEps, invEpsVolAve and every product equal the
    oracle's bit for bit."""
    rng = np.random.default_rng(9000 + seed)
    shape = _random_shape_xml(rng, 2, [0])
    diag = rng.uniform(2.0, 12.0, 3)
    off = rng.uniform(-0.4, 0.4, 3) if seed % 2 else np.zeros(3)
    text = ("<S>\n polarization = TE\n <Grid>\n resolution = [6, 6, 5]\n origin = [-0.5, -0.52, -0.49]\n </Grid>\n"
            " <BoundaryConditions>\n lower bcs = [pec, pmc, periodic]\n upper bcs = [pec, pec, periodic]\n </BoundaryConditions>\n"
            " <Dielectric>\n name = blob\n eps diag = [%r, %r, %r]\n eps off diag = [%r, %r, %r]\n%s\n </Dielectric>\n</S>"
            % tuple([float(v) for v in diag] + [float(v) for v in off] + [shape]))
    deck = mxdeck.Deck(text)
    ops = deck.sim().wave_ops()
    sim = mx.Problem(text=text).sim()
    assert "invEps" in ops
    for opname, ref in ops.items():
        a, b = sim.op(opname).tocsr(), ref.tocsr()
        assert a.shape == b.shape, opname
        d = a - b
        if d.nnz:
            bad = np.abs(d.data)
            assert np.all((bad == 0.0) | np.isnan(bad)), (opname, float(np.nanmax(bad)), shape)
            assert np.array_equal(np.isnan(a.toarray()), np.isnan(b.toarray())), opname
