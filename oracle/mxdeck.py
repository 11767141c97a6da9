"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's `.mx` deck reader.

Follows src/MxProblem.cpp:225-691 (block dispatch, parameter tables, shape factory),
src/MxShape.cpp:30-60 (generic transforms, applied in document order) and
src/MxUtil.cpp:227-315 (`getAttr`: body line `name = value`, then tag attribute, then default).
Built on xml.etree so that it shares nothing with the C++ reader it checks
(include/mx/MxProblem.h).  Parity unpinned beyond the grammar: the decks under tests/golden/ are
hand-written, not output of the reference's Python writer.
"""
import xml.etree.ElementTree as ET

import numpy as np

from . import geometry as geo
from . import yee
from .yee import YeeSim

SHAPE_TAGS = ("Ellipsoid", "Sphere", "Cylinder", "Torus", "Cone", "Slab", "HalfSpace", "ShapeSubtract",
              "ShapeIntersection", "ShapeUnion", "ShapeMirror", "ShapeRepeat")

# name, default, type (src/MxProblem.cpp:20-66)
EIG_PARAMS = [("type", "krylov-schur", str), ("basis", "20", int), ("max restarts", "100", int),
              ("invert", "true", bool), ("block size", "1", int), ("nev", "10", int), ("tol", "1.0e-6", float),
              ("spectrum", "LM", str), ("output", "2", int), ("shift", "0.0", float)]
LIN_PARAMS = [("type", "gmres", str), ("tol", "1.0e-6", float), ("basis", "40", int)]
AMG_PARAMS = [("smoother", "Chebyshev", str), ("sweeps", "2", int), ("coarse smoother", "Chebyshev", str),
              ("max levels", "10", int)]
ILU_PARAMS = [("drop tol", "1.e-3", float), ("fill", "10.", float)]


def body_lines(node):
    """Lines of the block that are not inside a nested block (MxUtil::XML::nodeLines)."""
    chunks = [node.text or ""] + [(c.tail or "") for c in node]
    out = []
    for ch in chunks:
        for line in ch.split("\n"):
            line = line.strip()
            if line:
                out.append(line)
    return out


def get_attr(name, node, default=None):
    if node is None:
        return default
    for line in body_lines(node):
        left, _, right = line.partition("=")
        if left.strip() == name and right.strip() != "":
            return right.strip()
    if node.get(name, "") != "":
        return node.get(name)
    if default is None:
        raise KeyError("attribute '%s' is required in <%s>" % (name, node.tag))
    return default


def items(s):
    for ch in ",[]\t":
        s = s.replace(ch, " ")
    return s.split()


def vec(s, dim=3, fill=0.0):
    v = [float(x) for x in items(s)]
    return (v + [fill] * dim)[:dim]


def _convert(sval, typ):
    if typ is int:
        return int(float(sval)) if "." in sval or "e" in sval.lower() else int(sval)
    if typ is float:
        return float(sval)
    if typ is bool:
        return sval == "true"
    return sval


def transforms(shape, node):
    if get_attr("invert", node, "false") == "true":
        shape.invert()
    for child in node:
        if child.tag == "Rotate":
            pivot = get_attr("pivot", child, "")
            shape.rotate(vec(get_attr("axis", child)), float(get_attr("angle", child)),
                         vec(pivot) if pivot != "" else None)
        elif child.tag == "Translate":
            shape.translate(vec(get_attr("vector", child)))
        elif child.tag == "Reflect":
            shape.reflect(vec(get_attr("normal", child)), vec(get_attr("point in plane", child)))
        elif child.tag == "Scale":
            shape.scale(vec(get_attr("magnitudes", child), fill=1.0))
    return shape


def make_shape(node):
    zero = (0.0, 0.0, 0.0)
    tag = node.tag
    if tag == "Sphere":
        get_attr("name", node)
        return transforms(geo.Sphere(float(get_attr("radius", node, "1")), zero), node)
    if tag == "Cylinder":
        get_attr("name", node)
        return transforms(geo.Cylinder(float(get_attr("radius", node)), vec(get_attr("axis", node)), zero), node)
    if tag == "Slab":
        get_attr("name", node)
        return transforms(geo.Slab(float(get_attr("thickness", node)), vec(get_attr("normal", node)), zero), node)
    if tag == "HalfSpace":
        get_attr("name", node)
        return transforms(geo.HalfSpace(zero, vec(get_attr("normal", node))), node)
    if tag == "ShapeIntersection":
        parts = [make_shape(c) for c in node if c.tag in SHAPE_TAGS]
        return transforms(geo.Intersection(parts), node)
    if tag == "Ellipsoid":
        get_attr("name", node)
        return transforms(geo.Ellipsoid(vec(get_attr("axes", node)), zero), node)
    if tag == "Torus":
        get_attr("name", node)
        return transforms(geo.Torus(float(get_attr("major radius", node)), float(get_attr("minor radius", node)),
                                    vec(get_attr("axis", node)), zero), node)
    if tag == "Cone":
        get_attr("name", node)
        return transforms(geo.Cone(float(get_attr("angle", node)), vec(get_attr("axis", node)), zero), node)
    parts = [make_shape(c) for c in node if c.tag in SHAPE_TAGS]
    if tag == "ShapeUnion":
        return transforms(geo.Union(parts), node)
    if tag == "ShapeSubtract":                                    # src/MxProblem.cpp:297-314: base, then the shape removed
        if len(parts) != 2:
            raise ValueError("'ShapeSubtract' should have exactly 2 shape subobjects")
        return transforms(geo.Subtract(parts[0], [parts[1]]), node)
    if tag == "ShapeMirror":
        if len(parts) != 1:
            raise ValueError("ShapeMirror XML block must contain one shape block")
        return transforms(geo.Mirror(parts[0], vec(get_attr("normal", node)), vec(get_attr("point in plane", node))), node)
    if tag == "ShapeRepeat":
        if len(parts) != 1:
            raise ValueError("ShapeRepeat XML block must contain one shape block")
        return transforms(geo.Repeat(parts[0], vec(get_attr("origin", node)), vec(get_attr("direction", node)),
                                     float(get_attr("step", node)), int(get_attr("pos steps", node)),
                                     int(get_attr("neg steps", node))), node)
    raise ValueError("did not understand shape tag '%s'" % tag)


class Deck:
    """Parsed deck: grid, boundary conditions, PEC level set and the flat parameter list."""

    def __init__(self, text):
        root = ET.fromstring(text)
        self.tag = root.tag
        self.params = {"polarization": get_attr("polarization", root), "is complex": False}
        self.n, self.origin, self.size = [16] * 3, [0.0] * 3, [1.0] * 3
        self.lower, self.upper, self.phase = ["periodic"] * 3, ["periodic"] * 3, [0.0] * 3
        self.pec, self.dm_frac = None, 0.0
        self.dielectrics = []
        self.mus = []
        self.pmls = []
        eig = None
        for child in root:
            if child.tag == "Grid":
                self.n = [int(x) for x in vec(get_attr("resolution", child, "[16, 16, 16]"))]
                self.size = vec(get_attr("dimensions", child, "[1.0, 1.0, 1.0]"))
                self.origin = vec(get_attr("origin", child, "[0.0, 0.0, 0.0]"))
            elif child.tag == "BoundaryConditions":
                self.lower = (items(get_attr("lower bcs", child, "[periodic, periodic, periodic]")) + [""] * 3)[:3]
                self.upper = (items(get_attr("upper bcs", child, "[periodic, periodic, periodic]")) + [""] * 3)[:3]
                self.phase = vec(get_attr("phase shifts", child, "[0.0, 0.0, 0.0]"))
            elif child.tag == "PEC":
                self.dm_frac = float(get_attr("D-M frac", child, "0.0"))
                self.pec = make_shape([c for c in child if c.tag in SHAPE_TAGS][0])
            elif child.tag == "Dielectric":                      # src/MxProblem.cpp:487-527
                name = get_attr("name", child, "dielectric%d" % len(self.dielectrics))
                diag = vec(get_attr("eps diag", child, "[1.0, 1.0, 1.0]"))
                off = vec(get_attr("eps off diag", child, "[0.0, 0.0, 0.0]"))
                losstan = float(get_attr("loss tangent", child, "0.0"))
                eps = np.diag(diag) if losstan == 0.0 else np.diag([complex(v, v * losstan) for v in diag])
                if losstan != 0.0:
                    self.params["is complex"] = True
                eps[0, 1] = eps[1, 0] = off[2]
                eps[0, 2] = eps[2, 0] = off[1]
                eps[1, 2] = eps[2, 1] = off[0]
                self.dielectrics.append((make_shape([c for c in child if c.tag in SHAPE_TAGS][0]), eps, name))
            elif child.tag == "Mu":                              # src/MxProblem.cpp:515-556
                name = get_attr("name", child, "mu%d" % len(self.mus))
                diag = vec(get_attr("mu diag", child, "[1.0, 1.0, 1.0]"))
                off = vec(get_attr("mu off diag", child, "[0.0, 0.0, 0.0]"))
                tan = float(get_attr("loss tangent", child, "0.0"))
                mu = np.diag(diag)
                if tan != 0:                                     # :536-547
                    mu = mu.astype(complex) * complex(1.0, 0.0)
                    for i in range(3):
                        mu[i, i] = complex(diag[i], diag[i] * tan)
                    self.params["is complex"] = True
                mu[0, 1] = mu[1, 0] = off[2]
                mu[0, 2] = mu[2, 0] = off[1]
                mu[1, 2] = mu[2, 1] = off[0]
                self.mus.append((make_shape([c for c in child if c.tag in SHAPE_TAGS][0]), mu, name))
            elif child.tag == "PML":                             # src/MxProblem.cpp:559-598
                shp = [c for c in child if c.tag in SHAPE_TAGS]
                if len(shp) != 1:
                    raise ValueError("PML object needs to contain exactly 1 Shape object")
                self.pmls.append(yee.Pml(make_shape(shp[0]), vec(get_attr("profile begin", child)), vec(get_attr("profile end", child)),
                                         alpha_max=float(get_attr("max alpha", child, "1.0")),
                                         exponent=float(get_attr("exponent", child, "2.0")),
                                         name=get_attr("name", child, "PML%d" % len(self.pmls))))
                self.params["is complex"] = True
            elif child.tag == "Eigensolver":
                eig = child
        self._eigensolver(eig)
        if any(p != 0.0 for p in self.phase):
            self.params["is complex"] = True

    def _eigensolver(self, node):
        for name, dflt, typ in EIG_PARAMS:
            self.params["eigensolver : " + name] = _convert(get_attr(name, node, dflt), typ)
        if not self.params["eigensolver : invert"]:
            return
        lin = node.find("LinearSolver") if node is not None else None
        for name, dflt, typ in LIN_PARAMS:
            self.params["linear solver : " + name] = _convert(get_attr(name, lin, dflt), typ)
        prec = lin.find("Preconditioner") if lin is not None else None
        if prec is not None:
            ptype = get_attr("type", prec)
            self.params["linear solver : prec type"] = ptype
            table = {"amg": AMG_PARAMS, "ilu": ILU_PARAMS}[ptype]
            for name, dflt, typ in table:
                self.params["linear solver : %s : %s" % (ptype, name)] = _convert(get_attr(name, prec, dflt), typ)

    def sim(self, strict=True):
        return YeeSim(self.n, origin=self.origin, size=self.size, lower=self.lower, upper=self.upper,
                      pec=self.pec, phase_shifts=self.phase, strict=strict, dm_frac=self.dm_frac,
                      dielectrics=self.dielectrics, mus=self.mus, pmls=self.pmls)


def load(path):
    with open(path) as f:
        return Deck(f.read())
