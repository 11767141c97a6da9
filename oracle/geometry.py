"""Oracle: level-set shapes and cut-cell fractions (TEST INFRASTRUCTURE ONLY).

numpy / pure-Python restatement of the reference geometry rules that decide the
Dey-Mittra fractions (and therefore the assembled pencil):

  shapes      src/MxShape.hpp:145-172 (func = sign * _func(Ainv p - Ainv b)),
              src/MxCylinder.hpp:34-40, src/MxHalfSpace.hpp:31-33,
              src/MxSlab.hpp:95-100,140-146, src/MxSphere.hpp:31-33,
              src/MxShapeIntersection.hpp:122-136,186-206, src/MxEllipsoid.hpp, src/MxTorus.hpp,
              src/MxCone.hpp, src/MxShapeUnion.hpp, src/MxShapeSubtract.hpp, src/MxShapeMirror.hpp,
              src/MxShapeRepeat.hpp
  sign()      src/MxUtil.hpp:26,44-49 (dEps = 0)
  rootFind    src/MxUtil.hpp:297-362 (safeguarded Newton, maxiter forced 500)
  segment     src/MxSegment.cpp:21-42
  polygon     src/MxPolytope.cpp:8-80, src/MxConvexPolygon.cpp:24-76,165-252,256-291,
              src/MxCartRect.hpp:37-91
  polyhedron  src/MxConvexPolyhedron.cpp:66-346, src/MxCartBox.cpp:22-62,
              src/MxCartBox.hpp:33-47

Every shape has a vectorised numpy evaluator (func/grad, used to classify all
entities at once) and a scalar twin on Python float triples (func1/grad1, same
arithmetic, used inside the per-entity cut loops).
"""
import math

import numpy as np


# ----------------------------------------------------------------------------
# shapes
# ----------------------------------------------------------------------------
class Shape:
    """Affine-transformed level set; func(p) > 0 inside (src/MxShape.hpp:145-148)."""

    def __init__(self):
        self.sign = 1.0
        self.Ainv = np.eye(3)
        self.b = np.zeros(3)
        self._sync()

    def _sync(self):
        self.Ainvb = np.array(self._sync_exact())
        self._a = self.Ainv.tolist()
        self._ab = self.Ainvb.tolist()

    def translate(self, v):
        """src/MxShape.cpp:183-189."""
        self.b = self.b + np.asarray(v, dtype=float)
        self._sync()
        return self

    def invert(self):
        self.sign *= -1.0
        return self

    # generic transforms (src/MxShape.cpp:98-216): A <- T A, b <- T (b - p) + p, Ainv = A^-1.
    # Plain Python floats in a fixed summation order, so the C++ host code can match bit for bit.
    def _apply(self, T, p):
        A = getattr(self, "A", None)
        if A is None:
            A = [[1.0 if i == j else 0.0 for j in range(3)] for i in range(3)]
        self.A = _matmul3(T, A)
        b = [float(x) for x in self.b]
        d = [b[i] - p[i] for i in range(3)]
        self.b = np.array([_row3(T[i], d) + p[i] for i in range(3)])
        self.Ainv = np.array(_inverse3(self.A))
        self._sync()
        return self

    def _sync_exact(self):
        a = self.Ainv.tolist()
        b = [float(x) for x in self.b]
        return [_row3(a[i], b) for i in range(3)]

    def rotate(self, axis, angle, pivot=None):
        """src/MxShape.cpp:98-127 (Rodrigues); without a pivot: about the current translation."""
        import math
        nn = math.sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2])
        a = [axis[0] / nn, axis[1] / nn, axis[2] / nn]
        M = [[-0.0, a[2], -a[1]], [-a[2], -0.0, a[0]], [a[1], -a[0], -0.0]]
        MM = _matmul3(M, M)
        s, c1 = math.sin(angle), 1.0 - math.cos(angle)
        R = [[(1.0 if i == j else 0.0) + (M[i][j] * s + MM[i][j] * c1) for j in range(3)] for i in range(3)]
        p = [float(x) for x in (self.b if pivot is None else pivot)]
        return self._apply(R, p)

    def scale(self, magnitudes, origin=(0.0, 0.0, 0.0)):
        """src/MxShape.cpp:163-172."""
        S = [[float(magnitudes[i]) if i == j else 0.0 for j in range(3)] for i in range(3)]
        return self._apply(S, [float(x) for x in origin])

    def reflect(self, normal, point):
        """src/MxShape.cpp:200-216."""
        import math
        nn = math.sqrt(normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2])
        n = [normal[0] / nn, normal[1] / nn, normal[2] / nn]
        M = [[(1.0 if i == j else 0.0) - 2.0 * n[i] * n[j] for j in range(3)] for i in range(3)]
        return self._apply(M, [float(x) for x in point])

    # vectorised
    def _local(self, p):
        p = np.asarray(p, dtype=float)
        a, b = self._a, self._ab
        return np.stack([a[i][0] * p[..., 0] + a[i][1] * p[..., 1] + a[i][2] * p[..., 2] - b[i]
                         for i in range(3)], axis=-1)

    def func(self, p):
        return self.sign * self._func(self._local(p))

    # scalar
    def _local1(self, p):
        a, b = self._a, self._ab
        return (a[0][0] * p[0] + a[0][1] * p[1] + a[0][2] * p[2] - b[0],
                a[1][0] * p[0] + a[1][1] * p[1] + a[1][2] * p[2] - b[1],
                a[2][0] * p[0] + a[2][1] * p[1] + a[2][2] * p[2] - b[2])

    def func1(self, p):
        return self.sign * self._func1(self._local1(p))

    def grad1(self, p):
        """src/MxShape.hpp:163-167: sign * (_gradFunc(q) * Ainv) (row vector times matrix)."""
        g = self._grad1(self._local1(p))
        a, s = self._a, self.sign
        return (s * (g[0] * a[0][0] + g[1] * a[1][0] + g[2] * a[2][0]),
                s * (g[0] * a[0][1] + g[1] * a[1][1] + g[2] * a[2][1]),
                s * (g[0] * a[0][2] + g[1] * a[1][2] + g[2] * a[2][2]))


class HalfSpace(Shape):
    """src/MxHalfSpace.hpp:14-33: _func(q) = n.q, plane through `point`."""

    def __init__(self, point, normal):
        super().__init__()
        n = np.asarray(normal, dtype=float)
        self.n = (n / _norm3(n)).tolist()
        self.translate(point)

    def _func(self, q):
        n = self.n
        return n[0] * q[..., 0] + n[1] * q[..., 1] + n[2] * q[..., 2]

    def _func1(self, q):
        n = self.n
        return n[0] * q[0] + n[1] * q[1] + n[2] * q[2]

    def _grad1(self, q):
        return tuple(self.n)


class Cylinder(Shape):
    """src/MxCylinder.hpp:34-40,58-72: _func(q) = r^2 - q.(P q), P = I - a a^T."""

    def __init__(self, radius, axis, location):
        super().__init__()
        a = np.asarray(axis, dtype=float)
        a = a / _norm3(a)
        self.r2 = radius * radius
        self.P = (np.eye(3) - np.outer(a, a)).tolist()
        self.translate(location)

    def _Pq(self, q0, q1, q2):
        P = self.P
        return (P[0][0] * q0 + P[0][1] * q1 + P[0][2] * q2,
                P[1][0] * q0 + P[1][1] * q1 + P[1][2] * q2,
                P[2][0] * q0 + P[2][1] * q1 + P[2][2] * q2)

    def _func(self, q):
        q0, q1, q2 = q[..., 0], q[..., 1], q[..., 2]
        p = self._Pq(q0, q1, q2)
        return self.r2 - (q0 * p[0] + q1 * p[1] + q2 * p[2])

    def _func1(self, q):
        p = self._Pq(q[0], q[1], q[2])
        return self.r2 - (q[0] * p[0] + q[1] * p[1] + q[2] * p[2])

    def _grad1(self, q):
        p = self._Pq(q[0], q[1], q[2])
        return (-2.0 * p[0], -2.0 * p[1], -2.0 * p[2])


class Sphere(Shape):
    """src/MxSphere.hpp:31-33: _func(q) = 1 - q.q / r^2."""

    def __init__(self, radius, location):
        super().__init__()
        self.r2 = radius * radius
        self.translate(location)

    def _func(self, q):
        return 1.0 - (q[..., 0] * q[..., 0] + q[..., 1] * q[..., 1] + q[..., 2] * q[..., 2]) / self.r2

    def _func1(self, q):
        return 1.0 - (q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) / self.r2

    def _grad1(self, q):
        return (-2.0 * q[0] / self.r2, -2.0 * q[1] / self.r2, -2.0 * q[2] / self.r2)


class Intersection(Shape):
    """src/MxShapeIntersection.hpp:122-136 (min of members), 186-206 (gradient of
    the first member attaining the min, strict '<')."""

    def __init__(self, shapes):
        super().__init__()
        self.shapes = list(shapes)

    def _func(self, q):
        f = self.shapes[0].func(q)
        for s in self.shapes[1:]:
            f = np.minimum(f, s.func(q))
        return f

    def _func1(self, q):
        fmin = float("inf")
        for s in self.shapes:
            f = s.func1(q)
            if f < fmin:
                fmin = f
        return fmin

    def _grad1(self, q):
        fmin = float("inf")
        pick = None
        for s in self.shapes:
            f = s.func1(q)
            if f < fmin:
                fmin = f
                pick = s
        return pick.grad1(q)


class Slab(Shape):
    """src/MxSlab.hpp:66-75,132-138: intersection of two half spaces at -/+ l/2."""

    def __init__(self, thickness, normal, midpoint):
        super().__init__()
        n = np.asarray(normal, dtype=float)
        n = n / _norm3(n)
        self.slab = Intersection([HalfSpace(-0.5 * thickness * n, n),
                                  HalfSpace(0.5 * thickness * n, -n)])
        self.translate(midpoint)

    def _func(self, q):
        return self.slab.func(q)

    def _func1(self, q):
        return self.slab.func1(q)

    def _grad1(self, q):
        return self.slab.grad1(q)


class Ellipsoid(Shape):
    """src/MxEllipsoid.hpp:33-35,46-48: _func(q) = 1 - sum q_i^2 / a_i^2."""

    def __init__(self, axes, location):
        super().__init__()
        self.ia = [1.0 / (float(a) * float(a)) for a in axes]
        self.translate(location)

    def _func(self, q):
        ia = self.ia
        return 1.0 - (q[..., 0] * (ia[0] * q[..., 0]) + q[..., 1] * (ia[1] * q[..., 1]) + q[..., 2] * (ia[2] * q[..., 2]))

    def _func1(self, q):
        ia = self.ia
        return 1.0 - (q[0] * (ia[0] * q[0]) + q[1] * (ia[1] * q[1]) + q[2] * (ia[2] * q[2]))

    def _grad1(self, q):
        ia = self.ia
        return ((-2.0 * ia[0]) * q[0], (-2.0 * ia[1]) * q[1], (-2.0 * ia[2]) * q[2])


class _Axial(Shape):
    """Shared by Torus and Cone: unit axis a and the projector P = I - a a^T."""

    def _set_axis(self, axis):
        nn = _norm3(axis)
        self.a = [float(axis[0]) / nn, float(axis[1]) / nn, float(axis[2]) / nn]
        a = self.a
        self.P = [[(1.0 if i == j else 0.0) - a[i] * a[j] for j in range(3)] for i in range(3)]

    def _Pq(self, q0, q1, q2):
        P = self.P
        return (P[0][0] * q0 + P[0][1] * q1 + P[0][2] * q2,
                P[1][0] * q0 + P[1][1] * q1 + P[1][2] * q2,
                P[2][0] * q0 + P[2][1] * q1 + P[2][2] * q2)

    def _aq(self, q0, q1, q2):
        a = self.a
        return a[0] * q0 + a[1] * q1 + a[2] * q2


class Torus(_Axial):
    """src/MxTorus.hpp:31-37: _func(q) = r^2 - (R - |P q|)^2 - (a.q)^2."""

    def __init__(self, major_radius, minor_radius, axis, location):
        super().__init__()
        self.R = float(major_radius)
        self.r2 = float(minor_radius) * float(minor_radius)
        self._set_axis(axis)
        self.translate(location)

    def _func(self, q):
        q0, q1, q2 = q[..., 0], q[..., 1], q[..., 2]
        p = self._Pq(q0, q1, q2)
        nrm = np.sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2])
        ax = self._aq(q0, q1, q2)
        return self.r2 - (self.R - nrm) * (self.R - nrm) - ax * ax

    def _func1(self, q):
        p = self._Pq(q[0], q[1], q[2])
        nrm = math.sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2])
        ax = self._aq(q[0], q[1], q[2])
        return self.r2 - (self.R - nrm) * (self.R - nrm) - ax * ax

    def _grad1(self, q):
        p = self._Pq(q[0], q[1], q[2])
        nrm = math.sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2])
        ax = self._aq(q[0], q[1], q[2])
        f = 1.0 - (self.R / nrm if nrm != 0.0 else float("inf"))
        a, P = self.a, self.P
        # ((1 - R/|Pq|) * P) * q: the scalar scales the projector first (src/MxTorus.hpp:36)
        g = [(f * P[i][0]) * q[0] + (f * P[i][1]) * q[1] + (f * P[i][2]) * q[2] for i in range(3)]
        return (2.0 * (g[0] - ax * a[0]), 2.0 * (g[1] - ax * a[1]), 2.0 * (g[2] - ax * a[2]))


class Cone(_Axial):
    """src/MxCone.hpp:31-37: _func(q) = (tan(theta) a.q)^2 - q.(P q); a double cone with its vertex at `vertex`."""

    def __init__(self, angle, axis, vertex):
        super().__init__()
        self.tan = math.tan(float(angle))
        self._set_axis(axis)
        self.translate(vertex)

    def _func(self, q):
        q0, q1, q2 = q[..., 0], q[..., 1], q[..., 2]
        p = self._Pq(q0, q1, q2)
        t = self.tan * self._aq(q0, q1, q2)
        return t * t - (q0 * p[0] + q1 * p[1] + q2 * p[2])

    def _func1(self, q):
        p = self._Pq(q[0], q[1], q[2])
        t = self.tan * self._aq(q[0], q[1], q[2])
        return t * t - (q[0] * p[0] + q[1] * p[1] + q[2] * p[2])

    def _grad1(self, q):
        p = self._Pq(q[0], q[1], q[2])
        c = self.tan * self.tan * self._aq(q[0], q[1], q[2])
        a = self.a
        return (2.0 * (c * a[0] - p[0]), 2.0 * (c * a[1] - p[1]), 2.0 * (c * a[2] - p[2]))


class Union(Shape):
    """src/MxShapeUnion.hpp:68-83 (max of members), 136-154 (gradient of the first member attaining the max)."""

    def __init__(self, shapes=()):
        super().__init__()
        self.shapes = list(shapes)

    def _func(self, q):
        f = self.shapes[0].func(q)
        for s in self.shapes[1:]:
            f = np.maximum(f, s.func(q))
        return f

    def _func1(self, q):
        fmax = -float("inf")
        for s in self.shapes:
            f = s.func1(q)
            if f > fmax:
                fmax = f
        return fmax

    def _grad1(self, q):
        fmax = -float("inf")
        pick = None
        for s in self.shapes:
            f = s.func1(q)
            if f > fmax:
                fmax = f
                pick = s
        return pick.grad1(q)


class Subtract(Shape):
    """src/MxShapeSubtract.hpp:73-90,157-175; the removed shapes form an inverted union."""

    def __init__(self, base, removed):
        super().__init__()
        self.base = base
        self.rm = Union(removed).invert()

    def _func(self, q):
        fb, fr = self.base.func(q), self.rm.func(q)
        in_b, in_r = fb > 0, fr < 0
        return np.where(in_r & in_b, fr, np.where(~in_r & ~in_b, fb, np.where(fb < fr, fb, fr)))

    def _pick(self, q):
        fb, fr = self.base.func1(q), self.rm.func1(q)
        in_b, in_r = fb > 0, fr < 0
        if in_r and in_b:
            return fr, self.rm
        if not in_r and not in_b:
            return fb, self.base
        return (fb, self.base) if fb < fr else (fr, self.rm)

    def _func1(self, q):
        return self._pick(q)[0]

    def _grad1(self, q):
        return self._pick(q)[1].grad1(q)


class Mirror(Shape):
    """src/MxShapeMirror.hpp:86-116,156-170: the shape where the half space (point, normal) is positive, its reflection elsewhere."""

    def __init__(self, shape, normal, point=(0.0, 0.0, 0.0)):
        import copy
        super().__init__()
        nn = _norm3(normal)
        n = [float(normal[0]) / nn, float(normal[1]) / nn, float(normal[2]) / nn]
        p = [float(x) for x in point]
        self.shape = shape
        self.plane = HalfSpace(p, n)
        self.mirror = copy.deepcopy(shape).reflect(n, p)

    def _func(self, q):
        return np.where(self.plane.func(q) > 0.0, self.shape.func(q), self.mirror.func(q))

    def _func1(self, q):
        return self.shape.func1(q) if self.plane.func1(q) > 0.0 else self.mirror.func1(q)

    def _grad1(self, q):
        return self.shape.grad1(q) if self.plane.func1(q) > 0.0 else self.mirror.grad1(q)


class Repeat(Shape):
    """src/MxShapeRepeat.hpp:86-120,140-150: copies at origin + i * step * direction for i = -num_neg .. num_pos."""

    def __init__(self, shape, origin, direction, step, num_pos, num_neg):
        super().__init__()
        nn = _norm3(direction)
        self.dir = [float(direction[0]) / nn, float(direction[1]) / nn, float(direction[2]) / nn]
        self.o = [float(x) for x in origin]
        self.s, self.np_, self.nn_ = float(step), float(int(num_pos)), float(-int(num_neg))
        self.shape = shape

    def _fold1(self, p):
        o, d = self.o, self.dir
        slab = ((p[0] - o[0]) * d[0] + (p[1] - o[1]) * d[1] + (p[2] - o[2]) * d[2]) / self.s + 0.5
        k = min(math.floor(slab), self.np_) if slab >= 0.0 else max(math.floor(slab), self.nn_)
        sk = self.s * float(k)
        return (p[0] - sk * d[0], p[1] - sk * d[1], p[2] - sk * d[2])

    def _func(self, q):
        o, d = self.o, self.dir
        q0, q1, q2 = q[..., 0], q[..., 1], q[..., 2]
        slab = ((q0 - o[0]) * d[0] + (q1 - o[1]) * d[1] + (q2 - o[2]) * d[2]) / self.s + 0.5
        fl = np.floor(slab)
        k = np.where(slab >= 0.0, np.minimum(fl, self.np_), np.maximum(fl, self.nn_))
        sk = self.s * k
        return self.shape.func(np.stack([q0 - sk * d[0], q1 - sk * d[1], q2 - sk * d[2]], axis=-1))

    def _func1(self, q):
        return self.shape.func1(self._fold1(q))

    def _grad1(self, q):
        return self.shape.grad1(self._fold1(q))


def pillbox(radius=0.4, length=0.8):
    """example/pillbox.py:14-16: z-axis cylinder intersected with a z slab."""
    return Intersection([Cylinder(radius, [0, 0, 1], [0, 0, 0]),
                         Slab(length, [0, 0, 1], [0, 0, 0])])


# ----------------------------------------------------------------------------
# scalar helpers (Python float triples)
# ----------------------------------------------------------------------------
def _norm3(v):
    """sqrt(v0^2 + v1^2 + v2^2) in this order (np.linalg.norm scales and may differ in the last bit)."""
    import math
    return math.sqrt(float(v[0]) * float(v[0]) + float(v[1]) * float(v[1]) + float(v[2]) * float(v[2]))


def _row3(r, v):
    return r[0] * v[0] + r[1] * v[1] + r[2] * v[2]


def _matmul3(x, y):
    out = [[0.0] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(3):
            acc = 0.0
            for k in range(3):
                acc += x[i][k] * y[k][j]
            out[i][j] = acc
    return out


def _inverse3(a):
    """Cofactor inverse, same expression order as the host code (mx_geometry.cpp)."""
    det = (a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) -
           a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
           a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]))
    return [[(a[1][1] * a[2][2] - a[1][2] * a[2][1]) / det,
             (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det,
             (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det],
            [(a[1][2] * a[2][0] - a[1][0] * a[2][2]) / det,
             (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det,
             (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det],
            [(a[1][0] * a[2][1] - a[1][1] * a[2][0]) / det,
             (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det,
             (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det]]


def sgn(v):
    """src/MxUtil.hpp:44-49 with dEps = 0 (vectorised)."""
    return np.sign(v).astype(int)


def _sgn1(v):
    return -1 if v < 0.0 else (0 if v == 0.0 else 1)


def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def _norm(v):
    return math.sqrt(_dot(v, v))


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _axpy(p, d, t):
    return (p[0] + d[0] * t, p[1] + d[1] * t, p[2] + d[2] * t)


def root_find(shape, p1, p2, tol=1.e-12):
    """src/MxUtil.hpp:297-362.  maxiter is overwritten with 500 (:299)."""
    d = _sub(p2, p1)
    length = _norm(d)
    scale_tol = tol * length
    direc = (d[0] / length, d[1] / length, d[2] / length)
    x2 = length
    f1 = shape.func1(p1)
    f2 = shape.func1(p2)
    if f1 == 0:
        return p1
    if f2 == 0:
        return p2
    x1 = 0.0
    if f1 < 0:
        xl, xh = x1, x2
    else:
        xl, xh = x2, x1
    root = 0.5 * (xl + xh)
    dxold = abs(xh - xl)
    dx = dxold
    pt = _axpy(p1, direc, root)
    f = shape.func1(pt)
    df = _dot(direc, shape.grad1(pt))
    for _ in range(500):
        if ((((root - xh) * df - f) * ((root - xl) * df - f) >= 0) or
                (abs(2. * f) > abs(dxold * df))):
            dxold = dx
            dx = 0.5 * (xh - xl)
            root = xl + dx
            if xl == root:
                return _axpy(p1, direc, root)
        else:
            dxold = dx
            dx = f / df
            tmp = root
            root -= dx
            if tmp == root:
                return _axpy(p1, direc, root)
        if abs(dx) < scale_tol:
            return _axpy(p1, direc, root)
        pt = _axpy(p1, direc, root)
        f = shape.func1(pt)
        df = _dot(direc, shape.grad1(pt))
        if f < 0:
            xl = root
        else:
            xh = root
    return _axpy(p1, direc, root)


# ----------------------------------------------------------------------------
# segment (E edges) -- src/MxSegment.cpp:21-42
# ----------------------------------------------------------------------------
def segment_fractions(shape, mids, axis, length):
    """Length fraction of Cartesian segments centred at `mids` (vectorised classify,
    scalar loop over the bracketing ones)."""
    mids = np.asarray(mids, dtype=float)
    direc = np.zeros(3)
    direc[axis] = 1.0
    p1 = mids + 0.5 * length * direc
    p2 = mids - 0.5 * length * direc
    f1s = sgn(shape.func(p1))
    f2s = sgn(shape.func(p2))
    out = np.zeros(len(mids))
    one = ((f1s == 1) & (f2s != -1)) | ((f2s == 1) & (f1s != -1))
    zero = ~one & (((f1s == -1) & (f2s != 1)) | ((f2s == -1) & (f1s != 1)))
    both_on = ~one & ~zero & (f1s == 0) & (f2s == 0)
    cut = ~one & ~zero & ~both_on
    out[one] = 1.0
    if both_on.any():
        out[both_on] = (sgn(shape.func(mids[both_on])) == 1).astype(float)
    for i in np.nonzero(cut)[0]:
        a = tuple(p1[i].tolist())
        b = tuple(p2[i].tolist())
        p = root_find(shape, a, b)
        if f1s[i] == 1:
            out[i] = _norm(_sub(p, a)) / length
        else:
            out[i] = _norm(_sub(p, b)) / length
    return out


# ----------------------------------------------------------------------------
# rectangle (B faces) -- src/MxCartRect.hpp, src/MxConvexPolygon.cpp
# ----------------------------------------------------------------------------
# edge -> (vert, vert), ascending vertex order (src/MxCartRect.hpp:70-74 +
# src/MxConvexPolygon.cpp:11-21)
RECT_EDGES = ((0, 1), (0, 2), (1, 3), (2, 3))


def rect_vertices(p, normal_axis, l1, l2):
    """src/MxCartRect.hpp:55-66 (d1, d2 per face type) and :80-89 (vertex order)."""
    d1 = np.zeros(3)
    d2 = np.zeros(3)
    d1[(normal_axis + 1) % 3] = 1.0
    d2[(normal_axis + 2) % 3] = 1.0
    p = np.asarray(p, dtype=float)
    v = np.empty(p.shape[:-1] + (4, 3))
    v[..., 0, :] = p + 0.5 * (-l1 * d1 - l2 * d2)
    v[..., 1, :] = p + 0.5 * (l1 * d1 - l2 * d2)
    v[..., 2, :] = p + 0.5 * (-l1 * d1 + l2 * d2)
    v[..., 3, :] = p + 0.5 * (l1 * d1 + l2 * d2)
    return v


def _edge_crossings(shape, verts, fs, edges):
    """src/MxPolytope.cpp:38-66 (getEdgeX)."""
    edge_x = [None] * len(edges)
    inds = []
    for e, (a, b) in enumerate(edges):
        f1s, f2s = fs[a], fs[b]
        if (f1s == -1 and f2s == 1) or (f1s == 1 and f2s == -1):
            edge_x[e] = root_find(shape, verts[a], verts[b])
            inds.append(e)
        elif f1s == 0 and f2s == 1:
            edge_x[e] = verts[a]
            inds.append(e)
        elif f2s == 0 and f1s == 1:
            edge_x[e] = verts[b]
            inds.append(e)
    return edge_x, inds


def _polygon_cut_fraction(shape, verts, fs, volume):
    """src/MxConvexPolygon.cpp:24-76; the 3-D getCorner (:165-252) returns the first
    crossed edge point."""
    edge_x, inds = _edge_crossings(shape, verts, fs, RECT_EDGES)
    v0 = edge_x[inds[0]]
    area = 0.0
    for e, (a, b) in enumerate(RECT_EDGES):
        w1, w2 = verts[a], verts[b]
        f1s, f2s = fs[a], fs[b]
        if edge_x[e] is not None:
            v1 = w1 if f1s != -1 else edge_x[e]
            v2 = w2 if f2s != -1 else edge_x[e]
        elif f1s == 1 or f2s == 1:
            v1, v2 = w1, w2
        else:
            continue
        area += 0.5 * _norm(_cross(_sub(v1, v0), _sub(v2, v0)))
    return area / volume


def _state(fs):
    """src/MxPolytope.cpp:8-34 -> 0 ON, 1 IN, 2 OUT, 3 CUT (vectorised over rows)."""
    has_neg = (fs == -1).any(axis=-1)
    has_pos = (fs == 1).any(axis=-1)
    state = np.full(fs.shape[:-1], 3, dtype=int)
    state[~has_neg & ~has_pos] = 0
    state[~has_neg & has_pos] = 1
    state[has_neg & ~has_pos] = 2
    return state


def _fractions(shape, centers, verts, cut_fn, volume):
    fvals = shape.func(verts)
    fs = sgn(fvals)
    state = _state(fs)
    out = np.zeros(len(centers))
    out[state == 1] = 1.0
    on = state == 0
    if on.any():
        out[on] = (sgn(shape.func(centers[on])) == 1).astype(float)
    for i in np.nonzero(state == 3)[0]:
        vl = [tuple(v) for v in verts[i].tolist()]
        out[i] = cut_fn(shape, vl, fs[i].tolist(), volume)
    return out


def rect_fractions(shape, centers, normal_axis, l1, l2):
    """Area fraction of Cartesian rectangles (src/MxConvexPolygon.cpp:256-291)."""
    centers = np.asarray(centers, dtype=float)
    verts = rect_vertices(centers, normal_axis, l1, l2)
    return _fractions(shape, centers, verts, _polygon_cut_fraction, l1 * l2)


# ----------------------------------------------------------------------------
# box (Psi cells) -- src/MxCartBox.*, src/MxConvexPolyhedron.cpp
# ----------------------------------------------------------------------------
BOX_EDGES = ((0, 4), (1, 5), (2, 6), (3, 7), (0, 2), (1, 3), (4, 6), (5, 7),
             (0, 1), (2, 3), (4, 5), (6, 7))          # src/MxCartBox.cpp:23-35
BOX_EDGE_FACES = ((0, 2), (1, 2), (0, 3), (1, 3), (0, 4), (1, 4), (0, 5), (1, 5),
                  (2, 4), (3, 4), (2, 5), (3, 5))     # src/MxCartBox.cpp:37-49
BOX_FACE_EDGES = tuple(tuple(e for e in range(12) if f in BOX_EDGE_FACES[e])
                       for f in range(6))             # src/MxConvexPolyhedron.cpp:43-52
BOX_FACE_VERTS = tuple(tuple(sorted({v for e in BOX_FACE_EDGES[f] for v in BOX_EDGES[e]}))
                       for f in range(6))             # faceVert = edgeFace^T edgeVert (:52), :54-63


def box_vertices(p, lx, ly, lz):
    """src/MxCartBox.hpp:33-47."""
    p = np.asarray(p, dtype=float)
    v = np.empty(p.shape[:-1] + (8, 3))
    k = 0
    for sx in (-0.5, 0.5):
        for sy in (-0.5, 0.5):
            for sz in (-0.5, 0.5):
                v[..., k, 0] = p[..., 0] + sx * lx
                v[..., k, 1] = p[..., 1] + sy * ly
                v[..., k, 2] = p[..., 2] + sz * lz
                k += 1
    return v


def _polyhedron_cut_fraction(shape, verts, fs, volume):
    """src/MxConvexPolyhedron.cpp:112-346: pyramids from every crossing point to the
    inside part of every face, weighted by the face cut lengths."""
    edge_x, _ = _edge_crossings(shape, verts, fs, BOX_EDGES)
    face_area = [None] * 6
    face_vert = [None] * 6
    face_cut_len = [0.0] * 6

    def accumulate(area, tri):
        if _dot(area, tri) > 0:
            return (area[0] + tri[0], area[1] + tri[1], area[2] + tri[2])
        return (area[0] - tri[0], area[1] - tri[1], area[2] - tri[2])

    for f in range(6):
        if not any(fs[v] == 1 for v in BOX_FACE_VERTS[f]):
            continue
        v0 = None
        v0_edge = None
        for e in BOX_FACE_EDGES[f]:
            if edge_x[e] is not None:
                v0 = edge_x[e]
                v0_edge = e
                break
        area = (0.0, 0.0, 0.0)
        if v0 is not None:
            face_vert[f] = v0
            for e in BOX_FACE_EDGES[f]:
                a, b = BOX_EDGES[e]
                w1, w2 = verts[a], verts[b]
                if edge_x[e] is not None and e != v0_edge:
                    v1 = edge_x[e]
                    v2 = w1 if fs[a] != -1 else w2
                    face_cut_len[f] = _norm(_sub(v0, v1))
                elif fs[a] == 1 or fs[b] == 1:
                    v1, v2 = w1, w2
                else:
                    continue
                area = accumulate(area, _cross(_sub(v1, v0), _sub(v2, v0)))
        else:
            base = BOX_FACE_VERTS[f][0]
            v0 = verts[base]
            face_vert[f] = v0
            for e in BOX_FACE_EDGES[f]:
                a, b = BOX_EDGES[e]
                if a != base and b != base:
                    area = accumulate(area, _cross(_sub(verts[a], v0), _sub(verts[b], v0)))
        face_area[f] = (0.5 * area[0], 0.5 * area[1], 0.5 * area[2])
    vol = 0.0
    wt_sum = 0.0
    for e in range(12):
        if edge_x[e] is None:
            continue
        f0, f1 = BOX_EDGE_FACES[e]
        wt = face_cut_len[f0] + face_cut_len[f1]
        wt_sum += wt
        for f in range(6):
            if face_area[f] is not None:
                vol += wt * abs(_dot(_sub(face_vert[f], edge_x[e]), face_area[f])) / 3.
    vol = vol / wt_sum if wt_sum != 0.0 else float('nan')   # C++ yields 0/0 = NaN here
    return vol / volume


def box_fractions(shape, centers, lx, ly, lz):
    """Volume fraction of Cartesian boxes (src/MxConvexPolyhedron.cpp:66-108)."""
    centers = np.asarray(centers, dtype=float)
    verts = box_vertices(centers, lx, ly, lz)
    return _fractions(shape, centers, verts, _polyhedron_cut_fraction, lx * ly * lz)
