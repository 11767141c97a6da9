// See mx_geometry.hpp for the reference lines each routine follows.
#include "mx_geometry.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace mx {

Mat3 matmul(const Mat3& x, const Mat3& y) {
  Mat3 r{};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += x.a[i][k] * y.a[k][j];
      r.a[i][j] = s;
    }
  return r;
}

Mat3 inverse(const Mat3& m) {
  const double (*a)[3] = m.a;
  const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) -
                     a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                     a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
  Mat3 r{};
  r.a[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) / det;
  r.a[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det;
  r.a[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
  r.a[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) / det;
  r.a[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det;
  r.a[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
  r.a[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) / det;
  r.a[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det;
  r.a[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
  return r;
}

Shape::Shape() : sign_(1.0), A_(Mat3::eye()), Ainv_(Mat3::eye()) {}

void Shape::refreshPlain() {
  bool id = Ainvb_[0] == 0.0 && Ainvb_[1] == 0.0 && Ainvb_[2] == 0.0;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) id = id && Ainv_.a[i][j] == (i == j ? 1.0 : 0.0);
  plain_ = id;
}

void Shape::translate(const Vec3& v) {  // src/MxShape.cpp:183-189
  b_ = b_ + v;
  Ainvb_ = Ainv_.mul(b_);
  refreshPlain();
}

void Shape::rotate(const Vec3& axis, double angle, const Vec3& pivot) {  // src/MxShape.cpp:98-127
  const double nn = norm(axis);
  const Vec3 a(axis[0] / nn, axis[1] / nn, axis[2] / nn);
  Mat3 M{};
  M.a[0][1] = -a[2]; M.a[1][0] = a[2];
  M.a[0][2] = a[1];  M.a[2][0] = -a[1];
  M.a[1][2] = -a[0]; M.a[2][1] = a[0];
  // the reference fills M(0,1) = M(1,0) = a2 and negates M(1,0), etc.: M = -[a]_x
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M.a[i][j] = -M.a[i][j];
  const Mat3 MM = matmul(M, M);
  Mat3 R = Mat3::eye();
  const double s = std::sin(angle), c1 = 1.0 - std::cos(angle);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R.a[i][j] += M.a[i][j] * s + MM.a[i][j] * c1;
  A_ = matmul(R, A_);
  b_ = R.mul(b_ - pivot) + pivot;
  Ainv_ = inverse(A_);
  Ainvb_ = Ainv_.mul(b_);
  refreshPlain();
}

void Shape::scale(const Vec3& mags, const Vec3& origin) {  // src/MxShape.cpp:163-172
  Mat3 S{};
  for (int i = 0; i < 3; ++i) S.a[i][i] = mags[i];
  A_ = matmul(S, A_);
  b_ = S.mul(b_ - origin) + origin;
  Ainv_ = inverse(A_);
  Ainvb_ = Ainv_.mul(b_);
  refreshPlain();
}

void Shape::reflect(const Vec3& normal, const Vec3& p) {  // src/MxShape.cpp:200-216
  const double nn = norm(normal);
  const Vec3 n(normal[0] / nn, normal[1] / nn, normal[2] / nn);
  Mat3 M = Mat3::eye();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M.a[i][j] -= 2.0 * n[i] * n[j];
  A_ = matmul(M, A_);
  b_ = M.mul(b_ - p) + p;
  Ainv_ = inverse(A_);
  Ainvb_ = Ainv_.mul(b_);
  refreshPlain();
}

HalfSpace::HalfSpace(const Vec3& point, const Vec3& normal) {
  const double nn = norm(normal);
  n_ = Vec3(normal[0] / nn, normal[1] / nn, normal[2] / nn);
  translate(point);
}

Cylinder::Cylinder(double radius, const Vec3& axis, const Vec3& location) : r2_(radius * radius) {
  const double nn = norm(axis);
  const Vec3 a(axis[0] / nn, axis[1] / nn, axis[2] / nn);
  P_ = Mat3::eye();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) P_.a[i][j] -= a[i] * a[j];
  translate(location);
}

Sphere::Sphere(double radius, const Vec3& location) : r2_(radius * radius) { translate(location); }

double ShapeIntersection::localFunc(const Vec3& q) const {  // src/MxShapeIntersection.hpp:122-136
  double fmin = std::numeric_limits<double>::max();
  for (const auto& s : shapes_) {
    const double f = s->func(q);
    if (f < fmin) fmin = f;
  }
  return fmin;
}

Vec3 ShapeIntersection::localGrad(const Vec3& q) const {  // :186-206
  double fmin = std::numeric_limits<double>::max();
  const Shape* pick = nullptr;
  for (const auto& s : shapes_) {
    const double f = s->func(q);
    if (f < fmin) { fmin = f; pick = s.get(); }
  }
  return pick->gradFunc(q);
}

Slab::Slab(double l, const Vec3& normal, const Vec3& mid) {  // src/MxSlab.hpp:66-75,132-138
  const double nn = norm(normal);
  const Vec3 n(normal[0] / nn, normal[1] / nn, normal[2] / nn);
  slab_.add(std::make_shared<HalfSpace>((-0.5 * l) * n, n));
  slab_.add(std::make_shared<HalfSpace>((0.5 * l) * n, -1.0 * n));
  translate(mid);
}


static Mat3 complementProjector(const Vec3& a) {
  Mat3 P = Mat3::eye();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) P.a[i][j] -= a[i] * a[j];
  return P;
}
static Vec3 unit(const Vec3& v) {
  const double nn = norm(v);
  return Vec3(v[0] / nn, v[1] / nn, v[2] / nn);
}

Ellipsoid::Ellipsoid(const Vec3& axes, const Vec3& location)
    : ia_(1.0 / (axes[0] * axes[0]), 1.0 / (axes[1] * axes[1]), 1.0 / (axes[2] * axes[2])) { translate(location); }

Torus::Torus(double R, double r, const Vec3& axis, const Vec3& location) : R_(R), r2_(r * r), a_(unit(axis)) {
  P_ = complementProjector(a_);
  translate(location);
}
double Torus::localFunc(const Vec3& q) const {
  const double nrm = norm(P_.mul(q)), ax = dot(a_, q);
  return r2_ - (R_ - nrm) * (R_ - nrm) - ax * ax;
}
Vec3 Torus::localGrad(const Vec3& q) const {
  const double nrm = norm(P_.mul(q)), ax = dot(a_, q), f = 1.0 - R_ / nrm;
  double g[3];   // ((1 - R/|Pq|) * P) * q: the scalar scales the projector first (src/MxTorus.hpp:36)
  for (int i = 0; i < 3; ++i) g[i] = (f * P_.a[i][0]) * q[0] + (f * P_.a[i][1]) * q[1] + (f * P_.a[i][2]) * q[2];
  return Vec3(2.0 * (g[0] - ax * a_[0]), 2.0 * (g[1] - ax * a_[1]), 2.0 * (g[2] - ax * a_[2]));
}

Cone::Cone(double angle, const Vec3& axis, const Vec3& vertex) : tan_(std::tan(angle)), a_(unit(axis)) {
  P_ = complementProjector(a_);
  translate(vertex);
}
double Cone::localFunc(const Vec3& q) const {
  const double t = tan_ * dot(a_, q);
  return t * t - dot(q, P_.mul(q));
}
Vec3 Cone::localGrad(const Vec3& q) const {
  const Vec3 pq = P_.mul(q);
  const double c = tan_ * tan_ * dot(a_, q);
  return Vec3(2.0 * (c * a_[0] - pq[0]), 2.0 * (c * a_[1] - pq[1]), 2.0 * (c * a_[2] - pq[2]));
}

double ShapeUnion::localFunc(const Vec3& q) const {
  double fmax = -std::numeric_limits<double>::max();
  for (const auto& s : shapes_) {
    const double f = s->func(q);
    if (f > fmax) fmax = f;
  }
  return fmax;
}
Vec3 ShapeUnion::localGrad(const Vec3& q) const {
  double fmax = -std::numeric_limits<double>::max();
  const Shape* pick = nullptr;
  for (const auto& s : shapes_) {
    const double f = s->func(q);
    if (f > fmax) { fmax = f; pick = s.get(); }
  }
  return pick->gradFunc(q);
}

double ShapeSubtract::localFunc(const Vec3& q) const {
  const double fBase = base_->func(q), fRm = rm_.func(q);
  const bool inBase = fBase > 0, inRm = fRm < 0;
  if (inRm && inBase) return fRm;
  if (!inRm && !inBase) return fBase;
  return fBase < fRm ? fBase : fRm;
}
Vec3 ShapeSubtract::localGrad(const Vec3& q) const {
  const double fBase = base_->func(q), fRm = rm_.func(q);
  const bool inBase = fBase > 0, inRm = fRm < 0;
  if (inRm && inBase) return rm_.gradFunc(q);
  if (!inRm && !inBase) return base_->gradFunc(q);
  return fBase < fRm ? base_->gradFunc(q) : rm_.gradFunc(q);
}

ShapeMirror::ShapeMirror(std::shared_ptr<const Shape> shape, const Vec3& normal, const Vec3& p) : shape_(std::move(shape)) {
  const Vec3 n = unit(normal);
  plane_ = std::make_shared<HalfSpace>(p, n);
  std::shared_ptr<Shape> m = shape_->clone();
  m->reflect(n, p);
  mirror_ = m;
}

ShapeRepeat::ShapeRepeat(std::shared_ptr<const Shape> shape, const Vec3& origin, const Vec3& direction, double step,
                         int numPos, int numNeg)
    : shape_(std::move(shape)), o_(origin), dir_(unit(direction)), s_(step), np_(numPos), nn_(-numNeg) {}
Vec3 ShapeRepeat::fold(const Vec3& p) const {
  const double slabPt = dot(p - o_, dir_) / s_ + 0.5;
  const double k = slabPt >= 0.0 ? std::min(std::floor(slabPt), np_) : std::max(std::floor(slabPt), nn_);
  const double sk = s_ * k;
  return Vec3(p[0] - sk * dir_[0], p[1] - sk * dir_[1], p[2] - sk * dir_[2]);
}

Vec3 rootFind(const Shape& shape, const Vec3& p1, const Vec3& p2, double tol) {
  const Vec3 d = p2 - p1;
  const double len = norm(d);
  const double scaleTol = tol * len;
  const Vec3 dir(d[0] / len, d[1] / len, d[2] / len);
  const double x2 = len;
  const double f1 = shape.func(p1), f2 = shape.func(p2);
  if (f1 == 0) return p1;
  if (f2 == 0) return p2;
  double xl, xh;
  if (f1 < 0) { xl = 0.0; xh = x2; } else { xl = x2; xh = 0.0; }
  double root = 0.5 * (xl + xh);
  double dxold = std::fabs(xh - xl);
  double dx = dxold;
  Vec3 pt = p1 + dir * root;
  double f = shape.func(pt);
  double df = dot(dir, shape.gradFunc(pt));
  for (int i = 0; i < 500; ++i) {
    if ((((root - xh) * df - f) * ((root - xl) * df - f) >= 0) ||
        (std::fabs(2. * f) > std::fabs(dxold * df))) {
      dxold = dx;
      dx = 0.5 * (xh - xl);
      root = xl + dx;
      if (xl == root) return p1 + dir * root;
    } else {
      dxold = dx;
      dx = f / df;
      const double tmp = root;
      root -= dx;
      if (tmp == root) return p1 + dir * root;
    }
    if (std::fabs(dx) < scaleTol) return p1 + dir * root;
    pt = p1 + dir * root;
    f = shape.func(pt);
    df = dot(dir, shape.gradFunc(pt));
    if (f < 0) xl = root; else xh = root;
  }
  return p1 + dir * root;
}

double segmentFraction(const Shape& shape, const Vec3& mid, int axis, double len) {
  Vec3 dir; dir[axis] = 1.0;
  const Vec3 p1 = mid + (0.5 * len) * dir;
  const Vec3 p2 = mid - (0.5 * len) * dir;
  const int f1s = sgn(shape.func(p1)), f2s = sgn(shape.func(p2));
  if ((f1s == 1 && f2s != -1) || (f2s == 1 && f1s != -1)) return 1.0;
  if ((f1s == -1 && f2s != 1) || (f2s == -1 && f1s != 1)) return 0.0;
  if (f1s == 0 && f2s == 0) return sgn(shape.func(mid)) == 1 ? 1.0 : 0.0;
  const Vec3 p = rootFind(shape, p1, p2);
  return f1s == 1 ? norm(p - p1) / len : norm(p - p2) / len;
}

namespace {

enum State { ON, IN, OUT, CUT };

template <int NV>
State stateOf(const Shape& shape, const Vec3* verts, double* fvals, int* fs) {
  bool neg = false, pos = false;
  for (int i = 0; i < NV; ++i) {
    fvals[i] = shape.func(verts[i]);
    fs[i] = sgn(fvals[i]);
    if (fs[i] == -1) neg = true; else if (fs[i] == 1) pos = true;
  }
  if (!neg && !pos) return ON;
  if (!neg) return IN;
  if (!pos) return OUT;
  return CUT;
}

// src/MxPolytope.cpp:38-66
template <int NE>
void edgeCrossings(const Shape& shape, const Vec3* verts, const int* fs, const int (*edges)[2],
                   Vec3* edgeX, bool* has, int* first) {
  *first = -1;
  for (int e = 0; e < NE; ++e) {
    const int a = edges[e][0], b = edges[e][1];
    const int f1s = fs[a], f2s = fs[b];
    has[e] = false;
    if ((f1s == -1 && f2s == 1) || (f1s == 1 && f2s == -1)) {
      edgeX[e] = rootFind(shape, verts[a], verts[b]); has[e] = true;
    } else if (f1s == 0 && f2s == 1) {
      edgeX[e] = verts[a]; has[e] = true;
    } else if (f2s == 0 && f1s == 1) {
      edgeX[e] = verts[b]; has[e] = true;
    }
    if (has[e] && *first < 0) *first = e;
  }
}

const int RECT_EDGES[4][2] = {{0, 1}, {0, 2}, {1, 3}, {2, 3}};
const int BOX_EDGES[12][2] = {{0, 4}, {1, 5}, {2, 6}, {3, 7}, {0, 2}, {1, 3}, {4, 6}, {5, 7},
                              {0, 1}, {2, 3}, {4, 5}, {6, 7}};
const int BOX_EDGE_FACES[12][2] = {{0, 2}, {1, 2}, {0, 3}, {1, 3}, {0, 4}, {1, 4}, {0, 5}, {1, 5},
                                   {2, 4}, {3, 4}, {2, 5}, {3, 5}};

struct BoxTopo {
  int faceEdges[6][4];
  int faceVerts[6][4];
  BoxTopo() {
    for (int f = 0; f < 6; ++f) {
      int ne = 0;
      bool inface[8] = {false};
      for (int e = 0; e < 12; ++e)
        if (BOX_EDGE_FACES[e][0] == f || BOX_EDGE_FACES[e][1] == f) {
          faceEdges[f][ne++] = e;
          inface[BOX_EDGES[e][0]] = inface[BOX_EDGES[e][1]] = true;
        }
      int nv = 0;
      for (int v = 0; v < 8; ++v) if (inface[v]) faceVerts[f][nv++] = v;
    }
  }
};
const BoxTopo BOX;

}  // namespace

double rectFraction(const Shape& shape, const Vec3& p, int axis, double l1, double l2) {
  Vec3 d1, d2;
  d1[(axis + 1) % 3] = 1.0;
  d2[(axis + 2) % 3] = 1.0;
  Vec3 verts[4];
  verts[0] = p + 0.5 * ((-l1) * d1 - l2 * d2);
  verts[1] = p + 0.5 * (l1 * d1 - l2 * d2);
  verts[2] = p + 0.5 * ((-l1) * d1 + l2 * d2);
  verts[3] = p + 0.5 * (l1 * d1 + l2 * d2);
  double fvals[4]; int fs[4];
  switch (stateOf<4>(shape, verts, fvals, fs)) {
    case ON: return sgn(shape.func(p)) == 1 ? 1.0 : 0.0;
    case IN: return 1.0;
    case OUT: return 0.0;
    default: break;
  }
  Vec3 edgeX[4]; bool has[4]; int first;
  edgeCrossings<4>(shape, verts, fs, RECT_EDGES, edgeX, has, &first);
  const Vec3 v0 = edgeX[first];
  double area = 0.0;
  for (int e = 0; e < 4; ++e) {
    const int a = RECT_EDGES[e][0], b = RECT_EDGES[e][1];
    Vec3 v1, v2;
    if (has[e]) {
      v1 = fs[a] != -1 ? verts[a] : edgeX[e];
      v2 = fs[b] != -1 ? verts[b] : edgeX[e];
    } else if (fs[a] == 1 || fs[b] == 1) {
      v1 = verts[a]; v2 = verts[b];
    } else {
      continue;
    }
    area += 0.5 * norm(cross(v1 - v0, v2 - v0));
  }
  return area / (l1 * l2);
}

double boxFraction(const Shape& shape, const Vec3& p, double lx, double ly, double lz) {
  Vec3 verts[8];
  int k = 0;
  for (int sx = -1; sx <= 1; sx += 2)
    for (int sy = -1; sy <= 1; sy += 2)
      for (int sz = -1; sz <= 1; sz += 2)
        verts[k++] = Vec3(p[0] + (0.5 * sx) * lx, p[1] + (0.5 * sy) * ly, p[2] + (0.5 * sz) * lz);
  double fvals[8]; int fs[8];
  switch (stateOf<8>(shape, verts, fvals, fs)) {
    case ON: return sgn(shape.func(p)) == 1 ? 1.0 : 0.0;
    case IN: return 1.0;
    case OUT: return 0.0;
    default: break;
  }
  Vec3 edgeX[12]; bool has[12]; int first;
  edgeCrossings<12>(shape, verts, fs, BOX_EDGES, edgeX, has, &first);
  bool faceOn[6] = {false};
  Vec3 faceArea[6], faceVert[6];
  double faceCutLen[6] = {0, 0, 0, 0, 0, 0};
  auto accumulate = [](Vec3& area, const Vec3& tri) {
    if (dot(area, tri) > 0) area = area + tri; else area = area - tri;
  };
  for (int f = 0; f < 6; ++f) {
    bool inside = false;
    for (int i = 0; i < 4; ++i) if (fs[BOX.faceVerts[f][i]] == 1) { inside = true; break; }
    if (!inside) continue;
    int v0e = -1;
    for (int i = 0; i < 4; ++i) if (has[BOX.faceEdges[f][i]]) { v0e = BOX.faceEdges[f][i]; break; }
    Vec3 area;
    if (v0e >= 0) {
      const Vec3 v0 = edgeX[v0e];
      faceVert[f] = v0;
      for (int i = 0; i < 4; ++i) {
        const int e = BOX.faceEdges[f][i];
        const int a = BOX_EDGES[e][0], b = BOX_EDGES[e][1];
        Vec3 v1, v2;
        if (has[e] && e != v0e) {
          v1 = edgeX[e];
          v2 = fs[a] != -1 ? verts[a] : verts[b];
          faceCutLen[f] = norm(v0 - v1);
        } else if (fs[a] == 1 || fs[b] == 1) {
          v1 = verts[a]; v2 = verts[b];
        } else {
          continue;
        }
        accumulate(area, cross(v1 - v0, v2 - v0));
      }
    } else {
      const int base = BOX.faceVerts[f][0];
      const Vec3 v0 = verts[base];
      faceVert[f] = v0;
      for (int i = 0; i < 4; ++i) {
        const int e = BOX.faceEdges[f][i];
        const int a = BOX_EDGES[e][0], b = BOX_EDGES[e][1];
        if (a != base && b != base) accumulate(area, cross(verts[a] - v0, verts[b] - v0));
      }
    }
    faceArea[f] = 0.5 * area;
    faceOn[f] = true;
  }
  double vol = 0.0, wtSum = 0.0;
  for (int e = 0; e < 12; ++e) {
    if (!has[e]) continue;
    const double wt = faceCutLen[BOX_EDGE_FACES[e][0]] + faceCutLen[BOX_EDGE_FACES[e][1]];
    wtSum += wt;
    for (int f = 0; f < 6; ++f)
      if (faceOn[f]) vol += wt * std::fabs(dot(faceVert[f] - edgeX[e], faceArea[f])) / 3.;
  }
  vol /= wtSum;
  return vol / (lx * ly * lz);
}

}  // namespace mx
