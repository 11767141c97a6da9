// Level-set shapes and cut-cell fractions used by the Dey-Mittra operators (host side, setup).
//
// Restates the geometry rules of the reference that fix the assembled pencil:
//   shapes      src/MxShape.hpp:145-172, src/MxShape.cpp:91-205 (affine transforms),
//               src/MxCylinder.hpp:34-40, src/MxSphere.hpp:31-33, src/MxHalfSpace.hpp:31-33,
//               src/MxSlab.hpp:95-100,132-138, src/MxShapeIntersection.hpp:122-136,186-206
//   sign()      src/MxUtil.hpp:26,44-49 (dEps = 0)
//   rootFind    src/MxUtil.hpp:297-362 (maxiter overwritten with 500)
//   segment     src/MxSegment.cpp:21-42
//   polygon     src/MxPolytope.cpp:8-80, src/MxConvexPolygon.cpp:24-76,165-252,256-291,
//               src/MxCartRect.hpp:37-91
//   polyhedron  src/MxConvexPolyhedron.cpp:66-346, src/MxCartBox.cpp:22-62, src/MxCartBox.hpp:33-47
#ifndef MX_GEOMETRY_HPP
#define MX_GEOMETRY_HPP

#include <array>
#include <cmath>
#include <limits>
#include <memory>
#include <vector>

namespace mx {

struct Vec3 {
  double v[3];
  Vec3() : v{0, 0, 0} {}
  Vec3(double a, double b, double c) : v{a, b, c} {}
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vec3 operator*(const Vec3& a, double s) { return Vec3(a[0] * s, a[1] * s, a[2] * s); }
inline Vec3 operator*(double s, const Vec3& a) { return Vec3(a[0] * s, a[1] * s, a[2] * s); }
inline double dot(const Vec3& a, const Vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double norm(const Vec3& a) { return std::sqrt(dot(a, a)); }
inline Vec3 cross(const Vec3& a, const Vec3& b) {
  return Vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline int sgn(double x) { return x < 0.0 ? -1 : (x == 0.0 ? 0 : 1); }

struct Mat3 {
  double a[3][3];
  static Mat3 eye() { Mat3 m{}; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m.a[i][j] = i == j; return m; }
  Vec3 mul(const Vec3& p) const {
    return Vec3(a[0][0] * p[0] + a[0][1] * p[1] + a[0][2] * p[2],
                a[1][0] * p[0] + a[1][1] * p[1] + a[1][2] * p[2],
                a[2][0] * p[0] + a[2][1] * p[1] + a[2][2] * p[2]);
  }
  Vec3 lmul(const Vec3& g) const {  // row vector times matrix
    return Vec3(g[0] * a[0][0] + g[1] * a[1][0] + g[2] * a[2][0],
                g[0] * a[0][1] + g[1] * a[1][1] + g[2] * a[2][1],
                g[0] * a[0][2] + g[1] * a[1][2] + g[2] * a[2][2]);
  }
};
Mat3 matmul(const Mat3& x, const Mat3& y);
Mat3 inverse(const Mat3& m);

// func(p) > 0 inside; func = sign * _func(Ainv p - Ainv b)
class Shape {
 public:
  Shape();
  virtual ~Shape() {}
  // An untransformed shape (Ainv = I, Ainv b = 0 exactly: every node of a CSG tree the deck does not move) skips the
  // mat-vec: I p - 0 is p to the bit for every p (+-0 apart, which no local function tells apart) - the fractions of a
  // 256^3 lattice evaluate func 4.4e8 times.
  double func(const Vec3& p) const { return sign_ * (plain_ ? localFunc(p) : localFunc(Ainv_.mul(p) - Ainvb_)); }
  Vec3 gradFunc(const Vec3& p) const { return sign_ * Ainv_.lmul(localGrad(Ainv_.mul(p) - Ainvb_)); }
  void translate(const Vec3& v);
  void rotate(const Vec3& axis, double angle, const Vec3& pivot);
  void scale(const Vec3& magnitudes, const Vec3& origin);
  void reflect(const Vec3& normal, const Vec3& pointInPlane);
  void invert() { sign_ *= -1.0; }
  const Vec3& translation() const { return b_; }
  // deep copy including the transform (MxShape::clone); members of composite shapes are shared (immutable)
  virtual std::shared_ptr<Shape> clone() const = 0;

 protected:
  virtual double localFunc(const Vec3& q) const = 0;
  virtual Vec3 localGrad(const Vec3& q) const = 0;

 private:
  void refreshPlain();
  double sign_;
  Mat3 A_, Ainv_;
  Vec3 b_, Ainvb_;
  bool plain_ = true;
};

class HalfSpace : public Shape {
 public:
  HalfSpace(const Vec3& pointInPlane, const Vec3& normal);
  std::shared_ptr<Shape> clone() const override { return std::make_shared<HalfSpace>(*this); }
 protected:
  double localFunc(const Vec3& q) const override { return dot(n_, q); }
  Vec3 localGrad(const Vec3&) const override { return n_; }
 private:
  Vec3 n_;
};

class Cylinder : public Shape {
 public:
  Cylinder(double radius, const Vec3& axis, const Vec3& location);
  std::shared_ptr<Shape> clone() const override { return std::make_shared<Cylinder>(*this); }
 protected:
  double localFunc(const Vec3& q) const override { return r2_ - dot(q, P_.mul(q)); }
  Vec3 localGrad(const Vec3& q) const override { return -2.0 * P_.mul(q); }
 private:
  double r2_;
  Mat3 P_;
};

class Sphere : public Shape {
 public:
  Sphere(double radius, const Vec3& location);
  std::shared_ptr<Shape> clone() const override { return std::make_shared<Sphere>(*this); }
 protected:
  double localFunc(const Vec3& q) const override { return 1.0 - dot(q, q) / r2_; }
  Vec3 localGrad(const Vec3& q) const override {   // -2.0 * p / r2, left to right
    return Vec3(-2.0 * q[0] / r2_, -2.0 * q[1] / r2_, -2.0 * q[2] / r2_);
  }
 private:
  double r2_;
};

class ShapeIntersection : public Shape {
 public:
  ShapeIntersection() {}
  void add(std::shared_ptr<const Shape> s) { shapes_.push_back(std::move(s)); }
  std::shared_ptr<Shape> clone() const override { return std::make_shared<ShapeIntersection>(*this); }
 protected:
  double localFunc(const Vec3& q) const override;
  Vec3 localGrad(const Vec3& q) const override;
 private:
  std::vector<std::shared_ptr<const Shape>> shapes_;
};

class Slab : public Shape {
 public:
  Slab(double thickness, const Vec3& normal, const Vec3& midplanePoint);
  std::shared_ptr<Shape> clone() const override { return std::make_shared<Slab>(*this); }
 protected:
  double localFunc(const Vec3& q) const override { return slab_.func(q); }
  Vec3 localGrad(const Vec3& q) const override { return slab_.gradFunc(q); }
 private:
  ShapeIntersection slab_;
};

// src/MxEllipsoid.hpp:33-35,46-48: 1 - sum p_i^2 / a_i^2
class Ellipsoid : public Shape {
 public:
  Ellipsoid(const Vec3& axes, const Vec3& location);
  std::shared_ptr<Shape> clone() const override { return std::make_shared<Ellipsoid>(*this); }
 protected:
  double localFunc(const Vec3& q) const override {
    return 1.0 - (q[0] * (ia_[0] * q[0]) + q[1] * (ia_[1] * q[1]) + q[2] * (ia_[2] * q[2]));
  }
  Vec3 localGrad(const Vec3& q) const override { return Vec3((-2.0 * ia_[0]) * q[0], (-2.0 * ia_[1]) * q[1], (-2.0 * ia_[2]) * q[2]); }
 private:
  Vec3 ia_;
};

// src/MxTorus.hpp:31-37: r^2 - (R - |P p|)^2 - (a.p)^2
class Torus : public Shape {
 public:
  Torus(double majorRadius, double minorRadius, const Vec3& axis, const Vec3& location);
  std::shared_ptr<Shape> clone() const override { return std::make_shared<Torus>(*this); }
 protected:
  double localFunc(const Vec3& q) const override;
  Vec3 localGrad(const Vec3& q) const override;
 private:
  double R_, r2_;
  Vec3 a_;
  Mat3 P_;
};

// src/MxCone.hpp:31-37: (tan(theta) a.p)^2 - p.(P p), vertex at the translation point (double cone)
class Cone : public Shape {
 public:
  Cone(double angle, const Vec3& axis, const Vec3& vertex);
  std::shared_ptr<Shape> clone() const override { return std::make_shared<Cone>(*this); }
 protected:
  double localFunc(const Vec3& q) const override;
  Vec3 localGrad(const Vec3& q) const override;
 private:
  double tan_;
  Vec3 a_;
  Mat3 P_;
};

// src/MxShapeUnion.hpp:68-83,136-154: max of the members, gradient of the first member attaining it
class ShapeUnion : public Shape {
 public:
  ShapeUnion() {}
  void add(std::shared_ptr<const Shape> s) { shapes_.push_back(std::move(s)); }
  std::shared_ptr<Shape> clone() const override { return std::make_shared<ShapeUnion>(*this); }
 protected:
  double localFunc(const Vec3& q) const override;
  Vec3 localGrad(const Vec3& q) const override;
 private:
  std::vector<std::shared_ptr<const Shape>> shapes_;
};

// src/MxShapeSubtract.hpp:73-90,157-175: base minus the (inverted) union of the subtracted shapes
class ShapeSubtract : public Shape {
 public:
  ShapeSubtract() { rm_.invert(); }
  void setBaseShape(std::shared_ptr<const Shape> s) { base_ = std::move(s); }
  void subtractShape(std::shared_ptr<const Shape> s) { rm_.add(std::move(s)); }
  std::shared_ptr<Shape> clone() const override { return std::make_shared<ShapeSubtract>(*this); }
 protected:
  double localFunc(const Vec3& q) const override;
  Vec3 localGrad(const Vec3& q) const override;
 private:
  std::shared_ptr<const Shape> base_;
  ShapeUnion rm_;
};

// src/MxShapeMirror.hpp:86-116,156-170: the shape on the positive side of the plane, its mirror image on the other
class ShapeMirror : public Shape {
 public:
  ShapeMirror(std::shared_ptr<const Shape> shape, const Vec3& normal, const Vec3& pointInPlane);
  std::shared_ptr<Shape> clone() const override { return std::make_shared<ShapeMirror>(*this); }
 protected:
  double localFunc(const Vec3& q) const override { return plane_->func(q) > 0.0 ? shape_->func(q) : mirror_->func(q); }
  Vec3 localGrad(const Vec3& q) const override { return plane_->func(q) > 0.0 ? shape_->gradFunc(q) : mirror_->gradFunc(q); }
 private:
  std::shared_ptr<const Shape> shape_, mirror_, plane_;
};

// src/MxShapeRepeat.hpp:86-120,140-150: copies of the shape at origin + i * step * direction, i = -numNeg..numPos
class ShapeRepeat : public Shape {
 public:
  ShapeRepeat(std::shared_ptr<const Shape> shape, const Vec3& origin, const Vec3& direction, double step, int numPos, int numNeg);
  std::shared_ptr<Shape> clone() const override { return std::make_shared<ShapeRepeat>(*this); }
 protected:
  double localFunc(const Vec3& q) const override { return shape_->func(fold(q)); }
  Vec3 localGrad(const Vec3& q) const override { return shape_->gradFunc(fold(q)); }
 private:
  Vec3 fold(const Vec3& p) const;
  std::shared_ptr<const Shape> shape_;
  Vec3 o_, dir_;
  double s_, np_, nn_;
};

Vec3 rootFind(const Shape& shape, const Vec3& p1, const Vec3& p2, double tol = 1.e-12);

// fraction of a Cartesian segment (centre `mid`, direction `axis`, length `len`) inside the shape
double segmentFraction(const Shape& shape, const Vec3& mid, int axis, double len);
// fraction of a Cartesian rectangle with normal `axis`, side l1 along axis+1, l2 along axis+2
double rectFraction(const Shape& shape, const Vec3& center, int axis, double l1, double l2);
// fraction of a Cartesian box
double boxFraction(const Shape& shape, const Vec3& center, double lx, double ly, double lz);

}  // namespace mx
#endif
