// Basic types of the drop-in C++ surface + the small slice of Teuchos the reference's
// signatures mention (RCP, ScalarTraits, SerialDenseMatrix, ParameterList), so the same class
// declarations compile without Trilinos.  Reference: src/MxTypes.h:14-51.
#ifndef MX_TYPES
#define MX_TYPES

#include <complex>
#include <cstddef>
#include <cstdlib>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "mxg.h"

typedef size_t MxIndex;                       // src/MxTypes.h:47
typedef std::complex<double> MxComplex;       // src/MxTypes.h:50

enum MxPolType { TE, TM, TEM };               // src/MxTypes.h:20
enum MxBCType { PEC, PMC, PERIODIC, ZERO, CONSTANT };

namespace Teuchos {

struct ENull {};
static const ENull null = ENull();

// Reference-counted pointer with the part of Teuchos::RCP's interface the reference uses.
template <typename T>
class RCP {
 public:
  RCP() {}
  RCP(ENull) {}
  explicit RCP(T* p) : p_(p) {}
  RCP(const std::shared_ptr<T>& p) : p_(p) {}
  template <typename U> RCP(const RCP<U>& o) : p_(o.shared()) {}
  T* get() const { return p_.get(); }
  T* operator->() const { return p_.get(); }
  T& operator*() const { return *p_; }
  bool is_null() const { return !p_; }
  bool operator==(ENull) const { return !p_; }
  bool operator!=(ENull) const { return (bool)p_; }
  RCP& operator=(ENull) { p_.reset(); return *this; }
  int strong_count() const { return (int)p_.use_count(); }
  const std::shared_ptr<T>& shared() const { return p_; }
 private:
  std::shared_ptr<T> p_;
};

template <typename T> RCP<T> rcp(T* p) { return RCP<T>(p); }
template <typename T, typename U> RCP<T> rcp_dynamic_cast(const RCP<U>& o, bool throw_on_fail = false) {
  std::shared_ptr<T> p = std::dynamic_pointer_cast<T>(o.shared());
  if (!p && throw_on_fail) throw std::runtime_error("rcp_dynamic_cast failed");
  return RCP<T>(p);
}

template <typename S> struct ScalarTraits;
template <> struct ScalarTraits<double> {
  typedef double magnitudeType;
  static const bool isComplex = false;
  static double zero() { return 0.0; }
  static double one() { return 1.0; }
  static double real(double a) { return a; }
  static double imag(double) { return 0.0; }
  static double conjugate(double a) { return a; }
  static double magnitude(double a) { return a < 0 ? -a : a; }
};
template <> struct ScalarTraits<std::complex<double> > {
  typedef double magnitudeType;
  static const bool isComplex = true;
  static std::complex<double> zero() { return std::complex<double>(0.0, 0.0); }
  static std::complex<double> one() { return std::complex<double>(1.0, 0.0); }
  static double real(std::complex<double> a) { return a.real(); }
  static double imag(std::complex<double> a) { return a.imag(); }
  static std::complex<double> conjugate(std::complex<double> a) { return std::conj(a); }
  static double magnitude(std::complex<double> a) { return std::abs(a); }
};

// Column-major dense matrix with a stride (Teuchos::SerialDenseMatrix<int, S>).
template <typename Ordinal, typename S>
class SerialDenseMatrix {
 public:
  SerialDenseMatrix() : m_(0), n_(0), ld_(0) {}
  SerialDenseMatrix(Ordinal m, Ordinal n) : m_(m), n_(n), ld_(m > 0 ? m : 1), data_((size_t)(m > 0 ? m : 1) * n, S(0)) {}
  void shape(Ordinal m, Ordinal n) { m_ = m; n_ = n; ld_ = m > 0 ? m : 1; data_.assign((size_t)ld_ * n, S(0)); }
  S& operator()(Ordinal i, Ordinal j) { return data_[(size_t)i + (size_t)ld_ * j]; }
  const S& operator()(Ordinal i, Ordinal j) const { return data_[(size_t)i + (size_t)ld_ * j]; }
  Ordinal numRows() const { return m_; }
  Ordinal numCols() const { return n_; }
  Ordinal stride() const { return ld_; }
  S* values() { return data_.data(); }
  const S* values() const { return data_.data(); }
  void putScalar(S v) { for (auto& x : data_) x = v; }
 private:
  Ordinal m_, n_, ld_;
  std::vector<S> data_;
};

// string-keyed parameters with typed get(name, default) (Teuchos::ParameterList subset)
class ParameterList {
 public:
  template <typename T> void set(const std::string& k, const T& v) { std::ostringstream os; os.precision(17); os << v; kv_[k] = os.str(); }
  void set(const std::string& k, const char* v) { kv_[k] = v; }
  void set(const std::string& k, bool v) { kv_[k] = v ? "1" : "0"; }
  template <typename T> T get(const std::string& k, const T& def) const {
    auto it = kv_.find(k);
    if (it == kv_.end()) return def;
    std::istringstream is(it->second); T v; is >> v; return v;
  }
  std::string get(const std::string& k, const char* def) const {
    auto it = kv_.find(k);
    return it == kv_.end() ? std::string(def) : it->second;
  }
  bool isParameter(const std::string& k) const { return kv_.count(k) != 0; }
  friend std::ostream& operator<<(std::ostream& os, const ParameterList& p) {
    for (auto& e : p.kv_) os << e.first << " = " << e.second << "\n";
    return os;
  }
 private:
  std::map<std::string, std::string> kv_;
};
template <> inline std::string ParameterList::get<std::string>(const std::string& k, const std::string& def) const {
  auto it = kv_.find(k);
  return it == kv_.end() ? def : it->second;
}

}  // namespace Teuchos

using Teuchos::RCP;
using Teuchos::rcp;
using Teuchos::ScalarTraits;

// (re, im) view of a scalar for the C ABI
inline double mxRe(double a) { return a; }
inline double mxIm(double) { return 0.0; }
inline double mxRe(const MxComplex& a) { return a.real(); }
inline double mxIm(const MxComplex& a) { return a.imag(); }

// errors follow the reference: message on stdout, then `throw 1` (src/MxCrsMatrix.cpp:138-142)
inline void mxFail(const std::string& where) {
  std::cout << where << ": " << mxg_last_error() << "\n";
  throw 1;
}
#define MX_ABI(call, where) do { if ((call) != 0) mxFail(where); } while (0)

#endif
