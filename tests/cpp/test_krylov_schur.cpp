// CPU unit test of the host-side eigen-solver logic (include/mx/AnasaziCompat.hpp) with a plain
// in-memory MultiVec: no GPU, no oracle.  Checks a diagonal operator with known spectrum,
// an M-inner-product problem, a degenerate pair (block size 2) and the invariant-subspace exit.
#include <cstdio>
#include <cstdlib>
#include <complex>
#include <memory>
#include <random>

#include "mx/AnasaziCompat.hpp"

typedef double S;

template <typename T>
class DenseMVT : public Anasazi::MultiVec<T> {
 public:
  DenseMVT(int n, int k) : n_(n), k_(k), own_(new std::vector<T>((size_t)n * k, 0.0)), data_(own_.get()) { for (int j = 0; j < k; ++j) cols_.push_back(j); }
  DenseMVT(const DenseMVT& o, const std::vector<int>& idx, bool copy) : n_(o.n_), k_((int)idx.size()) {
    if (copy) {
      own_.reset(new std::vector<T>((size_t)n_ * k_));
      data_ = own_.get();
      for (int j = 0; j < k_; ++j) { cols_.push_back(j); for (int i = 0; i < n_; ++i) at(i, j) = o.at(i, idx[j]); }
    } else {
      own_ = o.own_; data_ = o.data_;
      for (int j = 0; j < k_; ++j) cols_.push_back(o.cols_[idx[j]]);
    }
  }
  T& at(int i, int j) { return (*data_)[(size_t)cols_[j] * n_ + i]; }
  T at(int i, int j) const { return (*data_)[(size_t)cols_[j] * n_ + i]; }
  Anasazi::MultiVec<T>* Clone(const int nv) const { return new DenseMVT(n_, nv); }
  Anasazi::MultiVec<T>* CloneCopy() const { std::vector<int> a(k_); for (int j = 0; j < k_; ++j) a[j] = j; return new DenseMVT(*this, a, true); }
  Anasazi::MultiVec<T>* CloneCopy(const std::vector<int>& idx) const { return new DenseMVT(*this, idx, true); }
  const Anasazi::MultiVec<T>* CloneView(const std::vector<int>& idx) const { return new DenseMVT(*this, idx, false); }
  Anasazi::MultiVec<T>* CloneViewNonConst(const std::vector<int>& idx) { return new DenseMVT(*this, idx, false); }
  int GetVecLength() const { return n_; }
  int GetNumberVecs() const { return k_; }
  void MvTimesMatAddMv(T alpha, const Anasazi::MultiVec<T>& A_, const Teuchos::SerialDenseMatrix<int, T>& B, T beta) {
    const DenseMVT& A = dynamic_cast<const DenseMVT<T>&>(A_);
    for (int j = 0; j < k_; ++j)
      for (int i = 0; i < n_; ++i) {
        T s = 0; for (int l = 0; l < A.k_; ++l) s += A.at(i, l) * B(l, j);
        at(i, j) = alpha * s + (beta == T(0.0) ? T(0.0) : beta * at(i, j));
      }
  }
  void MvAddMv(T a, const Anasazi::MultiVec<T>& A_, T b, const Anasazi::MultiVec<T>& B_) {
    const DenseMVT& A = dynamic_cast<const DenseMVT<T>&>(A_); const DenseMVT& B = dynamic_cast<const DenseMVT<T>&>(B_);
    for (int j = 0; j < k_; ++j) for (int i = 0; i < n_; ++i) at(i, j) = a * A.at(i, j) + b * B.at(i, j);
  }
  void MvTransMv(T alpha, const Anasazi::MultiVec<T>& A_, Teuchos::SerialDenseMatrix<int, T>& B) const {
    const DenseMVT& A = dynamic_cast<const DenseMVT<T>&>(A_);
    for (int l = 0; l < A.k_; ++l) for (int j = 0; j < k_; ++j) { T s = 0; for (int i = 0; i < n_; ++i) s += cj(A.at(i, l)) * at(i, j); B(l, j) = alpha * s; }
  }
  void MvDot(const Anasazi::MultiVec<T>& A_, std::vector<T>& b) const {
    const DenseMVT& A = dynamic_cast<const DenseMVT<T>&>(A_); b.resize(k_);
    for (int j = 0; j < k_; ++j) { T s = 0; for (int i = 0; i < n_; ++i) s += cj(A.at(i, j)) * at(i, j); b[j] = s; }
  }
  void MvNorm(std::vector<double>& nv) const { nv.resize(k_); for (int j = 0; j < k_; ++j) { double s = 0; for (int i = 0; i < n_; ++i) s += std::norm(at(i, j)); nv[j] = std::sqrt(s); } }
  void SetBlock(const Anasazi::MultiVec<T>& A_, const std::vector<int>& idx) {
    const DenseMVT& A = dynamic_cast<const DenseMVT<T>&>(A_);
    for (size_t j = 0; j < idx.size() && (int)j < A.k_; ++j) for (int i = 0; i < n_; ++i) at(i, idx[j]) = A.at(i, (int)j);
  }
  void MvScale(T a) { for (int j = 0; j < k_; ++j) for (int i = 0; i < n_; ++i) at(i, j) *= a; }
  void MvScale(const std::vector<T>& a) { for (int j = 0; j < k_; ++j) for (int i = 0; i < n_; ++i) at(i, j) *= a[j]; }
  void MvRandom() { static std::mt19937 g(7); std::uniform_real_distribution<double> u(-1, 1); for (int j = 0; j < k_; ++j) for (int i = 0; i < n_; ++i) at(i, j) = rnd(u, g, (T*)0); }
  void MvInit(T a) { for (int j = 0; j < k_; ++j) for (int i = 0; i < n_; ++i) at(i, j) = a; }
  void MvPrint(std::ostream&) const {}
 private:
  static double cj(double v) { return v; }
  static std::complex<double> cj(std::complex<double> v) { return std::conj(v); }
  template <typename U, typename G> static double rnd(U& u, G& g, double*) { return u(g); }
  template <typename U, typename G> static std::complex<double> rnd(U& u, G& g, std::complex<double>*) { double a = u(g), b = u(g); return std::complex<double>(a, b); }
  int n_, k_;
  std::shared_ptr<std::vector<T> > own_;
  std::vector<T>* data_;
  std::vector<int> cols_;
};

typedef DenseMVT<S> DenseMV;

struct DiagOp : Anasazi::Operator<S> {
  std::vector<double> d;
  void Apply(const Anasazi::MultiVec<S>& x_, Anasazi::MultiVec<S>& y_) const {
    const DenseMV& x = dynamic_cast<const DenseMV&>(x_); DenseMV& y = dynamic_cast<DenseMV&>(y_);
    for (int j = 0; j < x.GetNumberVecs(); ++j) for (int i = 0; i < x.GetVecLength(); ++i) y.at(i, j) = d[i] * x.at(i, j);
  }
};

typedef Anasazi::MultiVec<S> MV;
typedef Anasazi::Operator<S> OP;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static std::vector<double> run(RCP<OP> op, RCP<OP> M, int n, int nev, int block, int basis, Anasazi::ReturnType* rc, int* numVecs = 0) {
  RCP<MV> init(new DenseMV(n, block));
  init->MvRandom();
  RCP<Anasazi::BasicEigenproblem<S, MV, OP> > prob(new Anasazi::BasicEigenproblem<S, MV, OP>(op, init));
  prob->setNEV(nev); prob->setHermitian(true);
  if (!M.is_null()) prob->setM(M);
  CHECK(prob->setProblem());
  Teuchos::ParameterList pl;
  pl.set("Which", "LM"); pl.set("Block Size", block); pl.set("Num Blocks", basis); pl.set("Maximum Restarts", 200);
  pl.set("Convergence Tolerance", 1e-11); pl.set("Verbosity", 0);
  Anasazi::BlockKrylovSchurSolMgr<S, MV, OP> mgr(prob, pl);
  *rc = mgr.solve();
  std::vector<double> out;
  for (auto& v : prob->getSolution().Evals) out.push_back(v.realpart);
  if (numVecs) *numVecs = prob->getSolution().numVecs;
  return out;
}

int main() {
  Anasazi::ReturnType rc;
  {  // 1. diagonal operator with well separated large eigenvalues, restarts needed
    const int n = 400;
    RCP<DiagOp> op(new DiagOp());
    for (int i = 0; i < n; ++i) op->d.push_back(1.0 / (1.0 + i));       // largest: 1, 1/2, 1/3, ...
    std::vector<double> ev = run(RCP<OP>(op), Teuchos::null, n, 4, 1, 12, &rc);
    CHECK(rc == Anasazi::Converged);
    CHECK(ev.size() == 4);
    for (int i = 0; i < 4; ++i) CHECK(std::fabs(ev[i] - 1.0 / (1.0 + i)) < 1e-10);
  }
  {  // 2. doubly degenerate pair with block size 2 (the pillbox TE111 situation)
    const int n = 300;
    RCP<DiagOp> op(new DiagOp());
    for (int i = 0; i < n; ++i) op->d.push_back(1.0 / (2.0 + i / 2));   // pairs: 1/2,1/2,1/3,1/3,...
    std::vector<double> ev = run(RCP<OP>(op), Teuchos::null, n, 4, 2, 10, &rc);
    CHECK(rc == Anasazi::Converged);
    CHECK(std::fabs(ev[0] - 0.5) < 1e-10 && std::fabs(ev[1] - 0.5) < 1e-10);
    CHECK(std::fabs(ev[2] - 1.0 / 3) < 1e-10 && std::fabs(ev[3] - 1.0 / 3) < 1e-10);
  }
  {  // 3. M-inner product: Op = K M with K, M diagonal is self-adjoint w.r.t. M, eigenvalues k_i m_i
    const int n = 250;
    RCP<DiagOp> op(new DiagOp()), M(new DiagOp());
    std::vector<double> exact;
    for (int i = 0; i < n; ++i) { double m = 0.2 + (i % 7) * 0.3, kk = 1.0 / (1 + i); M->d.push_back(m); op->d.push_back(kk * m); exact.push_back(kk * m); }
    std::sort(exact.begin(), exact.end(), [](double a, double b) { return a > b; });
    std::vector<double> ev = run(RCP<OP>(op), RCP<OP>(M), n, 3, 1, 15, &rc);
    CHECK(rc == Anasazi::Converged);
    for (int i = 0; i < 3; ++i) CHECK(std::fabs(ev[i] - exact[i]) < 1e-10);
  }
  {  // 4. invariant subspace smaller than the basis: exits with what it has
    const int n = 6;
    RCP<DiagOp> op(new DiagOp());
    for (int i = 0; i < n; ++i) op->d.push_back(i + 1.0);
    int nv = 0;
    std::vector<double> ev = run(RCP<OP>(op), Teuchos::null, n, 3, 1, 20, &rc, &nv);
    CHECK(nv == 3);
    CHECK(std::fabs(ev[0] - 6.0) < 1e-9 && std::fabs(ev[1] - 5.0) < 1e-9 && std::fabs(ev[2] - 4.0) < 1e-9);
  }
  {  // 5. the small general eigen-solver: residuals on a random non-symmetric matrix
    const int n = 40;
    std::mt19937 g(3); std::uniform_real_distribution<double> u(-1, 1);
    std::vector<double> A(n * n);
    for (auto& a : A) a = u(g);
    std::vector<std::complex<double> > w, X;
    CHECK(Anasazi::detail::genEig(n, A, w, X));
    std::complex<double> tr(0.0);
    for (int c = 0; c < n; ++c) {
      tr += w[c];
      double r = 0;
      for (int i = 0; i < n; ++i) {
        std::complex<double> s(0.0);
        for (int j = 0; j < n; ++j) s += A[i + n * j] * X[j + n * c];
        s -= w[c] * X[i + n * c];
        r += std::norm(s);
      }
      CHECK(std::sqrt(r) < 1e-11);
    }
    double tra = 0; for (int i = 0; i < n; ++i) tra += A[i + n * i];
    CHECK(std::abs(tr - tra) < 1e-11);
  }
  {  // 5b. non-Hermitian Krylov-Schur: a non-normal operator (diagonal + strictly upper part) has the
     //     diagonal as its spectrum; block sizes 1 and 2; with and without an M inner product
    const int n = 300;
    struct TriOp : OP {
      std::vector<double> d; double off;
      void Apply(const MV& x_, MV& y_) const {
        const DenseMV& x = dynamic_cast<const DenseMV&>(x_); DenseMV& y = dynamic_cast<DenseMV&>(y_);
        const int nn = x.GetVecLength();
        for (int j = 0; j < x.GetNumberVecs(); ++j)
          for (int i = 0; i < nn; ++i) {
            double s = d[i] * x.at(i, j);
            if (i + 1 < nn) s += off * x.at(i + 1, j);
            if (i + 3 < nn) s -= 0.5 * off * x.at(i + 3, j);
            y.at(i, j) = s;
          }
      }
    };
    for (int block = 1; block <= 2; ++block)
      for (int withM = 0; withM < 2; ++withM) {
        RCP<TriOp> op(new TriOp()); op->off = 0.04;   // mildly non-normal: a large off-diagonal part fills the disc |z| < 0.45 with pseudo-eigenvalues
        for (int i = 0; i < n; ++i) op->d.push_back(1.0 / (1.0 + 0.7 * i));
        RCP<DiagOp> M(new DiagOp());
        for (int i = 0; i < n; ++i) M->d.push_back(0.5 + (i % 5) * 0.25);
        RCP<MV> init(new DenseMV(n, block)); init->MvRandom();
        RCP<Anasazi::BasicEigenproblem<S, MV, OP> > prob(new Anasazi::BasicEigenproblem<S, MV, OP>(RCP<OP>(op), init));
        prob->setNEV(4); prob->setHermitian(false);
        if (withM) prob->setM(RCP<OP>(M));
        CHECK(prob->setProblem());
        Teuchos::ParameterList pl;
        pl.set("Which", "LM"); pl.set("Block Size", block); pl.set("Num Blocks", 14); pl.set("Maximum Restarts", 300);
        pl.set("Convergence Tolerance", 1e-10); pl.set("Verbosity", 0);
        Anasazi::BlockKrylovSchurSolMgr<S, MV, OP> mgr(prob, pl);
        CHECK(mgr.solve() == Anasazi::Converged);
        const Anasazi::Eigensolution<S, MV>& sol = prob->getSolution();
        CHECK(sol.numVecs >= 4);
        for (int i = 0; i < 4 && i < sol.numVecs; ++i) {
          CHECK(std::fabs(sol.Evals[i].realpart - 1.0 / (1.0 + 0.7 * i)) < 1e-8);
          CHECK(sol.Evals[i].imagpart == 0.0 && sol.index[i] == 0);
        }
        // eigenvector residual of the dominant pair
        std::unique_ptr<MV> y(sol.Evecs->Clone(sol.numVecs));
        op->Apply(*sol.Evecs, *y);
        const DenseMV& xv = dynamic_cast<const DenseMV&>(*sol.Evecs); const DenseMV& yv = dynamic_cast<const DenseMV&>(*y);
        double r = 0, nx = 0;
        for (int i = 0; i < n; ++i) { const double t = yv.at(i, 0) - sol.Evals[0].realpart * xv.at(i, 0); r += t * t; nx += xv.at(i, 0) * xv.at(i, 0); }
        CHECK(std::sqrt(r / nx) < 1e-8);
      }
  }
  {  // 5c. complex conjugate pair: 2x2 rotation-scaling blocks; the dominant pair comes back as (Re, Im)
    const int n = 120;
    struct RotOp : OP {
      void Apply(const MV& x_, MV& y_) const {
        const DenseMV& x = dynamic_cast<const DenseMV&>(x_); DenseMV& y = dynamic_cast<DenseMV&>(y_);
        const int nn = x.GetVecLength();
        for (int j = 0; j < x.GetNumberVecs(); ++j) {
          y.at(0, j) = 2.0 * x.at(0, j) - 1.5 * x.at(1, j);       // eigenvalues 2 +- 1.5 i, modulus 2.5
          y.at(1, j) = 1.5 * x.at(0, j) + 2.0 * x.at(1, j);
          for (int i = 2; i < nn; ++i) y.at(i, j) = (2.0 / (1.0 + 0.5 * (i - 2))) * x.at(i, j) + (i + 1 < nn ? 0.1 * x.at(i + 1, j) : 0.0);
        }
      }
    };
    RCP<OP> op(new RotOp());
    RCP<MV> init(new DenseMV(n, 1)); init->MvRandom();
    RCP<Anasazi::BasicEigenproblem<S, MV, OP> > prob(new Anasazi::BasicEigenproblem<S, MV, OP>(op, init));
    prob->setNEV(3); prob->setHermitian(false);
    CHECK(prob->setProblem());
    Teuchos::ParameterList pl;
    pl.set("Which", "LM"); pl.set("Block Size", 1); pl.set("Num Blocks", 16); pl.set("Maximum Restarts", 300);
    pl.set("Convergence Tolerance", 1e-10); pl.set("Verbosity", 0);
    Anasazi::BlockKrylovSchurSolMgr<S, MV, OP> mgr(prob, pl);
    CHECK(mgr.solve() == Anasazi::Converged);
    const Anasazi::Eigensolution<S, MV>& sol = prob->getSolution();
    CHECK(sol.numVecs >= 3);
    CHECK(std::fabs(sol.Evals[0].realpart - 2.0) < 1e-8 && std::fabs(sol.Evals[0].imagpart - 1.5) < 1e-8 && sol.index[0] == 1);
    CHECK(std::fabs(sol.Evals[1].realpart - 2.0) < 1e-8 && std::fabs(sol.Evals[1].imagpart + 1.5) < 1e-8 && sol.index[1] == -1);
    CHECK(std::fabs(sol.Evals[2].realpart - 2.0) < 1e-8 && sol.Evals[2].imagpart == 0.0);
  }
  {  // 5d. complex scalars: Hermitian operator U D U^H (U = diagonal phases times a Householder reflector) and a
     //     non-Hermitian one (complex diagonal + upper part); both through the complex path
    typedef std::complex<double> Z;
    typedef Anasazi::MultiVec<Z> ZMV;
    typedef Anasazi::Operator<Z> ZOP;
    const int n = 200;
    struct HermOp : ZOP {
      std::vector<double> d; std::vector<Z> u;
      void Apply(const ZMV& x_, ZMV& y_) const {
        const DenseMVT<Z>& x = dynamic_cast<const DenseMVT<Z>&>(x_); DenseMVT<Z>& y = dynamic_cast<DenseMVT<Z>&>(y_);
        const int nn = x.GetVecLength();
        std::vector<Z> t(nn);
        for (int j = 0; j < x.GetNumberVecs(); ++j) {
          Z s(0.0);                                  // t = (I - 2 u u^H) x
          for (int i = 0; i < nn; ++i) s += std::conj(u[i]) * x.at(i, j);
          for (int i = 0; i < nn; ++i) t[i] = d[i] * (x.at(i, j) - 2.0 * u[i] * s);
          s = Z(0.0);
          for (int i = 0; i < nn; ++i) s += std::conj(u[i]) * t[i];
          for (int i = 0; i < nn; ++i) y.at(i, j) = t[i] - 2.0 * u[i] * s;
        }
      }
    };
    RCP<HermOp> hop(new HermOp());
    double un = 0;
    for (int i = 0; i < n; ++i) { hop->d.push_back(1.0 / (1.0 + 0.5 * i)); hop->u.push_back(Z(std::cos(0.3 * i), std::sin(0.7 * i))); un += std::norm(hop->u.back()); }
    for (auto& v : hop->u) v /= std::sqrt(un);
    struct TriOpZ : ZOP {
      std::vector<Z> d;
      void Apply(const ZMV& x_, ZMV& y_) const {
        const DenseMVT<Z>& x = dynamic_cast<const DenseMVT<Z>&>(x_); DenseMVT<Z>& y = dynamic_cast<DenseMVT<Z>&>(y_);
        const int nn = x.GetVecLength();
        for (int j = 0; j < x.GetNumberVecs(); ++j)
          for (int i = 0; i < nn; ++i) y.at(i, j) = d[i] * x.at(i, j) + (i + 1 < nn ? Z(0.03, 0.02) * x.at(i + 1, j) : Z(0.0));
      }
    };
    RCP<TriOpZ> top(new TriOpZ());
    for (int i = 0; i < n; ++i) top->d.push_back(Z(1.0 / (1.0 + 0.6 * i), 0.2 / (1.0 + i)));
    for (int which = 0; which < 2; ++which) {
      RCP<ZMV> init(new DenseMVT<Z>(n, 1)); init->MvRandom();
      RCP<ZOP> op = which == 0 ? RCP<ZOP>(hop) : RCP<ZOP>(top);
      RCP<Anasazi::BasicEigenproblem<Z, ZMV, ZOP> > prob(new Anasazi::BasicEigenproblem<Z, ZMV, ZOP>(op, init));
      prob->setNEV(3); prob->setHermitian(which == 0);
      CHECK(prob->setProblem());
      Teuchos::ParameterList pl;
      pl.set("Which", "LM"); pl.set("Block Size", 1); pl.set("Num Blocks", 14); pl.set("Maximum Restarts", 300);
      pl.set("Convergence Tolerance", 1e-10); pl.set("Verbosity", 0);
      Anasazi::BlockKrylovSchurSolMgr<Z, ZMV, ZOP> mgr(prob, pl);
      CHECK(mgr.solve() == Anasazi::Converged);
      const Anasazi::Eigensolution<Z, ZMV>& sol = prob->getSolution();
      CHECK(sol.numVecs >= 3);
      for (int i = 0; i < 3 && i < sol.numVecs; ++i) {
        const Z want = which == 0 ? Z(1.0 / (1.0 + 0.5 * i), 0.0) : top->d[i];
        CHECK(std::abs(Z(sol.Evals[i].realpart, sol.Evals[i].imagpart) - want) < 1e-8);
      }
    }
  }
  {  // 6. the small symmetric eigen-solver
    const int n = 5;
    std::vector<double> A(n * n), w, Z;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i + n * j] = 1.0 / (1 + i + j);   // Hilbert
    std::vector<double> A0 = A;
    Anasazi::detail::symEig(n, A, w, Z);
    for (int c = 0; c < n; ++c) {
      double r = 0;
      for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += A0[i + n * j] * Z[j + n * c]; s -= w[c] * Z[i + n * c]; r += s * s; }
      CHECK(std::sqrt(r) < 1e-13);
    }
  }
  if (fails == 0) std::printf("PASSED\n");
  return fails == 0 ? 0 : 1;
}
