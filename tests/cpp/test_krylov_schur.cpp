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

class DenseMV : public Anasazi::MultiVec<S> {
 public:
  DenseMV(int n, int k) : n_(n), k_(k), own_(new std::vector<double>((size_t)n * k, 0.0)), data_(own_.get()) { for (int j = 0; j < k; ++j) cols_.push_back(j); }
  DenseMV(const DenseMV& o, const std::vector<int>& idx, bool copy) : n_(o.n_), k_((int)idx.size()) {
    if (copy) {
      own_.reset(new std::vector<double>((size_t)n_ * k_));
      data_ = own_.get();
      for (int j = 0; j < k_; ++j) { cols_.push_back(j); for (int i = 0; i < n_; ++i) at(i, j) = o.at(i, idx[j]); }
    } else {
      own_ = o.own_; data_ = o.data_;
      for (int j = 0; j < k_; ++j) cols_.push_back(o.cols_[idx[j]]);
    }
  }
  double& at(int i, int j) { return (*data_)[(size_t)cols_[j] * n_ + i]; }
  double at(int i, int j) const { return (*data_)[(size_t)cols_[j] * n_ + i]; }
  Anasazi::MultiVec<S>* Clone(const int nv) const { return new DenseMV(n_, nv); }
  Anasazi::MultiVec<S>* CloneCopy() const { std::vector<int> a(k_); for (int j = 0; j < k_; ++j) a[j] = j; return new DenseMV(*this, a, true); }
  Anasazi::MultiVec<S>* CloneCopy(const std::vector<int>& idx) const { return new DenseMV(*this, idx, true); }
  const Anasazi::MultiVec<S>* CloneView(const std::vector<int>& idx) const { return new DenseMV(*this, idx, false); }
  Anasazi::MultiVec<S>* CloneViewNonConst(const std::vector<int>& idx) { return new DenseMV(*this, idx, false); }
  int GetVecLength() const { return n_; }
  int GetNumberVecs() const { return k_; }
  void MvTimesMatAddMv(S alpha, const Anasazi::MultiVec<S>& A_, const Teuchos::SerialDenseMatrix<int, S>& B, S beta) {
    const DenseMV& A = dynamic_cast<const DenseMV&>(A_);
    for (int j = 0; j < k_; ++j)
      for (int i = 0; i < n_; ++i) {
        double s = 0; for (int l = 0; l < A.k_; ++l) s += A.at(i, l) * B(l, j);
        at(i, j) = alpha * s + (beta == 0.0 ? 0.0 : beta * at(i, j));
      }
  }
  void MvAddMv(S a, const Anasazi::MultiVec<S>& A_, S b, const Anasazi::MultiVec<S>& B_) {
    const DenseMV& A = dynamic_cast<const DenseMV&>(A_); const DenseMV& B = dynamic_cast<const DenseMV&>(B_);
    for (int j = 0; j < k_; ++j) for (int i = 0; i < n_; ++i) at(i, j) = a * A.at(i, j) + b * B.at(i, j);
  }
  void MvTransMv(S alpha, const Anasazi::MultiVec<S>& A_, Teuchos::SerialDenseMatrix<int, S>& B) const {
    const DenseMV& A = dynamic_cast<const DenseMV&>(A_);
    for (int l = 0; l < A.k_; ++l) for (int j = 0; j < k_; ++j) { double s = 0; for (int i = 0; i < n_; ++i) s += A.at(i, l) * at(i, j); B(l, j) = alpha * s; }
  }
  void MvDot(const Anasazi::MultiVec<S>& A_, std::vector<S>& b) const {
    const DenseMV& A = dynamic_cast<const DenseMV&>(A_); b.resize(k_);
    for (int j = 0; j < k_; ++j) { double s = 0; for (int i = 0; i < n_; ++i) s += A.at(i, j) * at(i, j); b[j] = s; }
  }
  void MvNorm(std::vector<double>& nv) const { nv.resize(k_); for (int j = 0; j < k_; ++j) { double s = 0; for (int i = 0; i < n_; ++i) s += at(i, j) * at(i, j); nv[j] = std::sqrt(s); } }
  void SetBlock(const Anasazi::MultiVec<S>& A_, const std::vector<int>& idx) {
    const DenseMV& A = dynamic_cast<const DenseMV&>(A_);
    for (size_t j = 0; j < idx.size() && (int)j < A.k_; ++j) for (int i = 0; i < n_; ++i) at(i, idx[j]) = A.at(i, (int)j);
  }
  void MvScale(S a) { for (int j = 0; j < k_; ++j) for (int i = 0; i < n_; ++i) at(i, j) *= a; }
  void MvScale(const std::vector<S>& a) { for (int j = 0; j < k_; ++j) for (int i = 0; i < n_; ++i) at(i, j) *= a[j]; }
  void MvRandom() { static std::mt19937 g(7); std::uniform_real_distribution<double> u(-1, 1); for (int j = 0; j < k_; ++j) for (int i = 0; i < n_; ++i) at(i, j) = u(g); }
  void MvInit(S a) { for (int j = 0; j < k_; ++j) for (int i = 0; i < n_; ++i) at(i, j) = a; }
  void MvPrint(std::ostream&) const {}
 private:
  int n_, k_;
  std::shared_ptr<std::vector<double> > own_;
  std::vector<double>* data_;
  std::vector<int> cols_;
};

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
  {  // 5d. complex scalars + non-Hermitian are refused loudly (not built)
    typedef std::complex<double> Z;
    struct NoOp : Anasazi::Operator<Z> { void Apply(const Anasazi::MultiVec<Z>&, Anasazi::MultiVec<Z>&) const {} };
    RCP<Anasazi::Operator<Z> > op(new NoOp());
    RCP<Anasazi::MultiVec<Z> > init;   // never dereferenced: the constructor checks come first
    RCP<Anasazi::BasicEigenproblem<Z, Anasazi::MultiVec<Z>, Anasazi::Operator<Z> > > prob(
        new Anasazi::BasicEigenproblem<Z, Anasazi::MultiVec<Z>, Anasazi::Operator<Z> >(op, init));
    Teuchos::ParameterList pl;
    bool threw = false;
    try { Anasazi::BlockKrylovSchurSolMgr<Z, Anasazi::MultiVec<Z>, Anasazi::Operator<Z> > mgr(prob, pl); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
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
