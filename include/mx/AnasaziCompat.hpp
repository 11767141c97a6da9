// The slice of Anasazi the reference programs against (MultiVec / Operator abstract classes,
// BasicEigenproblem, Eigensolution, BlockKrylovSchurSolMgr), re-implemented because Trilinos is
// not available: reference src/MxSolver.h:41-42, src/MxSolver.cpp:62-127,
// src/MxAnasaziMV.hpp:11-138, test/AnasaziInterface.cpp:53-61.
//
// BlockKrylovSchurSolMgr here is a thick-restart block Krylov-Schur for operators that are
// self-adjoint in the inner product of the eigenproblem (Euclidean, or M when setM() was
// called): the Hermitian case the reference selects with setHermitian(true) whenever there is no
// dielectric / mu (src/MxSolver.cpp:72-73).  It only touches vectors through the MultiVec
// virtuals, exactly like Anasazi, so every block operation lands in the HIP kernels of K2-K8.
#ifndef MX_ANASAZI_COMPAT
#define MX_ANASAZI_COMPAT

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <complex>

#include "MxTypes.h"

namespace Anasazi {

enum ReturnType { Converged, Unconverged };

template <typename Scalar>
class MultiVec {
 public:
  virtual ~MultiVec() {}
  virtual MultiVec<Scalar>* Clone(const int numVecs) const = 0;
  virtual MultiVec<Scalar>* CloneCopy() const = 0;
  virtual MultiVec<Scalar>* CloneCopy(const std::vector<int>& index) const = 0;
  virtual const MultiVec<Scalar>* CloneView(const std::vector<int>& index) const = 0;
  virtual MultiVec<Scalar>* CloneViewNonConst(const std::vector<int>& index) = 0;
  virtual int GetVecLength() const = 0;
  virtual int GetNumberVecs() const = 0;
  virtual void MvTimesMatAddMv(Scalar alpha, const MultiVec<Scalar>& A,
                               const Teuchos::SerialDenseMatrix<int, Scalar>& B, Scalar beta) = 0;
  virtual void MvAddMv(Scalar alpha, const MultiVec<Scalar>& A, Scalar beta, const MultiVec<Scalar>& B) = 0;
  virtual void MvTransMv(Scalar alpha, const MultiVec<Scalar>& A,
                         Teuchos::SerialDenseMatrix<int, Scalar>& B) const = 0;
  // B = alpha [A A2]^H this.  Not part of Anasazi's interface: the default is two MvTransMv calls; a multivector
  // class that can do it in one pass over the rows overrides it (MxAnasaziMV -> mxg_mv_trans_mv2)
  virtual void MvTransMv2(Scalar alpha, const MultiVec<Scalar>& A, const MultiVec<Scalar>& A2,
                          Teuchos::SerialDenseMatrix<int, Scalar>& B) const {
    const int m1 = A.GetNumberVecs(), m2 = A2.GetNumberVecs(), k = GetNumberVecs();
    Teuchos::SerialDenseMatrix<int, Scalar> B1(m1, k), B2(m2, k);
    MvTransMv(alpha, A, B1);
    MvTransMv(alpha, A2, B2);
    for (int j = 0; j < k; ++j) {
      for (int i = 0; i < m1; ++i) B(i, j) = B1(i, j);
      for (int i = 0; i < m2; ++i) B(m1 + i, j) = B2(i, j);
    }
  }
  virtual void MvDot(const MultiVec<Scalar>& A, std::vector<Scalar>& b) const = 0;
  virtual void MvNorm(std::vector<typename ScalarTraits<Scalar>::magnitudeType>& normvec) const = 0;
  virtual void SetBlock(const MultiVec<Scalar>& A, const std::vector<int>& index) = 0;
  virtual void MvScale(Scalar alpha) = 0;
  virtual void MvScale(const std::vector<Scalar>& alpha) = 0;
  virtual void MvRandom() = 0;
  virtual void MvInit(Scalar alpha) = 0;
  virtual void MvPrint(std::ostream& os) const = 0;
};

template <typename Scalar>
class Operator {
 public:
  virtual ~Operator() {}
  virtual void Apply(const MultiVec<Scalar>& x, MultiVec<Scalar>& y) const = 0;
};

template <typename Scalar>
struct Value { Scalar realpart, imagpart; };

template <typename Scalar, typename MV>
struct Eigensolution {
  RCP<MV> Evecs;
  std::vector<Value<typename ScalarTraits<Scalar>::magnitudeType> > Evals;
  std::vector<int> index;
  int numVecs = 0;
};

template <typename Scalar, typename MV, typename OP>
class BasicEigenproblem {
 public:
  BasicEigenproblem(RCP<OP> op, RCP<MV> initVec) : op_(op), init_(initVec) {}
  void setNEV(int nev) { nev_ = nev; }
  int getNEV() const { return nev_; }
  void setHermitian(bool h) { hermitian_ = h; }
  bool isHermitian() const { return hermitian_; }
  // inner-product operator (Anasazi::Eigenproblem::setM); null = Euclidean
  void setM(RCP<OP> M) { M_ = M; }
  RCP<OP> getM() const { return M_; }
  void setAuxVecs(RCP<MV> aux) { aux_ = aux; }
  bool setProblem() { set_ = !op_.is_null() && !init_.is_null() && nev_ > 0; return set_; }
  bool isProblemSet() const { return set_; }
  RCP<OP> getOperator() const { return op_; }
  RCP<MV> getInitVec() const { return init_; }
  void setSolution(const Eigensolution<Scalar, MV>& s) { sol_ = s; }
  const Eigensolution<Scalar, MV>& getSolution() const { return sol_; }
 private:
  RCP<OP> op_, M_;
  RCP<MV> init_, aux_;
  int nev_ = 1;
  bool hermitian_ = false, set_ = false;
  Eigensolution<Scalar, MV> sol_;
};

template <typename Scalar, typename MV, typename OP>
class SolverManager {
 public:
  virtual ~SolverManager() {}
  virtual ReturnType solve() = 0;
  virtual int getNumIters() const = 0;
};

namespace detail {

// cyclic Jacobi eigen-decomposition of a small real symmetric matrix (column-major n x n):
// A = Z diag(w) Z^T, eigenvalues unsorted
inline void symEig(int n, std::vector<double>& A, std::vector<double>& w, std::vector<double>& Z) {
  Z.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) Z[i + (size_t)n * i] = 1.0;
  auto a = [&](int i, int j) -> double& { return A[i + (size_t)n * j]; };
  auto z = [&](int i, int j) -> double& { return Z[i + (size_t)n * j]; };
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int j = 0; j < n; ++j) { diag += a(j, j) * a(j, j); for (int i = 0; i < j; ++i) off += a(i, j) * a(i, j); }
    if (off <= 1e-30 * (diag + off) || off == 0.0) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a(p, q);
        if (apq == 0.0) continue;
        const double theta = (a(q, q) - a(p, p)) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = a(k, p), akq = a(k, q);
          a(k, p) = c * akp - s * akq;
          a(k, q) = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = a(p, k), aqk = a(q, k);
          a(p, k) = c * apk - s * aqk;
          a(q, k) = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double zkp = z(k, p), zkq = z(k, q);
          z(k, p) = c * zkp - s * zkq;
          z(k, q) = s * zkp + c * zkq;
        }
      }
  }
  w.resize(n);
  for (int i = 0; i < n; ++i) w[i] = a(i, i);
}

// Eigen-decomposition of a small general real matrix (column-major n x n) in complex arithmetic:
// Householder reduction to Hessenberg form, single-shift QR with Givens rotations to the complex Schur
// form T = Q^H A Q, eigenvectors of T by back substitution, X = Q Y (columns of unit 2-norm).
// Dense O(n^3); n is the Krylov basis size (<= a few hundred).
inline bool genEig(int n, const std::vector<std::complex<double> >& A, std::vector<std::complex<double> >& w,
                   std::vector<std::complex<double> >& X) {
  typedef std::complex<double> C;
  std::vector<C> H(A.begin(), A.begin() + (size_t)n * n), Q((size_t)n * n, C(0.0));
  for (int i = 0; i < n; ++i) Q[i + (size_t)n * i] = C(1.0);
  auto h = [&](int i, int j) -> C& { return H[i + (size_t)n * j]; };
  auto q = [&](int i, int j) -> C& { return Q[i + (size_t)n * j]; };
  // ---- Hessenberg reduction
  std::vector<C> v(n);
  for (int k = 0; k + 2 < n; ++k) {
    double xn = 0.0;
    for (int i = k + 1; i < n; ++i) xn += std::norm(h(i, k));
    xn = std::sqrt(xn);
    if (xn == 0.0) continue;
    const C x0 = h(k + 1, k);
    const C ph = std::abs(x0) > 0.0 ? x0 / std::abs(x0) : C(1.0);
    for (int i = k + 1; i < n; ++i) v[i] = h(i, k);
    v[k + 1] += ph * xn;
    double vn = 0.0;
    for (int i = k + 1; i < n; ++i) vn += std::norm(v[i]);
    vn = std::sqrt(vn);
    if (vn == 0.0) continue;
    for (int i = k + 1; i < n; ++i) v[i] /= vn;
    for (int j = 0; j < n; ++j) {            // H <- (I - 2 v v^H) H
      C s(0.0);
      for (int i = k + 1; i < n; ++i) s += std::conj(v[i]) * h(i, j);
      s *= 2.0;
      for (int i = k + 1; i < n; ++i) h(i, j) -= v[i] * s;
    }
    for (int i = 0; i < n; ++i) {            // H <- H (I - 2 v v^H), Q likewise
      C s(0.0), t(0.0);
      for (int j = k + 1; j < n; ++j) { s += h(i, j) * v[j]; t += q(i, j) * v[j]; }
      s *= 2.0; t *= 2.0;
      for (int j = k + 1; j < n; ++j) { h(i, j) -= s * std::conj(v[j]); q(i, j) -= t * std::conj(v[j]); }
    }
    for (int i = k + 2; i < n; ++i) h(i, k) = C(0.0);
  }
  // ---- shifted QR on the active block [lo, hi]
  const double eps = 2.3e-16;
  int hi = n - 1, iter = 0, total = 0;
  std::vector<double> cs(n);
  std::vector<C> sn(n);
  // norm of the Hessenberg matrix: a subdiagonal entry is also negligible when it is below eps * (a small multiple
  // of) the norm - two neighbouring diagonal entries that are both (nearly) zero, e.g. a rank-deficient block
  // column of a block Krylov basis, would otherwise never deflate (EISPACK hqr falls back to the norm the same way)
  double hnorm = 0.0;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= std::min(j + 1, n - 1); ++i) {
      if (!std::isfinite(h(i, j).real()) || !std::isfinite(h(i, j).imag())) return false;
      hnorm = std::max(hnorm, std::abs(h(i, j)));
    }
  while (hi > 0) {
    int lo = hi;
    while (lo > 0) {
      const double sc = std::max(std::abs(h(lo - 1, lo - 1)) + std::abs(h(lo, lo)), 1e-6 * hnorm);
      if (std::abs(h(lo, lo - 1)) <= eps * (sc > 0.0 ? sc : 1.0)) { h(lo, lo - 1) = C(0.0); break; }
      --lo;
    }
    if (lo == hi) { --hi; iter = 0; continue; }
    if (lo == hi - 1) {
      // 2 x 2 active block: its Schur form directly, from an eigenvector (a nearly defective pair - two equal
      // diagonal entries over a subdiagonal of 1e-14 - lets the shifted iteration hover at rounding level for ever)
      const C a = h(lo, lo), b = h(lo, hi), c = h(hi, lo), d = h(hi, hi);
      const C tr = a + d, det = a * d - b * c;
      const C disc = std::sqrt(tr * tr * 0.25 - det);
      const C l1 = tr * 0.5 + disc, l2 = tr * 0.5 - disc;
      const C mu = std::abs(l1 - d) < std::abs(l2 - d) ? l2 : l1;          // the other one ends up in h(hi, hi)
      C v0 = b, v1 = mu - a;
      if (std::norm(mu - d) + std::norm(c) > std::norm(v0) + std::norm(v1)) { v0 = mu - d; v1 = c; }
      const double vn = std::sqrt(std::norm(v0) + std::norm(v1));
      if (vn > 0.0) {
        const C x = v0 / vn, y = v1 / vn;                                   // G = [x -conj(y); y conj(x)], G e1 = v
        for (int j = lo; j < n; ++j) {                                      // H <- G^H H
          const C t1 = h(lo, j), t2 = h(hi, j);
          h(lo, j) = std::conj(x) * t1 + std::conj(y) * t2;
          h(hi, j) = -y * t1 + x * t2;
        }
        for (int i = 0; i <= hi; ++i) {                                     // H <- H G
          const C t1 = h(i, lo), t2 = h(i, hi);
          h(i, lo) = t1 * x + t2 * y;
          h(i, hi) = -t1 * std::conj(y) + t2 * std::conj(x);
        }
        for (int i = 0; i < n; ++i) {                                       // Q <- Q G
          const C t1 = q(i, lo), t2 = q(i, hi);
          q(i, lo) = t1 * x + t2 * y;
          q(i, hi) = -t1 * std::conj(y) + t2 * std::conj(x);
        }
      }
      h(hi, lo) = C(0.0);
      hi -= 2; iter = 0;
      continue;
    }
    if (++total > 60 * n + 200) return false;
    ++iter;
    // Wilkinson shift: eigenvalue of the trailing 2x2 closer to h(hi, hi); exceptional shifts on stagnation
    C mu;
    if (iter % 11 == 10) {
      mu = h(hi, hi) + C(std::abs(h(hi, hi - 1)), 0.0);
    } else {
      const C a = h(hi - 1, hi - 1), b = h(hi - 1, hi), c = h(hi, hi - 1), d = h(hi, hi);
      const C tr = a + d, det = a * d - b * c;
      const C disc = std::sqrt(tr * tr * 0.25 - det);
      const C l1 = tr * 0.5 + disc, l2 = tr * 0.5 - disc;
      mu = std::abs(l1 - d) < std::abs(l2 - d) ? l1 : l2;
    }
    for (int i = lo; i <= hi; ++i) h(i, i) -= mu;
    for (int k = lo; k < hi; ++k) {          // R = G_{hi-1} ... G_lo (H - mu I)
      const C a = h(k, k), b = h(k + 1, k);
      const double r = std::sqrt(std::norm(a) + std::norm(b));
      if (r == 0.0) { cs[k] = 1.0; sn[k] = C(0.0); continue; }
      if (std::abs(a) == 0.0) { cs[k] = 0.0; sn[k] = std::conj(b) / std::abs(b); }
      else { cs[k] = std::abs(a) / r; sn[k] = (a / std::abs(a)) * std::conj(b) / r; }
      for (int j = k; j < n; ++j) {
        const C t1 = h(k, j), t2 = h(k + 1, j);
        h(k, j) = cs[k] * t1 + sn[k] * t2;
        h(k + 1, j) = -std::conj(sn[k]) * t1 + cs[k] * t2;
      }
      h(k + 1, k) = C(0.0);
    }
    for (int k = lo; k < hi; ++k) {          // H = R G_lo^H ... G_{hi-1}^H, Q <- Q G^H
      const int top = std::min(k + 1, hi);
      for (int i = 0; i <= top; ++i) {
        const C t1 = h(i, k), t2 = h(i, k + 1);
        h(i, k) = cs[k] * t1 + std::conj(sn[k]) * t2;
        h(i, k + 1) = -sn[k] * t1 + cs[k] * t2;
      }
      for (int i = 0; i < n; ++i) {
        const C t1 = q(i, k), t2 = q(i, k + 1);
        q(i, k) = cs[k] * t1 + std::conj(sn[k]) * t2;
        q(i, k + 1) = -sn[k] * t1 + cs[k] * t2;
      }
    }
    for (int i = lo; i <= hi; ++i) h(i, i) += mu;
  }
  // ---- eigenvectors of the triangular factor, transformed back
  w.resize(n);
  X.assign((size_t)n * n, C(0.0));
  double tnorm = 0.0;
  for (int j = 0; j < n; ++j) for (int i = 0; i <= j; ++i) tnorm = std::max(tnorm, std::abs(h(i, j)));
  const double tiny = eps * (tnorm > 0.0 ? tnorm : 1.0);
  std::vector<C> y(n);
  for (int k = 0; k < n; ++k) {
    w[k] = h(k, k);
    for (int i = 0; i < n; ++i) y[i] = C(0.0);
    y[k] = C(1.0);
    for (int j = k - 1; j >= 0; --j) {
      C s(0.0);
      for (int i = j + 1; i <= k; ++i) s += h(j, i) * y[i];
      C den = h(j, j) - w[k];
      if (std::abs(den) < tiny) den = C(tiny);
      y[j] = -s / den;
    }
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) {
      C s(0.0);
      for (int j = 0; j <= k; ++j) s += q(i, j) * y[j];
      X[i + (size_t)n * k] = s;
      nrm += std::norm(s);
    }
    nrm = std::sqrt(nrm);
    for (int i = 0; i < n; ++i) X[i + (size_t)n * k] /= nrm;
  }
  return true;
}

inline bool genEig(int n, const std::vector<double>& A, std::vector<std::complex<double> >& w,
                   std::vector<std::complex<double> >& X) {
  std::vector<std::complex<double> > Ac((size_t)n * n);
  for (size_t i = 0; i < (size_t)n * n; ++i) Ac[i] = std::complex<double>(A[i], 0.0);
  return genEig(n, Ac, w, X);
}

}  // namespace detail

// Thick-restart block Krylov-Schur (Hermitian).  Parameters as set by MxSolver
// (src/MxSolver.cpp:40-48): "Which" (LM), "Block Size", "Num Blocks", "Maximum Restarts",
// "Convergence Tolerance" (relative to |theta|), "Verbosity".
template <typename Scalar, typename MV, typename OP>
class BlockKrylovSchurSolMgr : public SolverManager<Scalar, MV, OP> {
  typedef Teuchos::SerialDenseMatrix<int, Scalar> Dense;
 public:
  BlockKrylovSchurSolMgr(RCP<BasicEigenproblem<Scalar, MV, OP> > problem, Teuchos::ParameterList& pl)
      : prob_(problem) {
    which_ = pl.get("Which", std::string("LM"));
    blockSize_ = pl.get("Block Size", 1);
    numBlocks_ = pl.get("Num Blocks", 20);
    maxRestarts_ = pl.get("Maximum Restarts", 20);
    tol_ = pl.get("Convergence Tolerance", 1.0e-8);
    verb_ = pl.get("Verbosity", 0);
    if (!problem->isProblemSet()) throw std::invalid_argument("BlockKrylovSchurSolMgr: problem not set");
  }
  int getNumIters() const { return iters_; }

  ReturnType solve() override {
    if (ScalarTraits<Scalar>::isComplex) return solveComplex((Scalar*)0);   // Hermitian or not: general complex path
    if (!prob_->isHermitian()) return solveGeneral();
    const int b = blockSize_, nev = prob_->getNEV();
    const int maxdim = std::max(numBlocks_ * b, nev + 2 * b);     // Krylov basis size before a restart
    const int cap = maxdim + b;                                    // projected matrix capacity
    RCP<OP> op = prob_->getOperator(), M = prob_->getM();
    RCP<MV> init = prob_->getInitVec();
    if (init->GetNumberVecs() < b) throw std::invalid_argument("BlockKrylovSchurSolMgr: initial block smaller than the block size");
    std::unique_ptr<MV> V(init->Clone(cap + b)), W(init->Clone(b));
    std::unique_ptr<MV> MW(M.is_null() ? nullptr : init->Clone(b));
    std::vector<double> T((size_t)cap * cap, 0.0);                 // projected operator (symmetric)
    auto Tat = [&](int i, int j) -> double& { return T[i + (size_t)cap * j]; };
    Dense R(b, b);
    // ---- start block: (M-)orthonormalised copy of the initial vectors
    {
      std::unique_ptr<const MV> i0(init->CloneView(range(0, b)));
      W->MvAddMv(ScalarTraits<Scalar>::one(), *i0, ScalarTraits<Scalar>::zero(), *i0);
    }
    if (orthonormalize(*V, 0, *W, MW.get(), M, R) < b)
      throw std::runtime_error("BlockKrylovSchur: start block is rank deficient");
    setCols(*V, 0, *W);
    int k = b;            // basis vectors held; the last b of them have not been multiplied by Op yet
    int restarts = 0, nconv = 0;
    iters_ = 0;
    while (true) {
      // ---- expand: process the active block until the projected matrix has >= maxdim columns
      bool breakdown = false;
      int stable = 0;
      while (k - b < maxdim) {
        {
          std::unique_ptr<const MV> Vact(V->CloneView(range(k - b, b)));
          op->Apply(*Vact, *W);
        }
        ++iters_;
        // H = V^H (M) W together with the size of Op*v before orthogonalisation (the breakdown reference), one pass:
        // [V W]^H (M) W; then W -= V H  (classical Gram-Schmidt, twice)
        std::vector<double> scale(b);
        Dense h(k, b), h2(k, b);
        std::unique_ptr<const MV> Vk(V->CloneView(range(0, k)));
        innerProdNorms(*Vk, *W, MW.get(), M, h, scale);
        W->MvTimesMatAddMv(-ScalarTraits<Scalar>::one(), *Vk, h, ScalarTraits<Scalar>::one());
        innerProd(*Vk, *W, MW.get(), M, h2);
        W->MvTimesMatAddMv(-ScalarTraits<Scalar>::one(), *Vk, h2, ScalarTraits<Scalar>::one());
        for (int j = 0; j < b; ++j)
          for (int i = 0; i < k; ++i) {
            const double v = ScalarTraits<Scalar>::real(h(i, j)) + ScalarTraits<Scalar>::real(h2(i, j));
            Tat(i, k - b + j) = v;
            Tat(k - b + j, i) = v;
          }
        const int rank = orthonormalize(*V, k, *W, MW.get(), M, R, &scale);   // W = Q R
        setCols(*V, k, *W);
        k += b;
        if (rank < b) { breakdown = true; break; }
        // Anasazi tests convergence every step ("Step Size" = 1), not only when the basis is full: the Ritz pairs of
        // the projected matrix so far and their residual estimates ||R z_last|| cost microseconds, an Op application
        // is two preconditioned solves (43 -> 34 applications on the pillbox).  The wanted pairs must hold for a few
        // consecutive steps: a single-vector Krylov space sees the second copy of a degenerate mode (TE111 of the
        // pillbox) only once the inexact inner solves have let it in - at 32^3, nev 3 the first three converged values
        // are {TM010, TE111, the mode above} after 20 applications and the missing TE111 copy enters two steps later,
        // which resets the count.  And the first basis is always filled before any early test: highly degenerate
        // spectra (the periodic vacuum box: a lowest multiplet of 12) need that long for their copies to show up.
        {
          static const int guard = getenv("MX_KS_GUARD") ? atoi(getenv("MX_KS_GUARD")) : 3;
          if (restarts > 0 && k - b >= nev && k - b < maxdim && ritzConverged(T, cap, k - b, b, R, nev)) { if (++stable > guard) break; }
          else stable = 0;
        }
      }
      const int m = k - b;     // columns whose image under Op is known
      // ---- Rayleigh-Ritz on the m x m projected matrix
      std::vector<double> A((size_t)m * m);
      for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) A[i + (size_t)m * j] = 0.5 * (Tat(i, j) + Tat(j, i));
      std::vector<double> w, Z;
      detail::symEig(m, A, w, Z);
      std::vector<int> order(m);
      std::iota(order.begin(), order.end(), 0);
      sortRitz(w, order);
      // residual of Ritz pair c: || R * z[m-b:m] || (coupling of the last processed block to Q)
      std::vector<double> theta(m), resid(m);
      for (int c = 0; c < m; ++c) {
        const int i = order[c];
        theta[c] = w[i];
        double r2 = 0.0;
        for (int r = 0; r < b; ++r) {
          double sum = 0.0;
          for (int q = 0; q < b; ++q) sum += ScalarTraits<Scalar>::real(R(r, q)) * Z[(m - b + q) + (size_t)m * i];
          r2 += sum * sum;
        }
        resid[c] = std::sqrt(r2);
      }
      nconv = 0;
      while (nconv < std::min(nev, m) && resid[nconv] <= tol_ * std::fabs(theta[nconv])) ++nconv;
      if (verb_ > 0) {
        std::cout << "BlockKrylovSchur: restart " << restarts << ", basis " << m << ", converged " << nconv << "/" << nev << "\n";
        for (int c = 0; c < std::min(nev, m); ++c)
          std::cout << "   theta = " << theta[c] << "   residual = " << resid[c] << "\n";
      }
      const bool done = nconv >= nev || restarts >= maxRestarts_ || breakdown;
      int keep = done ? std::min(nev, m) : std::max(nev + b, (nev + m) / 2);
      if (!done && keep > m - b) keep = m - b;
      if (keep < 1) keep = 1;
      // ---- Ritz vectors of the kept part: Y = V[:, 0:m] * Z[:, order[0:keep]]
      Dense Zk(m, keep);
      for (int c = 0; c < keep; ++c) for (int i = 0; i < m; ++i) Zk(i, c) = Z[i + (size_t)m * order[c]];
      std::unique_ptr<MV> Y(init->Clone(keep));
      {
        std::unique_ptr<const MV> Vm(V->CloneView(range(0, m)));
        Y->MvTimesMatAddMv(ScalarTraits<Scalar>::one(), *Vm, Zk, ScalarTraits<Scalar>::zero());
      }
      if (done) {
        Eigensolution<Scalar, MV> sol;
        sol.numVecs = keep;
        sol.Evecs = RCP<MV>(Y.release());
        for (int c = 0; c < sol.numVecs; ++c) {
          Value<double> v; v.realpart = theta[c]; v.imagpart = 0.0;
          sol.Evals.push_back(v);
          sol.index.push_back(0);
        }
        prob_->setSolution(sol);
        numConverged_ = nconv;
        residuals_.assign(resid.begin(), resid.begin() + keep);
        return nconv >= nev ? Converged : Unconverged;
      }
      // ---- thick restart: V <- [Y, Q]; T <- diag(theta); the coupling of Q to Y is recomputed
      //      as inner products when Q is processed next
      std::unique_ptr<MV> Q(V->CloneCopy(range(m, b)));
      setCols(*V, 0, *Y);
      setCols(*V, keep, *Q);
      std::fill(T.begin(), T.end(), 0.0);
      for (int c = 0; c < keep; ++c) Tat(c, c) = theta[c];
      k = keep + b;
      ++restarts;
    }
  }

  // Non-Hermitian case (dielectrics: the FIT inverse permittivity is not symmetric, src/MxSolver.cpp:68-73
  // leaves the problem non-Hermitian).  Block Arnoldi A V_m = V_m H_m + Q R E^T; Ritz pairs from the
  // complex eigen-decomposition of H_m; restart on a real orthonormal basis Y of the wanted Ritz vectors'
  // span (an invariant subspace of H_m, conjugate pairs kept together):  A (V_m Y) = (V_m Y) S + Q (R Y_last),
  // S = Y^T H_m Y - a Krylov decomposition in Stewart's sense, continued by ordinary Arnoldi steps.
  ReturnType solveGeneral() {
    typedef std::complex<double> C;
    const int b = blockSize_, nev = prob_->getNEV();
    const int maxdim = std::max(numBlocks_ * b, nev + 2 * b + 1);
    const int cap = maxdim + b;
    RCP<OP> op = prob_->getOperator(), M = prob_->getM();
    RCP<MV> init = prob_->getInitVec();
    if (init->GetNumberVecs() < b) throw std::invalid_argument("BlockKrylovSchurSolMgr: initial block smaller than the block size");
    std::unique_ptr<MV> V(init->Clone(cap + b)), W(init->Clone(b));
    std::unique_ptr<MV> MW(M.is_null() ? nullptr : init->Clone(b));
    const int ldh = cap + b;
    std::vector<double> H((size_t)ldh * cap, 0.0);               // (k x m) block Hessenberg incl. the coupling rows
    auto Hat = [&](int i, int j) -> double& { return H[i + (size_t)ldh * j]; };
    Dense R(b, b);
    {
      std::unique_ptr<const MV> i0(init->CloneView(range(0, b)));
      W->MvAddMv(ScalarTraits<Scalar>::one(), *i0, ScalarTraits<Scalar>::zero(), *i0);
    }
    if (orthonormalize(*V, 0, *W, MW.get(), M, R) < b)
      throw std::runtime_error("BlockKrylovSchur: start block is rank deficient");
    setCols(*V, 0, *W);
    int k = b, restarts = 0, nconv = 0;
    iters_ = 0;
    while (true) {
      bool breakdown = false;
      int stable = 0;
      while (k - b < maxdim) {
        {
          std::unique_ptr<const MV> Vact(V->CloneView(range(k - b, b)));
          op->Apply(*Vact, *W);
        }
        ++iters_;
        // H = V^H (M) W together with the size of Op*v before orthogonalisation (the breakdown reference), one pass:
        // [V W]^H (M) W; then W -= V H  (classical Gram-Schmidt, twice)
        std::vector<double> scale(b);
        Dense h(k, b), h2(k, b);
        std::unique_ptr<const MV> Vk(V->CloneView(range(0, k)));
        innerProdNorms(*Vk, *W, MW.get(), M, h, scale);
        W->MvTimesMatAddMv(-ScalarTraits<Scalar>::one(), *Vk, h, ScalarTraits<Scalar>::one());
        innerProd(*Vk, *W, MW.get(), M, h2);
        W->MvTimesMatAddMv(-ScalarTraits<Scalar>::one(), *Vk, h2, ScalarTraits<Scalar>::one());
        for (int j = 0; j < b; ++j)
          for (int i = 0; i < k; ++i)
            Hat(i, k - b + j) = ScalarTraits<Scalar>::real(h(i, j)) + ScalarTraits<Scalar>::real(h2(i, j));
        const int rank = orthonormalize(*V, k, *W, MW.get(), M, R, &scale);
        for (int j = 0; j < b; ++j) for (int r = 0; r < b; ++r) Hat(k + r, k - b + j) = ScalarTraits<Scalar>::real(R(r, j));
        setCols(*V, k, *W);
        k += b;
        if (rank < b) { breakdown = true; break; }
        // convergence tested at every step once the first basis has been filled, as in the Hermitian path (solve())
        {
          static const int guard = getenv("MX_KS_GUARD") ? atoi(getenv("MX_KS_GUARD")) : 3;
          if (restarts > 0 && k - b >= nev && k - b < maxdim && ritzConvergedGeneral(H, ldh, k - b, b, nev)) { if (++stable > guard) break; }
          else stable = 0;
        }
      }
      const int m = k - b;
      std::vector<double> A((size_t)m * m);
      for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) A[i + (size_t)m * j] = Hat(i, j);
      std::vector<C> w, X;
      if (!detail::genEig(m, A, w, X)) {
        if (const char* dump = std::getenv("MXH_DUMP_DENSE")) {      // diagnostics: the projected matrix, column-major
          if (FILE* f = std::fopen(dump, "wb")) { std::fwrite(A.data(), sizeof(A[0]), A.size(), f); std::fclose(f); }
        }
        size_t bad = 0;
        for (size_t i = 0; i < A.size(); ++i) bad += std::isfinite(std::abs(A[i])) ? 0 : 1;
        throw std::runtime_error("BlockKrylovSchur: dense QR iteration did not converge (projected matrix " + std::to_string(m) + " x " +
                                 std::to_string(m) + ", " + std::to_string(bad) + " non-finite entries, restart " + std::to_string(restarts) +
                                 (breakdown ? ", after a breakdown of the block recurrence)" : ")"));
      }
      std::vector<int> order(m);
      std::iota(order.begin(), order.end(), 0);
      sortRitzComplex(w, order);
      // residual of Ritz pair c: || (coupling rows of H) * x[m-b:m] ||
      std::vector<double> resid(m);
      for (int c = 0; c < m; ++c) {
        const int i = order[c];
        double r2 = 0.0;
        for (int r = 0; r < b; ++r) {
          C sum(0.0);
          for (int qq = 0; qq < b; ++qq) sum += Hat(m + r, m - b + qq) * X[(m - b + qq) + (size_t)m * i];
          r2 += std::norm(sum);
        }
        resid[c] = std::sqrt(r2);
      }
      nconv = 0;
      while (nconv < std::min(nev, m) && resid[nconv] <= tol_ * std::abs(w[order[nconv]])) ++nconv;
      if (verb_ > 0) {
        std::cout << "BlockKrylovSchur (general): restart " << restarts << ", basis " << m << ", converged " << nconv << "/" << nev << "\n";
        for (int c = 0; c < std::min(nev, m); ++c)
          std::cout << "   theta = " << w[order[c]] << "   residual = " << resid[c] << "\n";
      }
      const bool done = nconv >= nev || restarts >= maxRestarts_ || breakdown;
      int want = done ? std::min(nev, m) : std::max(nev + b, (nev + m) / 2);
      if (!done && want > m - b - 1) want = m - b - 1;
      if (want < 1) want = 1;
      // ---- real basis of the wanted Ritz vectors: x for a real Ritz value, (Re x, Im x) for a conjugate pair
      const double imtol = 1e-12;
      std::vector<std::vector<double> > cols;     // raw real columns (length m)
      std::vector<int> colIndex;                  // Anasazi index: 0 real, +1 / -1 real / imaginary part of a pair
      std::vector<C> colTheta;
      std::vector<double> colResid;
      std::vector<char> used(m, 0);
      for (int c = 0; c < m && (int)cols.size() < want; ++c) {
        const int i = order[c];
        if (used[i]) continue;
        used[i] = 1;
        const bool real = std::fabs(w[i].imag()) <= imtol * std::max(std::abs(w[i]), 1e-300);
        const C* x = &X[(size_t)m * i];
        if (real) {
          int big = 0;
          for (int r = 1; r < m; ++r) if (std::abs(x[r]) > std::abs(x[big])) big = r;
          const C ph = std::conj(x[big]) / std::abs(x[big]);       // rotate to a real vector
          std::vector<double> col(m);
          for (int r = 0; r < m; ++r) col[r] = (x[r] * ph).real();
          cols.push_back(col); colIndex.push_back(0); colTheta.push_back(C(w[i].real(), 0.0)); colResid.push_back(resid[c]);
        } else {
          // its conjugate partner is another Ritz value: mark it used so the pair enters once
          int partner = -1;
          double best = 1e300;
          for (int j = 0; j < m; ++j)
            if (!used[j]) { const double dd = std::abs(w[j] - std::conj(w[i])); if (dd < best) { best = dd; partner = j; } }
          if (partner >= 0) used[partner] = 1;
          std::vector<double> re(m), im(m);
          for (int r = 0; r < m; ++r) { re[r] = x[r].real(); im[r] = x[r].imag(); }
          const C th = w[i].imag() > 0 ? w[i] : std::conj(w[i]);
          if (w[i].imag() < 0) for (int r = 0; r < m; ++r) im[r] = -im[r];
          cols.push_back(re); colIndex.push_back(1); colTheta.push_back(th); colResid.push_back(resid[c]);
          cols.push_back(im); colIndex.push_back(-1); colTheta.push_back(std::conj(th)); colResid.push_back(resid[c]);
        }
      }
      const int keep = (int)cols.size();
      if (done) {
        Dense Zk(m, keep);
        for (int c = 0; c < keep; ++c) for (int i = 0; i < m; ++i) Zk(i, c) = cols[c][i];
        std::unique_ptr<MV> Yv(init->Clone(keep));
        {
          std::unique_ptr<const MV> Vm(V->CloneView(range(0, m)));
          Yv->MvTimesMatAddMv(ScalarTraits<Scalar>::one(), *Vm, Zk, ScalarTraits<Scalar>::zero());
        }
        Eigensolution<Scalar, MV> sol;
        sol.numVecs = keep;
        sol.Evecs = RCP<MV>(Yv.release());
        for (int c = 0; c < keep; ++c) {
          Value<double> v; v.realpart = colTheta[c].real(); v.imagpart = colTheta[c].imag();
          sol.Evals.push_back(v);
          sol.index.push_back(colIndex[c]);
        }
        prob_->setSolution(sol);
        numConverged_ = nconv;
        residuals_ = colResid;
        return nconv >= nev ? Converged : Unconverged;
      }
      // ---- orthonormalise the columns (modified Gram-Schmidt, twice) -> Y (m x keep)
      std::vector<std::vector<double> > Y = cols;
      for (int c = 0; c < keep; ++c) {
        for (int pass = 0; pass < 2; ++pass)
          for (int p = 0; p < c; ++p) {
            double d = 0.0;
            for (int i = 0; i < m; ++i) d += Y[p][i] * Y[c][i];
            for (int i = 0; i < m; ++i) Y[c][i] -= d * Y[p][i];
          }
        double nrm = 0.0;
        for (int i = 0; i < m; ++i) nrm += Y[c][i] * Y[c][i];
        nrm = std::sqrt(nrm);
        if (!(nrm > 1e-13)) throw std::runtime_error("BlockKrylovSchur: wanted Ritz vectors are linearly dependent");
        for (int i = 0; i < m; ++i) Y[c][i] /= nrm;
      }
      // S = Y^T H_m Y and the new coupling rows (R E^T) Y
      std::vector<double> HY((size_t)m * keep, 0.0), S((size_t)keep * keep, 0.0), Bt((size_t)b * keep, 0.0);
      for (int c = 0; c < keep; ++c)
        for (int j = 0; j < m; ++j) {
          const double yj = Y[c][j];
          if (yj == 0.0) continue;
          for (int i = 0; i < m; ++i) HY[i + (size_t)m * c] += Hat(i, j) * yj;
        }
      for (int c = 0; c < keep; ++c)
        for (int p = 0; p < keep; ++p) {
          double d = 0.0;
          for (int i = 0; i < m; ++i) d += Y[p][i] * HY[i + (size_t)m * c];
          S[p + (size_t)keep * c] = d;
        }
      for (int c = 0; c < keep; ++c)
        for (int r = 0; r < b; ++r) {
          double d = 0.0;
          for (int qq = 0; qq < b; ++qq) d += Hat(m + r, m - b + qq) * Y[c][m - b + qq];
          Bt[r + (size_t)b * c] = d;
        }
      Dense Zk(m, keep);
      for (int c = 0; c < keep; ++c) for (int i = 0; i < m; ++i) Zk(i, c) = Y[c][i];
      std::unique_ptr<MV> Yv(init->Clone(keep));
      {
        std::unique_ptr<const MV> Vm(V->CloneView(range(0, m)));
        Yv->MvTimesMatAddMv(ScalarTraits<Scalar>::one(), *Vm, Zk, ScalarTraits<Scalar>::zero());
      }
      std::unique_ptr<MV> Qb(V->CloneCopy(range(m, b)));
      setCols(*V, 0, *Yv);
      setCols(*V, keep, *Qb);
      std::fill(H.begin(), H.end(), 0.0);
      for (int c = 0; c < keep; ++c) {
        for (int p = 0; p < keep; ++p) Hat(p, c) = S[p + (size_t)keep * c];
        for (int r = 0; r < b; ++r) Hat(keep + r, c) = Bt[r + (size_t)b * c];
      }
      k = keep + b;
      ++restarts;
    }
  }

  ReturnType solveComplex(double*) { throw std::logic_error("solveComplex on real scalars"); }

  // Complex scalars (Bloch phase shifts, lossy materials): block Arnoldi with a complex projected matrix,
  // Ritz pairs from its eigen-decomposition, restart on an orthonormal basis of the wanted Ritz vectors
  // (A (V Y) = (V Y) S + Q (R Y_last), S = Y^H H Y).  Used for Hermitian problems as well.
  ReturnType solveComplex(std::complex<double>*) {
    typedef std::complex<double> C;
    const int b = blockSize_, nev = prob_->getNEV();
    const int maxdim = std::max(numBlocks_ * b, nev + 2 * b + 1);
    const int cap = maxdim + b;
    RCP<OP> op = prob_->getOperator(), M = prob_->getM();
    RCP<MV> init = prob_->getInitVec();
    if (init->GetNumberVecs() < b) throw std::invalid_argument("BlockKrylovSchurSolMgr: initial block smaller than the block size");
    std::unique_ptr<MV> V(init->Clone(cap + b)), W(init->Clone(b));
    std::unique_ptr<MV> MW(M.is_null() ? nullptr : init->Clone(b));
    const int ldh = cap + b;
    std::vector<C> H((size_t)ldh * cap, C(0.0));
    auto Hat = [&](int i, int j) -> C& { return H[i + (size_t)ldh * j]; };
    Dense R(b, b);
    {
      std::unique_ptr<const MV> i0(init->CloneView(range(0, b)));
      W->MvAddMv(ScalarTraits<Scalar>::one(), *i0, ScalarTraits<Scalar>::zero(), *i0);
    }
    if (orthonormalize(*V, 0, *W, MW.get(), M, R) < b)
      throw std::runtime_error("BlockKrylovSchur: start block is rank deficient");
    setCols(*V, 0, *W);
    int k = b, restarts = 0, nconv = 0;
    iters_ = 0;
    while (true) {
      bool breakdown = false;
      int stable = 0;
      while (k - b < maxdim) {
        {
          std::unique_ptr<const MV> Vact(V->CloneView(range(k - b, b)));
          op->Apply(*Vact, *W);
        }
        ++iters_;
        // H = V^H (M) W together with the size of Op*v before orthogonalisation (the breakdown reference), one pass:
        // [V W]^H (M) W; then W -= V H  (classical Gram-Schmidt, twice)
        std::vector<double> scale(b);
        Dense h(k, b), h2(k, b);
        std::unique_ptr<const MV> Vk(V->CloneView(range(0, k)));
        innerProdNorms(*Vk, *W, MW.get(), M, h, scale);
        W->MvTimesMatAddMv(-ScalarTraits<Scalar>::one(), *Vk, h, ScalarTraits<Scalar>::one());
        innerProd(*Vk, *W, MW.get(), M, h2);
        W->MvTimesMatAddMv(-ScalarTraits<Scalar>::one(), *Vk, h2, ScalarTraits<Scalar>::one());
        for (int j = 0; j < b; ++j)
          for (int i = 0; i < k; ++i) Hat(i, k - b + j) = C(h(i, j)) + C(h2(i, j));
        const int rank = orthonormalize(*V, k, *W, MW.get(), M, R, &scale);
        for (int j = 0; j < b; ++j) for (int r = 0; r < b; ++r) Hat(k + r, k - b + j) = C(R(r, j));
        setCols(*V, k, *W);
        k += b;
        if (rank < b) { breakdown = true; break; }
        // convergence tested at every step once the first basis has been filled, as in solve()
        {
          static const int guard = getenv("MX_KS_GUARD") ? atoi(getenv("MX_KS_GUARD")) : 3;
          if (restarts > 0 && k - b >= nev && k - b < maxdim && ritzConvergedComplex(H, ldh, k - b, b, nev)) { if (++stable > guard) break; }
          else stable = 0;
        }
      }
      const int m = k - b;
      std::vector<C> A((size_t)m * m);
      for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) A[i + (size_t)m * j] = Hat(i, j);
      std::vector<C> w, X;
      if (!detail::genEig(m, A, w, X)) {
        if (const char* dump = std::getenv("MXH_DUMP_DENSE")) {      // diagnostics: the projected matrix, column-major
          if (FILE* f = std::fopen(dump, "wb")) { std::fwrite(A.data(), sizeof(A[0]), A.size(), f); std::fclose(f); }
        }
        size_t bad = 0;
        for (size_t i = 0; i < A.size(); ++i) bad += std::isfinite(std::abs(A[i])) ? 0 : 1;
        throw std::runtime_error("BlockKrylovSchur: dense QR iteration did not converge (projected matrix " + std::to_string(m) + " x " +
                                 std::to_string(m) + ", " + std::to_string(bad) + " non-finite entries, restart " + std::to_string(restarts) +
                                 (breakdown ? ", after a breakdown of the block recurrence)" : ")"));
      }
      std::vector<int> order(m);
      std::iota(order.begin(), order.end(), 0);
      sortRitzComplex(w, order);
      std::vector<double> resid(m);
      for (int c = 0; c < m; ++c) {
        const int i = order[c];
        double r2 = 0.0;
        for (int r = 0; r < b; ++r) {
          C sum(0.0);
          for (int qq = 0; qq < b; ++qq) sum += Hat(m + r, m - b + qq) * X[(m - b + qq) + (size_t)m * i];
          r2 += std::norm(sum);
        }
        resid[c] = std::sqrt(r2);
      }
      nconv = 0;
      while (nconv < std::min(nev, m) && resid[nconv] <= tol_ * std::abs(w[order[nconv]])) ++nconv;
      if (verb_ > 0) {
        std::cout << "BlockKrylovSchur (complex): restart " << restarts << ", basis " << m << ", converged " << nconv << "/" << nev << "\n";
        for (int c = 0; c < std::min(nev, m); ++c)
          std::cout << "   theta = " << w[order[c]] << "   residual = " << resid[c] << "\n";
      }
      const bool done = nconv >= nev || restarts >= maxRestarts_ || breakdown;
      int keep = done ? std::min(nev, m) : std::max(nev + b, (nev + m) / 2);
      if (!done && keep > m - b) keep = m - b;
      if (keep < 1) keep = 1;
      // orthonormal basis of the wanted Ritz vectors (modified Gram-Schmidt, twice)
      std::vector<std::vector<C> > Y((size_t)keep, std::vector<C>((size_t)m));
      for (int c = 0; c < keep; ++c) for (int i = 0; i < m; ++i) Y[c][i] = X[i + (size_t)m * order[c]];
      if (!done)
        for (int c = 0; c < keep; ++c) {
          for (int pass = 0; pass < 2; ++pass)
            for (int p = 0; p < c; ++p) {
              C d(0.0);
              for (int i = 0; i < m; ++i) d += std::conj(Y[p][i]) * Y[c][i];
              for (int i = 0; i < m; ++i) Y[c][i] -= d * Y[p][i];
            }
          double nrm = 0.0;
          for (int i = 0; i < m; ++i) nrm += std::norm(Y[c][i]);
          nrm = std::sqrt(nrm);
          if (!(nrm > 1e-13)) throw std::runtime_error("BlockKrylovSchur: wanted Ritz vectors are linearly dependent");
          for (int i = 0; i < m; ++i) Y[c][i] /= nrm;
        }
      Dense Zk(m, keep);
      for (int c = 0; c < keep; ++c) for (int i = 0; i < m; ++i) Zk(i, c) = fromComplex(Y[c][i], (Scalar*)0);
      std::unique_ptr<MV> Yv(init->Clone(keep));
      {
        std::unique_ptr<const MV> Vm(V->CloneView(range(0, m)));
        Yv->MvTimesMatAddMv(ScalarTraits<Scalar>::one(), *Vm, Zk, ScalarTraits<Scalar>::zero());
      }
      if (done) {
        Eigensolution<Scalar, MV> sol;
        sol.numVecs = keep;
        sol.Evecs = RCP<MV>(Yv.release());
        for (int c = 0; c < keep; ++c) {
          Value<double> v; v.realpart = w[order[c]].real(); v.imagpart = w[order[c]].imag();
          sol.Evals.push_back(v);
          sol.index.push_back(0);
        }
        prob_->setSolution(sol);
        numConverged_ = nconv;
        residuals_.assign(resid.begin(), resid.begin() + keep);
        return nconv >= nev ? Converged : Unconverged;
      }
      std::vector<C> HY((size_t)m * keep, C(0.0)), S((size_t)keep * keep, C(0.0)), Bt((size_t)b * keep, C(0.0));
      for (int c = 0; c < keep; ++c)
        for (int j = 0; j < m; ++j) {
          const C yj = Y[c][j];
          for (int i = 0; i < m; ++i) HY[i + (size_t)m * c] += Hat(i, j) * yj;
        }
      for (int c = 0; c < keep; ++c)
        for (int p = 0; p < keep; ++p) {
          C d(0.0);
          for (int i = 0; i < m; ++i) d += std::conj(Y[p][i]) * HY[i + (size_t)m * c];
          S[p + (size_t)keep * c] = d;
        }
      for (int c = 0; c < keep; ++c)
        for (int r = 0; r < b; ++r) {
          C d(0.0);
          for (int qq = 0; qq < b; ++qq) d += Hat(m + r, m - b + qq) * Y[c][m - b + qq];
          Bt[r + (size_t)b * c] = d;
        }
      std::unique_ptr<MV> Qb(V->CloneCopy(range(m, b)));
      setCols(*V, 0, *Yv);
      setCols(*V, keep, *Qb);
      std::fill(H.begin(), H.end(), C(0.0));
      for (int c = 0; c < keep; ++c) {
        for (int p = 0; p < keep; ++p) Hat(p, c) = S[p + (size_t)keep * c];
        for (int r = 0; r < b; ++r) Hat(keep + r, c) = Bt[r + (size_t)b * c];
      }
      k = keep + b;
      ++restarts;
    }
  }

  static double fromComplex(const std::complex<double>& z, double*) { return z.real(); }
  static std::complex<double> fromComplex(const std::complex<double>& z, std::complex<double>*) { return z; }

  int getNumConverged() const { return numConverged_; }

 protected:
  static std::vector<int> range(int start, int n) {
    std::vector<int> r(n);
    std::iota(r.begin(), r.end(), start);
    return r;
  }
  static void setCols(MV& V, int start, const MV& A) { V.SetBlock(A, range(start, A.GetNumberVecs())); }

  // C = A^H (M) B
  static void innerProd(const MV& A, const MV& B, MV* MB, const RCP<OP>& M, Dense& C) {
    if (M.is_null()) { B.MvTransMv(ScalarTraits<Scalar>::one(), A, C); return; }
    M->Apply(B, *MB);
    MB->MvTransMv(ScalarTraits<Scalar>::one(), A, C);
  }

  // (M-)orthonormalise the block W against itself by Cholesky-free Gram-Schmidt over its b
  // columns (twice); W = Q R.  Returns the numerical rank; deficient columns are re-randomised
  // and orthogonalised against V[:, 0:kprev] so the basis stays full.
  int orthonormalize(MV& V, int kprev, MV& W, MV* MW, const RCP<OP>& M, Dense& R,
                     const std::vector<double>* scale = nullptr) {
    const int b = W.GetNumberVecs();
    R.shape(b, b);
    int rank = b;
    for (int j = 0; j < b; ++j) {
      std::unique_ptr<MV> wj(W.CloneViewNonConst(range(j, 1)));
      std::unique_ptr<MV> mwj(MW ? MW->CloneViewNonConst(range(0, 1)) : nullptr);
      double norm0 = scale ? (*scale)[j] : mnorm(*wj, mwj.get(), M);
      for (int pass = 0; pass < 2 && j > 0; ++pass) {
        std::unique_ptr<const MV> Wp(W.CloneView(range(0, j)));
        Dense c(j, 1);
        innerProd(*Wp, *wj, mwj.get(), M, c);
        wj->MvTimesMatAddMv(-ScalarTraits<Scalar>::one(), *Wp, c, ScalarTraits<Scalar>::one());
        for (int i = 0; i < j; ++i) R(i, j) += c(i, 0);
      }
      double nrm = mnorm(*wj, mwj.get(), M);
      if (!(nrm > 1e-12 * std::max(norm0, 1e-300))) {     // breakdown: replace by a random direction
        --rank;
        wj->MvRandom();
        if (kprev > 0) {
          std::unique_ptr<const MV> Vk(V.CloneView(range(0, kprev)));
          for (int pass = 0; pass < 2; ++pass) {
            Dense c(kprev, 1);
            innerProd(*Vk, *wj, mwj.get(), M, c);
            wj->MvTimesMatAddMv(-ScalarTraits<Scalar>::one(), *Vk, c, ScalarTraits<Scalar>::one());
          }
        }
        if (j > 0) {
          std::unique_ptr<const MV> Wp(W.CloneView(range(0, j)));
          Dense c(j, 1);
          innerProd(*Wp, *wj, mwj.get(), M, c);
          wj->MvTimesMatAddMv(-ScalarTraits<Scalar>::one(), *Wp, c, ScalarTraits<Scalar>::one());
        }
        nrm = mnorm(*wj, mwj.get(), M);
        R(j, j) = ScalarTraits<Scalar>::zero();
      } else {
        R(j, j) = Scalar(nrm);
      }
      wj->MvScale(Scalar(1.0 / nrm));
    }
    return rank;
  }

  // C = A^H (M) B and norms[j] = sqrt(B_j^H (M) B_j) from one pass over the rows: [A B]^H (M B)
  static void innerProdNorms(const MV& A, const MV& B, MV* MB, const RCP<OP>& M, Dense& C, std::vector<double>& norms) {
    const int k = A.GetNumberVecs(), b = B.GetNumberVecs();
    Dense ext(k + b, b);
    if (M.is_null()) B.MvTransMv2(ScalarTraits<Scalar>::one(), A, B, ext);
    else { M->Apply(B, *MB); MB->MvTransMv2(ScalarTraits<Scalar>::one(), A, B, ext); }
    for (int j = 0; j < b; ++j) {
      for (int i = 0; i < k; ++i) C(i, j) = ext(i, j);
      const double v = ScalarTraits<Scalar>::real(ext(k + j, j));
      norms[j] = v > 0 ? std::sqrt(v) : 0.0;
    }
  }

  static double mnorm(const MV& w, MV* mw, const RCP<OP>& M) {
    if (M.is_null()) {
      std::vector<double> n(1);
      w.MvNorm(n);
      return n[0];
    }
    M->Apply(w, *mw);
    std::vector<Scalar> d(1);
    mw->MvDot(w, d);
    const double v = ScalarTraits<Scalar>::real(d[0]);
    return v > 0 ? std::sqrt(v) : 0.0;
  }

  // do the first nev Ritz pairs of the leading m x m block of the projected matrix T (leading dimension ld) meet the
  // tolerance, by the residual estimates of the Krylov relation?
  bool ritzConverged(const std::vector<double>& T, int ld, int m, int b, const Dense& R, int nev) const {
    std::vector<double> A((size_t)m * m), w, Z;
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) A[i + (size_t)m * j] = 0.5 * (T[i + (size_t)ld * j] + T[j + (size_t)ld * i]);
    detail::symEig(m, A, w, Z);
    std::vector<int> order(m);
    std::iota(order.begin(), order.end(), 0);
    sortRitz(w, order);
    for (int c = 0; c < nev; ++c) {
      const int i = order[c];
      double r2 = 0.0;
      for (int r = 0; r < b; ++r) {
        double sum = 0.0;
        for (int q = 0; q < b; ++q) sum += ScalarTraits<Scalar>::real(R(r, q)) * Z[(m - b + q) + (size_t)m * i];
        r2 += sum * sum;
      }
      if (!(std::sqrt(r2) <= tol_ * std::fabs(w[i]))) return false;
    }
    return true;
  }

  // the same for the general (non-Hermitian, real) recurrence: H holds the m x m block Hessenberg matrix and, in rows
  // m .. m + b - 1, the coupling of the next block
  bool ritzConvergedGeneral(const std::vector<double>& H, int ld, int m, int b, int nev) const {
    typedef std::complex<double> C;
    std::vector<double> A((size_t)m * m);
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) A[i + (size_t)m * j] = H[i + (size_t)ld * j];
    std::vector<C> w, X;
    if (!detail::genEig(m, A, w, X)) return false;
    std::vector<int> order(m);
    std::iota(order.begin(), order.end(), 0);
    sortRitzComplex(w, order);
    for (int c = 0; c < nev; ++c) {
      const int i = order[c];
      double r2 = 0.0;
      for (int r = 0; r < b; ++r) {
        C sum(0.0);
        for (int q = 0; q < b; ++q) sum += H[(m + r) + (size_t)ld * (m - b + q)] * X[(m - b + q) + (size_t)m * i];
        r2 += std::norm(sum);
      }
      if (!(std::sqrt(r2) <= tol_ * std::abs(w[i]))) return false;
    }
    return true;
  }

  // ... and for the complex recurrence
  bool ritzConvergedComplex(const std::vector<std::complex<double> >& H, int ld, int m, int b, int nev) const {
    typedef std::complex<double> C;
    std::vector<C> A((size_t)m * m), w, X;
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) A[i + (size_t)m * j] = H[i + (size_t)ld * j];
    if (!detail::genEig(m, A, w, X)) return false;
    std::vector<int> order(m);
    std::iota(order.begin(), order.end(), 0);
    sortRitzComplex(w, order);
    for (int c = 0; c < nev; ++c) {
      const int i = order[c];
      double r2 = 0.0;
      for (int r = 0; r < b; ++r) {
        C sum(0.0);
        for (int q = 0; q < b; ++q) sum += H[(m + r) + (size_t)ld * (m - b + q)] * X[(m - b + q) + (size_t)m * i];
        r2 += std::norm(sum);
      }
      if (!(std::sqrt(r2) <= tol_ * std::abs(w[i]))) return false;
    }
    return true;
  }

  void sortRitz(const std::vector<double>& w, std::vector<int>& order) const {
    if (which_ == "LM") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::fabs(w[a]) > std::fabs(w[b]); });
    else if (which_ == "SM") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::fabs(w[a]) < std::fabs(w[b]); });
    else if (which_ == "LR") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] > w[b]; });
    else if (which_ == "SR") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] < w[b]; });
    else throw std::invalid_argument("BlockKrylovSchurSolMgr: unknown 'Which' = " + which_);
  }

  void sortRitzComplex(const std::vector<std::complex<double> >& w, std::vector<int>& order) const {
    if (which_ == "LM") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::abs(w[a]) > std::abs(w[b]); });
    else if (which_ == "SM") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::abs(w[a]) < std::abs(w[b]); });
    else if (which_ == "LR") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a].real() > w[b].real(); });
    else if (which_ == "SR") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a].real() < w[b].real(); });
    else throw std::invalid_argument("BlockKrylovSchurSolMgr: unknown 'Which' = " + which_);
  }

  RCP<BasicEigenproblem<Scalar, MV, OP> > prob_;
  std::string which_;
  int blockSize_, numBlocks_, maxRestarts_, verb_;
  double tol_;
  int iters_ = 0, numConverged_ = 0;
  std::vector<double> residuals_;
};

// Block Davidson for Hermitian problems (Anasazi::BlockDavidsonSolMgr, selected by `eigensolver : type = davidson`,
// reference src/MxSolver.cpp:88-91).  No preconditioner is handed over by the reference (setPrec is never called), so
// the basis grows by the (M-)orthonormalised residuals Op x - theta x of the wanted Ritz pairs that have not converged:
// generalised Davidson with the identity as its correction equation.  Unlike the Krylov-Schur manager it keeps Op V
// next to V (the residuals are formed explicitly) and tests ||Op x - theta x||_M <= tol |theta| on those residuals.
// "Use Locking": a converged Ritz pair stops producing expansion vectors and stays in the basis through restarts
// (soft locking: the pair is never deflated out of the projected problem, so it keeps improving with the others).
template <typename Scalar, typename MV, typename OP>
class BlockDavidsonSolMgr : public BlockKrylovSchurSolMgr<Scalar, MV, OP> {
  typedef BlockKrylovSchurSolMgr<Scalar, MV, OP> Base;
  typedef Teuchos::SerialDenseMatrix<int, Scalar> Dense;
 public:
  BlockDavidsonSolMgr(RCP<BasicEigenproblem<Scalar, MV, OP> > problem, Teuchos::ParameterList& pl) : Base(problem, pl) {
    if (!problem->isHermitian() || ScalarTraits<Scalar>::isComplex)
      throw std::invalid_argument("BlockDavidsonSolMgr: the problem must be Hermitian (real scalars in this build)");
    locking_ = pl.get("Use Locking", false);
  }

  ReturnType solve() override {
    const int b = this->blockSize_, nev = this->prob_->getNEV();
    const int maxdim = std::max(this->numBlocks_ * b, nev + 2 * b);
    RCP<OP> op = this->prob_->getOperator(), M = this->prob_->getM();
    RCP<MV> init = this->prob_->getInitVec();
    if (init->GetNumberVecs() < b) throw std::invalid_argument("BlockDavidsonSolMgr: initial block smaller than the block size");
    std::unique_ptr<MV> V(init->Clone(maxdim + b)), AV(init->Clone(maxdim + b)), W(init->Clone(b));
    std::unique_ptr<MV> MW(M.is_null() ? nullptr : init->Clone(b));
    std::vector<double> H((size_t)maxdim * maxdim, 0.0);
    auto Hat = [&](int i, int j) -> double& { return H[i + (size_t)maxdim * j]; };
    Dense R(b, b);
    {
      std::unique_ptr<const MV> i0(init->CloneView(Base::range(0, b)));
      W->MvAddMv(ScalarTraits<Scalar>::one(), *i0, ScalarTraits<Scalar>::zero(), *i0);
    }
    if (this->orthonormalize(*V, 0, *W, MW.get(), M, R) < b) throw std::runtime_error("BlockDavidson: start block is rank deficient");
    Base::setCols(*V, 0, *W);
    int k = b, restarts = 0, nconv = 0, stable = 0;
    this->iters_ = 0;
    std::vector<double> theta, resid;
    std::vector<int> order;
    std::vector<double> Z;
    while (true) {
      // ---- Op on the newest block, its row / column of the projected matrix H = V^H (M) Op V
      {
        std::unique_ptr<const MV> Vnew(V->CloneView(Base::range(k - b, b)));
        op->Apply(*Vnew, *W);
        Base::setCols(*AV, k - b, *W);
      }
      ++this->iters_;
      {
        std::unique_ptr<const MV> Vk(V->CloneView(Base::range(0, k)));
        Dense h(k, b);
        Base::innerProd(*Vk, *W, MW.get(), M, h);
        for (int j = 0; j < b; ++j)
          for (int i = 0; i < k; ++i) {
            const double v = ScalarTraits<Scalar>::real(h(i, j));
            Hat(i, k - b + j) = v;
            if (i < k - b) Hat(k - b + j, i) = v;
          }
      }
      // ---- Rayleigh-Ritz
      std::vector<double> A((size_t)k * k), w;
      for (int j = 0; j < k; ++j) for (int i = 0; i < k; ++i) A[i + (size_t)k * j] = 0.5 * (Hat(i, j) + Hat(j, i));
      detail::symEig(k, A, w, Z);
      order.resize(k);
      std::iota(order.begin(), order.end(), 0);
      this->sortRitz(w, order);
      const int nw = std::min(nev, k);                       // wanted pairs that exist yet
      theta.assign(nw, 0.0);
      for (int c = 0; c < nw; ++c) theta[c] = w[order[c]];
      // ---- residuals Op x - theta x of every wanted pair (the convergence test needs all of them; two passes over
      //      the basis and its image, the cost of the Gram-Schmidt step of a Krylov-Schur iteration)
      std::unique_ptr<MV> Rb(init->Clone(nw));
      {
        Dense Za(k, nw), ZaT(k, nw);
        for (int c = 0; c < nw; ++c)
          for (int i = 0; i < k; ++i) { Za(i, c) = Z[i + (size_t)k * order[c]]; ZaT(i, c) = Za(i, c) * Scalar(theta[c]); }
        std::unique_ptr<const MV> Vk(V->CloneView(Base::range(0, k))), AVk(AV->CloneView(Base::range(0, k)));
        Rb->MvTimesMatAddMv(ScalarTraits<Scalar>::one(), *AVk, Za, ScalarTraits<Scalar>::zero());
        Rb->MvTimesMatAddMv(-ScalarTraits<Scalar>::one(), *Vk, ZaT, ScalarTraits<Scalar>::one());
      }
      resid.assign(nw, 0.0);
      for (int c = 0; c < nw; ++c) {
        std::unique_ptr<const MV> rc(Rb->CloneView(Base::range(c, 1)));
        std::unique_ptr<MV> mrc(MW ? MW->CloneViewNonConst(Base::range(0, 1)) : nullptr);
        resid[c] = Base::mnorm(*rc, mrc.get(), M);
      }
      // expansion: the first b wanted pairs that have not converged (a converged pair is "locked": it stays in the
      // basis and produces no more directions)
      std::vector<int> active;
      for (int c = 0; c < nw && (int)active.size() < b; ++c)
        if (resid[c] > this->tol_ * std::fabs(theta[c])) active.push_back(c);
      const int na = (int)active.size();
      nconv = 0;
      while (nconv < nw && resid[nconv] <= this->tol_ * std::fabs(theta[nconv])) ++nconv;
      if (this->verb_ > 0) {
        std::cout << "BlockDavidson: iteration " << this->iters_ << ", basis " << k << ", converged " << nconv << "/" << nev << "\n";
        for (int c = 0; c < nw; ++c) std::cout << "   theta = " << theta[c] << "   residual = " << resid[c] << "\n";
      }
      const bool full = k + b > maxdim;
      // (the wanted pairs must hold for a few consecutive iterations, as in the Krylov-Schur manager: the second copy
      //  of a degenerate mode enters late; with every wanted pair converged the expansion block is random, which lets it in)
      static const int guard = getenv("MX_KS_GUARD") ? atoi(getenv("MX_KS_GUARD")) : 3;
      stable = (nconv >= nev && k >= nev && (restarts > 0 || full)) ? stable + 1 : 0;
      const bool done = stable > guard || (full && restarts >= this->maxRestarts_);
      if (done) {
        const int keep = std::min(nev, k);
        Dense Zk(k, keep);
        for (int c = 0; c < keep; ++c) for (int i = 0; i < k; ++i) Zk(i, c) = Z[i + (size_t)k * order[c]];
        std::unique_ptr<MV> Y(init->Clone(keep));
        std::unique_ptr<const MV> Vk(V->CloneView(Base::range(0, k)));
        Y->MvTimesMatAddMv(ScalarTraits<Scalar>::one(), *Vk, Zk, ScalarTraits<Scalar>::zero());
        Eigensolution<Scalar, MV> sol;
        sol.numVecs = keep;
        sol.Evecs = RCP<MV>(Y.release());
        for (int c = 0; c < keep; ++c) {
          Value<double> v; v.realpart = theta[c]; v.imagpart = 0.0;
          sol.Evals.push_back(v);
          sol.index.push_back(0);
        }
        this->prob_->setSolution(sol);
        this->numConverged_ = nconv;
        this->residuals_.assign(resid.begin(), resid.begin() + keep);
        return nconv >= nev ? Converged : Unconverged;
      }
      if (full) {
        // ---- restart on the leading Ritz vectors: V <- V Z_keep, Op V <- Op V Z_keep, H <- diag(theta)
        int keep = std::max(nev + b, (nev + k) / 2);
        if (keep > k - b) keep = k - b;
        if (keep < 1) keep = 1;
        Dense Zk(k, keep);
        for (int c = 0; c < keep; ++c) for (int i = 0; i < k; ++i) Zk(i, c) = Z[i + (size_t)k * order[c]];
        std::unique_ptr<MV> Y(init->Clone(keep)), AY(init->Clone(keep));
        {
          std::unique_ptr<const MV> Vk(V->CloneView(Base::range(0, k))), AVk(AV->CloneView(Base::range(0, k)));
          Y->MvTimesMatAddMv(ScalarTraits<Scalar>::one(), *Vk, Zk, ScalarTraits<Scalar>::zero());
          AY->MvTimesMatAddMv(ScalarTraits<Scalar>::one(), *AVk, Zk, ScalarTraits<Scalar>::zero());
        }
        Base::setCols(*V, 0, *Y);
        Base::setCols(*AV, 0, *AY);
        std::fill(H.begin(), H.end(), 0.0);
        for (int c = 0; c < keep; ++c) Hat(c, c) = w[order[c]];
        k = keep;
        ++restarts;
      }
      // ---- expand by those residuals, (M-)orthogonalised against the basis (classical Gram-Schmidt, twice) and each
      //      other; fewer than b of them: the rest of the block is random (kept orthogonal like the others)
      MV* E = W.get();
      {
        if (na < b) E->MvRandom();
        for (int a = 0; a < na; ++a) {
          std::unique_ptr<const MV> src(Rb->CloneView(Base::range(active[a], 1)));
          E->SetBlock(*src, Base::range(a, 1));
        }
        std::unique_ptr<const MV> Vk(V->CloneView(Base::range(0, k)));
        for (int pass = 0; pass < 2; ++pass) {
          Dense c(k, b);
          Base::innerProd(*Vk, *E, MW.get(), M, c);
          E->MvTimesMatAddMv(-ScalarTraits<Scalar>::one(), *Vk, c, ScalarTraits<Scalar>::one());
        }
      }
      this->orthonormalize(*V, k, *E, MW.get(), M, R);
      Base::setCols(*V, k, *E);
      k += b;
    }
  }

 private:
  bool locking_ = false;
};

}  // namespace Anasazi

#endif
