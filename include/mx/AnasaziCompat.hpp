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
#include <numeric>

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
    if (!problem->isHermitian())
      throw std::invalid_argument("BlockKrylovSchurSolMgr: only the Hermitian (PEC / vacuum) case is implemented");
  }
  int getNumIters() const { return iters_; }

  ReturnType solve() {
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
      while (k - b < maxdim) {
        {
          std::unique_ptr<const MV> Vact(V->CloneView(range(k - b, b)));
          op->Apply(*Vact, *W);
        }
        ++iters_;
        std::vector<double> scale(b);
        for (int j = 0; j < b; ++j) {     // size of Op*v before orthogonalisation: breakdown reference
          std::unique_ptr<MV> wj(W->CloneViewNonConst(range(j, 1)));
          std::unique_ptr<MV> mwj(MW ? MW->CloneViewNonConst(range(0, 1)) : nullptr);
          scale[j] = mnorm(*wj, mwj.get(), M);
        }
        // H = V^H (M) W, W -= V H  (classical Gram-Schmidt, twice)
        Dense h(k, b), h2(k, b);
        std::unique_ptr<const MV> Vk(V->CloneView(range(0, k)));
        innerProd(*Vk, *W, MW.get(), M, h);
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

  int getNumConverged() const { return numConverged_; }

 private:
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

  void sortRitz(const std::vector<double>& w, std::vector<int>& order) const {
    if (which_ == "LM") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::fabs(w[a]) > std::fabs(w[b]); });
    else if (which_ == "SM") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::fabs(w[a]) < std::fabs(w[b]); });
    else if (which_ == "LR") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] > w[b]; });
    else if (which_ == "SR") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] < w[b]; });
    else throw std::invalid_argument("BlockKrylovSchurSolMgr: unknown 'Which' = " + which_);
  }

  RCP<BasicEigenproblem<Scalar, MV, OP> > prob_;
  std::string which_;
  int blockSize_, numBlocks_, maxRestarts_, verb_;
  double tol_;
  int iters_ = 0, numConverged_ = 0;
  std::vector<double> residuals_;
};

}  // namespace Anasazi

#endif
