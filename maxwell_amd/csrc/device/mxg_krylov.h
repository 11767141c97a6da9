// Krylov loops of the inner solves (K10 of SURVEY section 2.2), written on the PUBLIC C ABI of include/mxg.h only:
// every vector operation is an mxg_mv_* call, every product an mxg_csr_apply.  Shared by libmxg.so (mxg_solve.hip,
// which adds the fused GPU paths) and by the CPU backend of the same ABI that bench.py times as the CPU baseline
// (oracle/cpu_backend/mxg_cpu.cpp): "the same solver on the host cores" is literally this code.
// Replaces AztecOO::recursiveIterate + the TrueResidual() loop (reference src/MxMagWaveOp.cpp:874-885,907-913;
// options :326-353: AZ_kspace = "linear solver : basis", AZ_conv = AZ_noscaled: the test is on the unscaled true
// residual 2-norm).  Hooks: apply_prec(r, z): z = M^-1 r;  prec_view(v, z, &zc): the same, *zc = where the result
// sits (the multigrid leaves it in its own vector);  fused(...): a backend's fused CG, -1 = not applicable.
#ifndef MXG_KRYLOV_H
#define MXG_KRYLOV_H

#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

#include "mxg.h"

namespace mxg_krylov {

typedef std::complex<double> cplx;

#define MXK_TRY(call) do { int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

struct Guard {  // frees temporaries on every exit path
  std::vector<mxg_mv*> mvs;
  ~Guard() { for (auto* v : mvs) mxg_mv_destroy(v); }
  int make(mxg_ctx* c, int64_t n, int k, int cx, mxg_mv** out) {
    int rc = mxg_mv_create(c, n, k, cx, out);
    if (rc == 0) mvs.push_back(*out);
    return rc;
  }
  int view(mxg_mv* p, int col0, int n, mxg_mv** out) {
    std::vector<int> cols(n);
    for (int i = 0; i < n; ++i) cols[i] = col0 + i;
    int rc = mxg_mv_view(p, cols.data(), n, out);
    if (rc == 0) mvs.push_back(*out);
    return rc;
  }
};

inline int true_residual(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, const mxg_mv* x, mxg_mv* r, double* nrm) {
  MXK_TRY(mxg_csr_apply(c, A, x, r));
  MXK_TRY(mxg_mv_axpby(1.0, 0.0, b, -1.0, 0.0, r));   // r = b - A x
  return mxg_mv_norm2(r, nrm);
}
// the residual of the initial guess; with MXG_SOLVE_ZERO_GUESS it is b itself (b - A 0 = b to the bit: no product)
inline int initial_residual(mxg_ctx* c, const mxg_csr* A, const mxg_solve_opts* o, const mxg_mv* b, const mxg_mv* x, mxg_mv* r, double* nrm) {
  if (!(o->flags & MXG_SOLVE_ZERO_GUESS)) return true_residual(c, A, b, x, r, nrm);
  MXK_TRY(mxg_mv_copy(r, b));
  return mxg_mv_norm2(r, nrm);
}

struct Shape { int64_t nrows = 0; int cx = 0, k = 0; };
inline int shape_of(const mxg_csr* A, const mxg_mv* b, Shape* s) {
  int64_t nr = 0, nc = 0, ng = 0, nz = 0, nl = 0;
  MXK_TRY(mxg_csr_info(A, &nr, &nc, &ng, &nz, &s->cx));
  int cxv = 0;
  MXK_TRY(mxg_mv_info(b, &nl, &s->k, &cxv));
  s->nrows = nr;
  return 0;
}

template <class ApplyPrec, class Fused>
int solve_cg(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, mxg_mv* x, const mxg_solve_opts* o, mxg_solve_info* info,
             ApplyPrec apply_prec, Fused fused) {
  Shape sh;
  MXK_TRY(shape_of(A, b, &sh));
  const int cx = sh.cx;
  const int max_iters = o->max_iters > 0 ? o->max_iters : 100000;
  info->iters = 0; info->converged = 1; info->true_res = 0.0;
  Guard g;
  mxg_mv *r, *z, *p, *q;
  MXK_TRY(g.make(c, sh.nrows, 1, cx, &r));
  MXK_TRY(g.make(c, sh.nrows, 1, cx, &z));
  MXK_TRY(g.make(c, sh.nrows, 1, cx, &p));
  MXK_TRY(g.make(c, sh.nrows, 1, cx, &q));
  for (int col = 0; col < sh.k; ++col) {
    mxg_mv *bj, *xj;
    MXK_TRY(g.view(const_cast<mxg_mv*>(b), col, 1, &bj));
    MXK_TRY(g.view(x, col, 1, &xj));
    double res = 0.0;
    int it = 0;
    const int frc = fused(bj, xj, r, z, p, q, max_iters, &it, &res);
    if (frc > 0) return frc;
    if (frc == 0) {
      info->iters += it;
      if (res > info->true_res) info->true_res = res;
      if (res > o->tol) info->converged = 0;
      continue;
    }
    MXK_TRY(initial_residual(c, A, o, bj, xj, r, &res));
    // outer loop = the reference's `while (trueRes > linTol)` (src/MxMagWaveOp.cpp:876-885)
    while (res > o->tol && it < max_iters) {
      MXK_TRY(apply_prec(r, z));
      MXK_TRY(mxg_mv_copy(p, z));
      double rz[2];
      MXK_TRY(mxg_mv_dot(r, z, rz));
      double rnorm = res;
      while (rnorm > o->tol && it < max_iters) {
        MXK_TRY(mxg_csr_apply(c, A, p, q));
        double pq[2];
        MXK_TRY(mxg_mv_dot(p, q, pq));
        if (pq[0] == 0.0) break;
        const double alpha = rz[0] / pq[0];   // Hermitian A: both dots are real
        MXK_TRY(mxg_mv_axpby(alpha, 0.0, p, 1.0, 0.0, xj));
        MXK_TRY(mxg_mv_axpby(-alpha, 0.0, q, 1.0, 0.0, r));
        ++it;
        MXK_TRY(apply_prec(r, z));
        double rz_new[2];
        MXK_TRY(mxg_mv_dot(r, z, rz_new));
        if (o->precond == 0) rnorm = sqrt(fabs(rz_new[0]));
        else MXK_TRY(mxg_mv_norm2(r, &rnorm));
        const double beta = rz_new[0] / rz[0];
        rz[0] = rz_new[0];
        MXK_TRY(mxg_mv_axpby(1.0, 0.0, z, beta, 0.0, p));
      }
      MXK_TRY(true_residual(c, A, bj, xj, r, &res));
      if (rnorm > o->tol) break;  // stagnated or hit the cap: report what we have
    }
    info->iters += it;
    if (res > info->true_res) info->true_res = res;
    if (res > o->tol) info->converged = 0;
  }
  return 0;
}

// AZ_bicgstab (src/MxMagWaveOp.cpp:333-334): right-preconditioned BiCGStab, short recurrences for the
// non-symmetric operators (FIT inverse permittivity) - two products and two preconditioner applications per
// iteration, no basis to orthogonalise against.  Stops on the unscaled true residual like the other solvers.
template <class ApplyPrec>
int solve_bicgstab(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, mxg_mv* x, const mxg_solve_opts* o, mxg_solve_info* info,
                   ApplyPrec apply_prec) {
  Shape sh;
  MXK_TRY(shape_of(A, b, &sh));
  const int cx = sh.cx;
  const int max_iters = o->max_iters > 0 ? o->max_iters : 100000;
  info->iters = 0; info->converged = 1; info->true_res = 0.0;
  Guard g;
  mxg_mv *r, *rh, *p, *v, *y, *sv, *z, *t;
  for (mxg_mv** w : {&r, &rh, &p, &v, &y, &sv, &z, &t}) MXK_TRY(g.make(c, sh.nrows, 1, cx, w));
  auto cdot = [&](const mxg_mv* a, const mxg_mv* bb, cplx* out) -> int {   // a^H b
    double d[2] = {0.0, 0.0};
    MXK_TRY(mxg_mv_dot(a, bb, d));
    *out = cplx(d[0], cx ? d[1] : 0.0);
    return 0;
  };
  for (int col = 0; col < sh.k; ++col) {
    mxg_mv *bj, *xj;
    MXK_TRY(g.view(const_cast<mxg_mv*>(b), col, 1, &bj));
    MXK_TRY(g.view(x, col, 1, &xj));
    double res = 0.0;
    MXK_TRY(initial_residual(c, A, o, bj, xj, r, &res));
    int it = 0;
    while (res > o->tol && it < max_iters) {       // (re)start from the true residual
      MXK_TRY(mxg_mv_copy(rh, r));
      MXK_TRY(mxg_mv_fill(p, 0.0, 0.0));
      MXK_TRY(mxg_mv_fill(v, 0.0, 0.0));
      cplx rho = 1.0, alpha = 1.0, omega = 1.0;
      double rnorm = res;
      bool broke = false;
      while (rnorm > o->tol && it < max_iters) {
        cplx rho_new;
        MXK_TRY(cdot(rh, r, &rho_new));
        if (rho_new == cplx(0.0) || omega == cplx(0.0)) { broke = true; break; }
        const cplx beta = (rho_new / rho) * (alpha / omega);
        rho = rho_new;
        // p = r + beta (p - omega v)
        MXK_TRY(mxg_mv_axpby(-omega.real(), -omega.imag(), v, 1.0, 0.0, p));
        MXK_TRY(mxg_mv_axpby(1.0, 0.0, r, beta.real(), beta.imag(), p));
        MXK_TRY(apply_prec(p, y));
        MXK_TRY(mxg_csr_apply(c, A, y, v));
        cplx rhv;
        MXK_TRY(cdot(rh, v, &rhv));
        if (rhv == cplx(0.0)) { broke = true; break; }
        alpha = rho / rhv;
        MXK_TRY(mxg_mv_add(1.0, 0.0, r, -alpha.real(), -alpha.imag(), v, sv));      // s = r - alpha v
        MXK_TRY(apply_prec(sv, z));
        MXK_TRY(mxg_csr_apply(c, A, z, t));
        cplx ts, tt;
        MXK_TRY(cdot(t, sv, &ts));
        MXK_TRY(cdot(t, t, &tt));
        omega = tt == cplx(0.0) ? cplx(0.0) : ts / tt;
        MXK_TRY(mxg_mv_axpby(alpha.real(), alpha.imag(), y, 1.0, 0.0, xj));         // x += alpha y + omega z
        MXK_TRY(mxg_mv_axpby(omega.real(), omega.imag(), z, 1.0, 0.0, xj));
        MXK_TRY(mxg_mv_add(1.0, 0.0, sv, -omega.real(), -omega.imag(), t, r));      // r = s - omega t
        ++it;
        MXK_TRY(mxg_mv_norm2(r, &rnorm));
        if (!(rnorm == rnorm)) { broke = true; break; }
      }
      const double prev = res;
      MXK_TRY(true_residual(c, A, bj, xj, r, &res));
      if (!(res == res)) break;
      if (broke && !(res < prev)) break;          // breakdown without progress: report what we have
      if (!broke && rnorm > o->tol) break;        // iteration cap
      if (!(res < prev) && res > o->tol) break;   // recurrence and true residual disagree, no progress
    }
    info->iters += it;
    if (res > info->true_res) info->true_res = res;
    if (!(res <= o->tol)) info->converged = 0;
  }
  return 0;
}


// Restarted GMRES(m), right preconditioned; classical Gram-Schmidt with re-orthogonalisation expressed as two fused
// block products (mxg_mv_trans_mv / mxg_mv_times_mat): one iteration costs two reductions instead of j.
template <class PrecView>
int solve_gmres(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, mxg_mv* x, const mxg_solve_opts* o, mxg_solve_info* info,
                PrecView prec_view) {
  Shape sh;
  MXK_TRY(shape_of(A, b, &sh));
  const int cx = sh.cx;
  const int w = cx ? 2 : 1;
  const int m = o->restart > 0 ? o->restart : 40;
  const int max_iters = o->max_iters > 0 ? o->max_iters : 100000;
  info->iters = 0; info->converged = 1; info->true_res = 0.0;
  Guard g;
  mxg_mv *V, *r, *z, *wv;
  MXK_TRY(g.make(c, sh.nrows, m + 1, cx, &V));
  MXK_TRY(g.make(c, sh.nrows, 1, cx, &r));
  MXK_TRY(g.make(c, sh.nrows, 1, cx, &z));
  MXK_TRY(g.make(c, sh.nrows, 1, cx, &wv));
  std::vector<mxg_mv*> vcol(m + 1), vlead(m + 1);
  for (int j = 0; j <= m; ++j) {
    MXK_TRY(g.view(V, j, 1, &vcol[j]));
    MXK_TRY(g.view(V, 0, j + 1, &vlead[j]));   // columns 0..j
  }
  std::vector<cplx> H((size_t)(m + 1) * m), cs(m), sn(m), gvec(m + 1), yv(m);
  std::vector<double> hbuf((size_t)(m + 1) * w), h2buf((size_t)(m + 1) * w);
  auto Hat = [&](int i, int j) -> cplx& { return H[(size_t)i + (size_t)(m + 1) * j]; };

  for (int col = 0; col < sh.k; ++col) {
    mxg_mv *bj, *xj;
    MXK_TRY(g.view(const_cast<mxg_mv*>(b), col, 1, &bj));
    MXK_TRY(g.view(x, col, 1, &xj));
    double res = 0.0;
    MXK_TRY(initial_residual(c, A, o, bj, xj, r, &res));
    int it = 0;
    while (res > o->tol && it < max_iters) {
      // one restart cycle (right preconditioned: A M^-1 u = b, x = M^-1 u)
      const double beta = res;
      MXK_TRY(mxg_mv_copy(vcol[0], r));
      MXK_TRY(mxg_mv_scale(vcol[0], 1.0 / beta, 0.0));
      std::fill(gvec.begin(), gvec.end(), cplx(0.0));
      gvec[0] = beta;
      int j = 0;
      double est = beta;
      for (; j < m && it < max_iters; ++j) {
        mxg_mv* zc;
        MXK_TRY(prec_view(vcol[j], z, &zc));
        MXK_TRY(mxg_csr_apply(c, A, zc, wv));
        ++it;
        // CGS2: h = V^H w; w -= V h; h2 = V^H w; w -= V h2; h += h2
        MXK_TRY(mxg_mv_trans_mv(1.0, 0.0, vlead[j], wv, hbuf.data(), j + 1));
        MXK_TRY(mxg_mv_times_mat(-1.0, 0.0, vlead[j], hbuf.data(), j + 1, j + 1, 1, 1.0, 0.0, wv));
        MXK_TRY(mxg_mv_trans_mv(1.0, 0.0, vlead[j], wv, h2buf.data(), j + 1));
        MXK_TRY(mxg_mv_times_mat(-1.0, 0.0, vlead[j], h2buf.data(), j + 1, j + 1, 1, 1.0, 0.0, wv));
        for (int i = 0; i <= j; ++i)
          Hat(i, j) = cx ? cplx(hbuf[2 * i] + h2buf[2 * i], hbuf[2 * i + 1] + h2buf[2 * i + 1])
                         : cplx(hbuf[i] + h2buf[i], 0.0);
        double hn;
        MXK_TRY(mxg_mv_norm2(wv, &hn));
        Hat(j + 1, j) = hn;
        // previous rotations, then a new one to annihilate H(j+1, j)
        for (int i = 0; i < j; ++i) {
          cplx t = std::conj(cs[i]) * Hat(i, j) + std::conj(sn[i]) * Hat(i + 1, j);
          Hat(i + 1, j) = -sn[i] * Hat(i, j) + cs[i] * Hat(i + 1, j);
          Hat(i, j) = t;
        }
        {
          const cplx a = Hat(j, j), bb = Hat(j + 1, j);
          const double den = std::sqrt(std::norm(a) + std::norm(bb));
          if (den == 0.0) { cs[j] = 1.0; sn[j] = 0.0; }
          else { cs[j] = a / den; sn[j] = bb / den; }
          Hat(j, j) = std::conj(cs[j]) * a + std::conj(sn[j]) * bb;
          Hat(j + 1, j) = 0.0;
          gvec[j + 1] = -sn[j] * gvec[j];
          gvec[j] = std::conj(cs[j]) * gvec[j];
        }
        est = std::abs(gvec[j + 1]);
        if (hn != 0.0 && j + 1 <= m) {
          MXK_TRY(mxg_mv_copy(vcol[j + 1], wv));
          MXK_TRY(mxg_mv_scale(vcol[j + 1], 1.0 / hn, 0.0));
        }
        if (est <= o->tol || hn == 0.0) { ++j; break; }
      }
      // solve the j x j triangular system and update x += M^-1 (V y)
      const int jj = j;
      for (int i = jj - 1; i >= 0; --i) {
        cplx s = gvec[i];
        for (int l = i + 1; l < jj; ++l) s -= Hat(i, l) * yv[l];
        yv[i] = Hat(i, i) != cplx(0.0) ? s / Hat(i, i) : cplx(0.0);
      }
      if (jj > 0) {
        for (int i = 0; i < jj; ++i) {
          if (cx) { hbuf[2 * i] = yv[i].real(); hbuf[2 * i + 1] = yv[i].imag(); }
          else hbuf[i] = yv[i].real();
        }
        MXK_TRY(mxg_mv_times_mat(1.0, 0.0, vlead[jj - 1], hbuf.data(), jj, jj, 1, 0.0, 0.0, wv));
        mxg_mv* zc;
        MXK_TRY(prec_view(wv, z, &zc));
        MXK_TRY(mxg_mv_axpby(1.0, 0.0, zc, 1.0, 0.0, xj));
      }
      const double prev = res;
      MXK_TRY(true_residual(c, A, bj, xj, r, &res));
      if (jj == 0 || !(res < prev)) break;   // no progress: stop instead of spinning forever
    }
    info->iters += it;
    if (res > info->true_res) info->true_res = res;
    if (res > o->tol) info->converged = 0;
  }
  return 0;
}

}  // namespace mxg_krylov
#endif
