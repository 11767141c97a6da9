// Inner Krylov solves of MxMagWaveOp::Apply (K10/K11 of SURVEY section 2.2).
// Replaces AztecOO::recursiveIterate + the TrueResidual() loop (reference
// src/MxMagWaveOp.cpp:874-885,907-913; options :326-353: AZ_kspace = "linear solver : basis",
// AZ_conv = AZ_noscaled, i.e. the test is on the unscaled true residual 2-norm).
//
// Everything stays on the GPU; only scalars (dots, norms, the small Hessenberg) visit the
// host.  GMRES uses classical Gram-Schmidt with re-orthogonalisation expressed as two fused
// block products (mxg_mv_trans_mv / mxg_mv_times_mat), so one iteration costs two
// reductions instead of j.  The ILU / AMG preconditioners of the reference are Trilinos
// packages and are replaced by SpMV-only ones (Jacobi, Chebyshev polynomial): they change the
// iteration count, not the converged answer.
#include <cmath>
#include <complex>
#include <vector>

#include "mxg_internal.h"

namespace {

typedef std::complex<double> cplx;

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

__global__ void k_jacobi(const double* __restrict__ diag, int w, long long n,
                         const double* __restrict__ r, double* __restrict__ z) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (w == 1) {
      const double d = diag[i];
      z[i] = d != 0.0 ? r[i] / d : 0.0;
    } else {
      const double dr = diag[2 * i], di = diag[2 * i + 1];
      const double den = dr * dr + di * di;
      const double rr = r[2 * i], ri = r[2 * i + 1];
      z[2 * i] = den != 0.0 ? (rr * dr + ri * di) / den : 0.0;
      z[2 * i + 1] = den != 0.0 ? (ri * dr - rr * di) / den : 0.0;
    }
  }
}

// z = M^-1 r for one column
int apply_prec(mxg_ctx* c, const mxg_csr* A, const mxg_solve_opts* o, const mxg_mv* r, mxg_mv* z) {
  if (o->precond == 1) {
    const int w = A->is_complex ? 2 : 1;
    const int grid = mxg_grid_for(A->nrows, 256, 2, 8 * c->num_cu);
    hipLaunchKernelGGL(k_jacobi, dim3(grid), dim3(256), 0, c->stream, A->d_diag, w,
                       (long long)A->nrows, r->colptr(0), z->colptr(0));
    MXG_HIP(hipGetLastError());
    return 0;
  }
  return mxg_mv_copy(z, r);
}

int true_residual(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, const mxg_mv* x, mxg_mv* r,
                  double* nrm) {
  MXG_TRY(mxg_csr_apply(c, A, x, r));
  MXG_TRY(mxg_mv_axpby(1.0, 0.0, b, -1.0, 0.0, r));   // r = b - A x
  return mxg_mv_norm2(r, nrm);
}

}  // namespace

extern "C" int mxg_solve_cg(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, mxg_mv* x,
                            const mxg_solve_opts* o, mxg_solve_info* info) {
  MXG_CHECK(c && A && b && x && o && info, "mxg_solve_cg: NULL argument");
  MXG_CHECK(A->nrows == A->ncols_owned, "mxg_solve_cg: matrix must be square");
  MXG_CHECK(b->k == x->k && b->nlocal == A->nrows && x->nlocal == A->nrows,
            "mxg_solve_cg: shape mismatch");
  const int cx = A->is_complex;
  const int max_iters = o->max_iters > 0 ? o->max_iters : 100000;
  info->iters = 0; info->converged = 1; info->true_res = 0.0;
  Guard g;
  mxg_mv *r, *z, *p, *q;
  MXG_TRY(g.make(c, A->nrows, 1, cx, &r));
  MXG_TRY(g.make(c, A->nrows, 1, cx, &z));
  MXG_TRY(g.make(c, A->nrows, 1, cx, &p));
  MXG_TRY(g.make(c, A->nrows, 1, cx, &q));
  for (int col = 0; col < b->k; ++col) {
    mxg_mv *bj, *xj;
    MXG_TRY(g.view(const_cast<mxg_mv*>(b), col, 1, &bj));
    MXG_TRY(g.view(x, col, 1, &xj));
    double res = 0.0;
    MXG_TRY(true_residual(c, A, bj, xj, r, &res));
    int it = 0;
    // outer loop = the reference's `while (trueRes > linTol)` (src/MxMagWaveOp.cpp:876-885)
    while (res > o->tol && it < max_iters) {
      MXG_TRY(apply_prec(c, A, o, r, z));
      MXG_TRY(mxg_mv_copy(p, z));
      double rz[2];
      MXG_TRY(mxg_mv_dot(r, z, rz));
      double rnorm = res;
      while (rnorm > o->tol && it < max_iters) {
        MXG_TRY(mxg_csr_apply(c, A, p, q));
        double pq[2];
        MXG_TRY(mxg_mv_dot(p, q, pq));
        if (pq[0] == 0.0) break;
        const double alpha = rz[0] / pq[0];   // Hermitian A: both dots are real
        MXG_TRY(mxg_mv_axpby(alpha, 0.0, p, 1.0, 0.0, xj));
        MXG_TRY(mxg_mv_axpby(-alpha, 0.0, q, 1.0, 0.0, r));
        ++it;
        MXG_TRY(apply_prec(c, A, o, r, z));
        double rz_new[2];
        MXG_TRY(mxg_mv_dot(r, z, rz_new));
        if (o->precond == 0) rnorm = sqrt(fabs(rz_new[0]));
        else MXG_TRY(mxg_mv_norm2(r, &rnorm));
        const double beta = rz_new[0] / rz[0];
        rz[0] = rz_new[0];
        MXG_TRY(mxg_mv_axpby(1.0, 0.0, z, beta, 0.0, p));
      }
      MXG_TRY(true_residual(c, A, bj, xj, r, &res));
      if (rnorm > o->tol) break;  // stagnated or hit the cap: report what we have
    }
    info->iters += it;
    if (res > info->true_res) info->true_res = res;
    if (res > o->tol) info->converged = 0;
  }
  return 0;
}

extern "C" int mxg_solve_gmres(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, mxg_mv* x,
                               const mxg_solve_opts* o, mxg_solve_info* info) {
  MXG_CHECK(c && A && b && x && o && info, "mxg_solve_gmres: NULL argument");
  MXG_CHECK(A->nrows == A->ncols_owned, "mxg_solve_gmres: matrix must be square");
  MXG_CHECK(b->k == x->k && b->nlocal == A->nrows && x->nlocal == A->nrows,
            "mxg_solve_gmres: shape mismatch");
  const int cx = A->is_complex;
  const int w = cx ? 2 : 1;
  const int m = o->restart > 0 ? o->restart : 40;
  const int max_iters = o->max_iters > 0 ? o->max_iters : 100000;
  info->iters = 0; info->converged = 1; info->true_res = 0.0;
  Guard g;
  mxg_mv *V, *r, *z, *wv;
  MXG_TRY(g.make(c, A->nrows, m + 1, cx, &V));
  MXG_TRY(g.make(c, A->nrows, 1, cx, &r));
  MXG_TRY(g.make(c, A->nrows, 1, cx, &z));
  MXG_TRY(g.make(c, A->nrows, 1, cx, &wv));
  std::vector<mxg_mv*> vcol(m + 1), vlead(m + 1);
  for (int j = 0; j <= m; ++j) {
    MXG_TRY(g.view(V, j, 1, &vcol[j]));
    MXG_TRY(g.view(V, 0, j + 1, &vlead[j]));   // columns 0..j
  }
  std::vector<cplx> H((size_t)(m + 1) * m), cs(m), sn(m), gvec(m + 1), yv(m);
  std::vector<double> hbuf((size_t)(m + 1) * w), h2buf((size_t)(m + 1) * w);
  auto Hat = [&](int i, int j) -> cplx& { return H[(size_t)i + (size_t)(m + 1) * j]; };

  for (int col = 0; col < b->k; ++col) {
    mxg_mv *bj, *xj;
    MXG_TRY(g.view(const_cast<mxg_mv*>(b), col, 1, &bj));
    MXG_TRY(g.view(x, col, 1, &xj));
    double res = 0.0;
    MXG_TRY(true_residual(c, A, bj, xj, r, &res));
    int it = 0;
    while (res > o->tol && it < max_iters) {
      // one restart cycle (right preconditioned: A M^-1 u = b, x = M^-1 u)
      const double beta = res;
      MXG_TRY(mxg_mv_copy(vcol[0], r));
      MXG_TRY(mxg_mv_scale(vcol[0], 1.0 / beta, 0.0));
      std::fill(gvec.begin(), gvec.end(), cplx(0.0));
      gvec[0] = beta;
      int j = 0;
      double est = beta;
      for (; j < m && it < max_iters; ++j) {
        MXG_TRY(apply_prec(c, A, o, vcol[j], z));
        MXG_TRY(mxg_csr_apply(c, A, z, wv));
        ++it;
        // CGS2: h = V^H w; w -= V h; h2 = V^H w; w -= V h2; h += h2
        MXG_TRY(mxg_mv_trans_mv(1.0, 0.0, vlead[j], wv, hbuf.data(), j + 1));
        MXG_TRY(mxg_mv_times_mat(-1.0, 0.0, vlead[j], hbuf.data(), j + 1, j + 1, 1, 1.0, 0.0, wv));
        MXG_TRY(mxg_mv_trans_mv(1.0, 0.0, vlead[j], wv, h2buf.data(), j + 1));
        MXG_TRY(mxg_mv_times_mat(-1.0, 0.0, vlead[j], h2buf.data(), j + 1, j + 1, 1, 1.0, 0.0, wv));
        for (int i = 0; i <= j; ++i)
          Hat(i, j) = cx ? cplx(hbuf[2 * i] + h2buf[2 * i], hbuf[2 * i + 1] + h2buf[2 * i + 1])
                         : cplx(hbuf[i] + h2buf[i], 0.0);
        double hn;
        MXG_TRY(mxg_mv_norm2(wv, &hn));
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
          MXG_TRY(mxg_mv_copy(vcol[j + 1], wv));
          MXG_TRY(mxg_mv_scale(vcol[j + 1], 1.0 / hn, 0.0));
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
        MXG_TRY(mxg_mv_times_mat(1.0, 0.0, vlead[jj - 1], hbuf.data(), jj, jj, 1, 0.0, 0.0, wv));
        MXG_TRY(apply_prec(c, A, o, wv, z));
        MXG_TRY(mxg_mv_axpby(1.0, 0.0, z, 1.0, 0.0, xj));
      }
      const double prev = res;
      MXG_TRY(true_residual(c, A, bj, xj, r, &res));
      if (jj == 0 || !(res < prev)) break;   // no progress: stop instead of spinning forever
    }
    info->iters += it;
    if (res > info->true_res) info->true_res = res;
    if (res > o->tol) info->converged = 0;
  }
  return 0;
}
