// Inner Krylov solves of MxMagWaveOp::Apply (K10/K11 of SURVEY section 2.2).
// Replaces AztecOO::recursiveIterate + the TrueResidual() loop (reference
// src/MxMagWaveOp.cpp:874-885,907-913; options :326-353: AZ_kspace = "linear solver : basis",
// AZ_conv = AZ_noscaled, i.e. the test is on the unscaled true residual 2-norm).
//
// Everything stays on the GPU; only scalars (dots, norms, the small Hessenberg) visit the
// host.  The Krylov loops themselves (CG, BiCGStab, GMRES with CGS2 as two fused block products) are written on
// the public ABI in mxg_krylov.h; this file adds the preconditioners, the multigrid cycle and the fused CG.  The ILU / AMG preconditioners of the reference are Trilinos
// packages and are replaced by SpMV-only ones (Jacobi, Chebyshev polynomial): they change the
// iteration count, not the converged answer.
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

#include "mxg_internal.h"
#include "mxg_krylov.h"

namespace {

using mxg_krylov::true_residual;
using mxg_krylov::initial_residual;

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
  if (o->precond == 3) {
    MXG_CHECK(o->mg != nullptr, "precond == 3 needs opts->mg");
    return mxg_mg_apply(c, o->mg, r, z);
  }
  return mxg_mv_copy(z, r);
}


// ------------------------------------------------------------------------------------------
// Fused CG (real scalars, Jacobi or no preconditioner): three passes per iteration and no host
// round trip inside a chunk of iterations.
//   1. q = A p with p.q folded into the SpMV          (mxg_dev_csr_apply_dot)
//   2. r -= a q, z = D^-1 r, r.z and r.r               (k_cg_update; a = rz / pq read on the device)
//   3. x += a p, p = z + b p                           (k_cg_dir;   b = rz_new / rz read on the device)
// All scalars live in ctx->d_result; across ranks they are all-reduced in place on the stream.
// Every CHUNK iterations r.r comes to the host for the convergence test (unscaled 2-norm).
// ------------------------------------------------------------------------------------------
constexpr int CGB = 256;
enum { S_RZ0 = 0, S_RR = 1, S_RZ1 = 2, S_PQ = 3 };   // r.r sits between the two r.z slots: one all-reduce covers both

__global__ void k_cg_update(long long n, double* __restrict__ xu, const double* __restrict__ pu, double* __restrict__ r,
                            const double* __restrict__ q,
                            double* __restrict__ z, const double* __restrict__ diag,
                            const double* __restrict__ scal, int irz, double* __restrict__ partial,
                            double* __restrict__ zmg, const double* __restrict__ dmg, double omg) {
  __shared__ double lds[2][CGB / 64];
  const double pq = scal[S_PQ];
  const double alpha = pq != 0.0 ? scal[irz] / pq : 0.0;
  double srz = 0.0, srr = 0.0;
  const long long stride = (long long)gridDim.x * CGB;
  for (long long i = (long long)blockIdx.x * CGB + threadIdx.x; i < n; i += stride) {
    if (xu) xu[i] += alpha * pu[i];      // (x updated here instead of in k_cg_dir: MXG_CG_XMOVE=0)
    const double ri = r[i] - alpha * q[i];
    r[i] = ri;
    double zi = ri;
    if (diag) { const double d = diag[i]; zi = d != 0.0 ? ri / d : 0.0; }
    if (z) z[i] = zi;          // z == nullptr: the preconditioner (multigrid) forms z afterwards ...
    if (zmg) { const double d = dmg[i]; zmg[i] = d != 0.0 ? omg * ri / d : 0.0; }   // ... from this first smoother step (k_jac_first's arithmetic)
    srz += ri * zi;
    srr += ri * ri;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { srz += __shfl_down(srz, off, 64); srr += __shfl_down(srr, off, 64); }
  if ((threadIdx.x & 63) == 0) { lds[0][threadIdx.x >> 6] = srz; lds[1][threadIdx.x >> 6] = srr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < CGB / 64; ++w) { a += lds[0][w]; b += lds[1][w]; }
    partial[blockIdx.x] = a;
    partial[MXG_MAX_RED_BLOCKS + blockIdx.x] = b;
  }
}

__global__ void k_cg_reduce2(const double* __restrict__ partial, int nblocks, double* __restrict__ scal,
                             int irz_new) {
  __shared__ double lds[CGB];
  const double* p = partial + (size_t)blockIdx.x * MXG_MAX_RED_BLOCKS;
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += CGB) s += p[i];
  lds[threadIdx.x] = s;
  __syncthreads();
  for (int w = CGB / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) lds[threadIdx.x] += lds[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) scal[blockIdx.x == 0 ? irz_new : S_RR] = lds[0];
}

// x += a p (the step of this iteration: p is read here anyway, the update kernel does not touch it), p = z + b p
__global__ void k_cg_dir(long long n, double* __restrict__ p, const double* __restrict__ z, double* __restrict__ x,
                         const double* __restrict__ scal, int irz_old, int irz_new) {
  const double den = scal[irz_old], pq = scal[S_PQ];
  const double beta = den != 0.0 ? scal[irz_new] / den : 0.0;
  const double alpha = pq != 0.0 ? den / pq : 0.0;
  const long long stride = (long long)gridDim.x * CGB;
  for (long long i = (long long)blockIdx.x * CGB + threadIdx.x; i < n; i += stride) {
    const double po = p[i];
    if (x) x[i] += alpha * po;
    p[i] = z[i] + beta * po;
  }
}


// x += a p alone (a converged multigrid-CG solve leaves before the direction kernel)
__global__ void k_cg_xstep(long long n, const double* __restrict__ p, double* __restrict__ x,
                           const double* __restrict__ scal, int irz) {
  const double pq = scal[S_PQ];
  const double alpha = pq != 0.0 ? scal[irz] / pq : 0.0;
  const long long stride = (long long)gridDim.x * CGB;
  for (long long i = (long long)blockIdx.x * CGB + threadIdx.x; i < n; i += stride) x[i] += alpha * p[i];
}

// ------------------------------------------------------------------------------------------
// Aggregation multigrid V-cycle (K11 replacement, SpMV-only): every ingredient is a CSR product
// (level operator, prolongation, restriction) or a fused vector kernel.
// ------------------------------------------------------------------------------------------
}  // namespace

struct mxg_mg {
  mxg_ctx* ctx = nullptr;
  int nlevels = 0;
  std::vector<mxg_csr*> A, P, R;
  std::vector<mxg_mv*> r, z, t;       // per level: right-hand side, correction, scratch (t doubles as the
                                      // ping-pong partner of z for the fused sweeps)
  int pre = 2, post = 2, coarse = 20;
  double omega = 0.7, cscale = 1.9;
  // decomposed grid: replicated continuation below the last row-partitioned level (mxg_mg_set_tail)
  // Chebyshev smoother (the reference's ML default, `smoother = Chebyshev`): per level the largest eigenvalue of
  // D^-1 A (power iteration at setup); interval [lmax / ratio, boost * lmax]
  bool cheb = false;
  int deep_mult = 1;                  // smoother steps on levels >= 1 = deep_mult x the level-0 count
  double cheb_ratio = 20.0;
  int coarse_cheb_degree = 0;         // coarsest level: Chebyshev of this degree instead of `coarse` Jacobi sweeps (0 = Jacobi)
  double coarse_cheb_ratio = 100.0;
  std::vector<double> lmax;
  std::vector<mxg_mv*> u;             // product scratch of the unfused Chebyshev step
  double* dot_out = nullptr;          // multigrid-CG: where the cycle's last level-0 smoother step leaves z . r (device) ...
  bool dot_done = false;              // ... and whether it did (tile form only; the caller runs a dot kernel otherwise)
  bool first_done = false;            // multigrid-CG: the level-0 vector z already holds the first smoother step (k_cg_update wrote it)
  mxg_mg* tail = nullptr;
  std::vector<int64_t> tail_counts;
  int64_t tail_pad = 0, tail_off = 0;
  double* d_gsend = nullptr;
  double* d_grecv = nullptr;
};

namespace {

// z = omega * D^-1 r   (first sweep from a zero guess)
__global__ void k_jac_first(long long n, const double* __restrict__ diag, double omega,
                            const double* __restrict__ r, double* __restrict__ z) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double d = diag[i];
    z[i] = d != 0.0 ? omega * r[i] / d : 0.0;
  }
}
// z += omega * D^-1 (r - t), t = A z
__global__ void k_jac_step(long long n, const double* __restrict__ diag, double omega,
                           const double* __restrict__ r, const double* __restrict__ t, double* __restrict__ z) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double d = diag[i];
    if (d != 0.0) z[i] += omega * (r[i] - t[i]) / d;
  }
}
// t = r - t
__global__ void k_resid(long long n, const double* __restrict__ r, double* __restrict__ t) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) t[i] = r[i] - t[i];
}
// z += s * t
__global__ void k_axpy1(long long n, double s, const double* __restrict__ t, double* __restrict__ z) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) z[i] += s * t[i];
}

// t = z + beta (z - t) + omega D^-1 (r - u), u = A z   (unfused Chebyshev step; t holds the previous iterate)
__global__ void k_cheb_step(long long n, const double* __restrict__ diag, double omega, double beta,
                            const double* __restrict__ r, const double* __restrict__ u, const double* __restrict__ z,
                            double* __restrict__ t, int prev_zero) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double d = diag[i], zi = z[i], prev = prev_zero ? 0.0 : t[i];
    t[i] = d != 0.0 ? zi + beta * (zi - prev) + omega * (r[i] - u[i]) / d : zi;
  }
}

// The whole Chebyshev iteration of a SMALL level (the coarsest: ~1 000 rows of <= 16 entries) in ONE workgroup: every
// thread keeps its two rows of the matrix in registers, the iterates live in LDS, a step is a barrier instead of a launch
// (16 launches of ~7 us each were 7 % of the 256^3 eigen-solve; walking the CSR arrays from L2 inside the workgroup was
// no faster than the launches - the row walks are dependent loads).  Same arithmetic as the fused epilogues (row sums in
// storage order, the step formulas of EPI 5 / 4), from a zero guess: identical iterates.
constexpr int CHEB_SMALL_ROWS = 2048, CHEB_SMALL_STEPS = 32, CHEB_SMALL_LEN = 16;
struct ChebCoef { double omega[CHEB_SMALL_STEPS], beta[CHEB_SMALL_STEPS]; };
__global__ __launch_bounds__(1024) void k_cheb_small(const long long* __restrict__ rowptr, const int* __restrict__ colidx,
                                                      const double* __restrict__ vals, const double* __restrict__ diag, int n,
                                                      const double* __restrict__ r, double* __restrict__ zout, int sweeps,
                                                      const ChebCoef cf) {
  __shared__ double za[CHEB_SMALL_ROWS], zb[CHEB_SMALL_ROWS];
  double* z = za;
  double* t = zb;
  int cc[2][CHEB_SMALL_LEN], len[2];
  double vv[2][CHEB_SMALL_LEN], dg[2], rr[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = threadIdx.x + 1024 * k;
    len[k] = 0; dg[k] = 0.0; rr[k] = 0.0;
    long long q0 = 0;
    if (row < n) { q0 = rowptr[row]; len[k] = (int)(rowptr[row + 1] - q0); dg[k] = diag[row]; rr[k] = r[row]; }
#pragma unroll
    for (int i = 0; i < CHEB_SMALL_LEN; ++i) {
      cc[k][i] = i < len[k] ? colidx[q0 + i] : 0;
      vv[k][i] = i < len[k] ? vals[q0 + i] : 0.0;
    }
    if (row < n) z[row] = dg[k] != 0.0 ? cf.omega[0] * rr[k] / dg[k] : 0.0;
  }
  __syncthreads();
  for (int s = 1; s < sweeps; ++s) {
    const double omega = cf.omega[s], beta = cf.beta[s];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = threadIdx.x + 1024 * k;
      if (row < n) {
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < CHEB_SMALL_LEN; ++i) {
          const double s2 = sum + vv[k][i] * z[cc[k][i]];
          sum = i < len[k] ? s2 : sum;
        }
        const double d = dg[k], zi = z[row];
        const double prev = s == 1 ? 0.0 : t[row];                  // the iterate before z (zero after the first step)
        t[row] = d != 0.0 ? (s == 1 ? zi + beta * zi + omega * (rr[k] - sum) / d : zi + beta * (zi - prev) + omega * (rr[k] - sum) / d) : zi;
      }
    }
    __syncthreads();
    double* w = z; z = t; t = w;
  }
  for (int i = threadIdx.x; i < n; i += 1024) zout[i] = z[i];
}

// longest row of a small matrix (cached in the handle; the row pointers of <= CHEB_SMALL_ROWS rows are 16 KiB)
int small_max_row_len(mxg_ctx* c, mxg_csr* A, int* out) {
  if (A->small_maxlen < 0) {
    std::vector<long long> rp((size_t)A->nrows + 1);
    MXG_HIP(hipMemcpyAsync(rp.data(), A->d_rowptr, sizeof(long long) * (A->nrows + 1), hipMemcpyDeviceToHost, c->stream));
    MXG_HIP(hipStreamSynchronize(c->stream));
    long long m = 0;
    for (long long r2 = 0; r2 < A->nrows; ++r2) m = std::max(m, rp[r2 + 1] - rp[r2]);
    A->small_maxlen = (int)std::min<long long>(m, 1 << 30);
  }
  *out = A->small_maxlen;
  return 0;
}

// `sweeps` steps of the Chebyshev iteration for D^-1 A on [lmax / ratio, lmax] (Saad, Iterative Methods, Alg. 12.1)
int mg_smooth_cheb(mxg_mg* mg, int l, int sweeps, bool zero_guess, double ratio, bool want_dot = false) {
  mxg_ctx* c = mg->ctx;
  mxg_csr* A = mg->A[l];
  const long long n = A->nrows;
  const int grid = mxg_grid_for(n, 256, 2, 8 * c->num_cu);
  const double lmax = mg->lmax[l], lmin = lmax / ratio;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma1 = theta / delta;
  double rho = 1.0 / sigma1;
  const bool skip_first = l == 0 && zero_guess && mg->first_done;
  if (l == 0) mg->first_done = false;
  static const bool small_kernel = !(getenv("MXG_MG_SMALL_KERNEL") && atoi(getenv("MXG_MG_SMALL_KERNEL")) == 0);
  int maxlen = CHEB_SMALL_LEN + 1;
  if (small_kernel && zero_guess && sweeps >= 2 && sweeps <= CHEB_SMALL_STEPS && n <= CHEB_SMALL_ROWS && n > 0 && !A->is_complex &&
      A->nghost == 0 && A->nneigh == 0)
    MXG_TRY(small_max_row_len(c, A, &maxlen));
  if (maxlen <= CHEB_SMALL_LEN) {
    ChebCoef cf;
    double rh = rho;
    for (int s = 0; s < sweeps; ++s) {
      if (s == 0) { cf.omega[0] = 1.0 / theta; cf.beta[0] = 0.0; }
      else { const double rn = 1.0 / (2.0 * sigma1 - rh); cf.omega[s] = 2.0 * rn / delta; cf.beta[s] = rn * rh; rh = rn; }
    }
    hipLaunchKernelGGL(k_cheb_small, dim3(1), dim3(1024), 0, c->stream, (const long long*)A->d_rowptr, (const int*)A->d_colidx,
                       (const double*)A->d_vals, (const double*)A->d_diag, (int)n, (const double*)mg->r[l]->colptr(0),
                       mg->z[l]->colptr(0), sweeps, cf);
    MXG_HIP(hipGetLastError());
    return 0;
  }
  bool prev_zero = false;             // the previous iterate is the zero guess: nothing to read
  for (int s = 0; s < sweeps; ++s) {
    double omega, beta;
    if (s == 0) { omega = 1.0 / theta; beta = 0.0; }
    else { const double rn = 1.0 / (2.0 * sigma1 - rho); omega = 2.0 * rn / delta; beta = rn * rho; rho = rn; }
    if (s == 0 && zero_guess && skip_first) { prev_zero = true; continue; }
    if (s == 0 && zero_guess) {
      hipLaunchKernelGGL(k_jac_first, dim3(grid), dim3(256), 0, c->stream, n, (const double*)A->d_diag, omega,
                         (const double*)mg->r[l]->colptr(0), mg->z[l]->colptr(0));
      prev_zero = true;
      continue;
    }
    // the previous iterate sits in t and is overwritten by the new one
    const int epi = s == 0 ? 1 : (prev_zero ? 5 : 4);
    int rc = 1;
    if (want_dot && s == sweeps - 1 && mg->dot_out && epi != 5) {      // last step of the cycle: z . r rides along
      rc = mxg_dev_csr_apply_epi_dot(c, A, mg->z[l], mg->t[l], epi, mg->r[l]->colptr(0), A->d_diag, omega, beta, mg->dot_out);
      if (rc == 0) mg->dot_done = true;
    }
    if (rc == 1) rc = mxg_dev_csr_apply_epi(c, A, mg->z[l], mg->t[l], epi, mg->r[l]->colptr(0), A->d_diag, omega, beta);
    if (rc > 1) return 1;
    if (rc == 1) {
      MXG_TRY(mxg_csr_apply(c, A, mg->z[l], mg->u[l]));
      hipLaunchKernelGGL(k_cheb_step, dim3(grid), dim3(256), 0, c->stream, n, (const double*)A->d_diag, omega,
                         s == 0 ? 0.0 : beta, (const double*)mg->r[l]->colptr(0), (const double*)mg->u[l]->colptr(0),
                         (const double*)mg->z[l]->colptr(0), mg->t[l]->colptr(0), (s == 0 || prev_zero) ? 1 : 0);
    }
    prev_zero = false;
    std::swap(mg->z[l], mg->t[l]);
  }
  MXG_HIP(hipGetLastError());
  return 0;
}

int mg_smooth(mxg_mg* mg, int l, int sweeps, bool zero_guess, bool want_dot = false) {
  mxg_ctx* c = mg->ctx;
  mxg_csr* A = mg->A[l];
  const long long n = A->nrows;
  const int grid = mxg_grid_for(n, 256, 2, 8 * c->num_cu);
  if (mg->cheb && l < mg->nlevels - 1) return mg_smooth_cheb(mg, l, sweeps, zero_guess, mg->cheb_ratio, want_dot);
  if (mg->cheb && mg->coarse_cheb_degree > 0) return mg_smooth_cheb(mg, l, mg->coarse_cheb_degree, zero_guess, mg->coarse_cheb_ratio);
  for (int s = 0; s < sweeps; ++s) {
    if (s == 0 && zero_guess) {
      hipLaunchKernelGGL(k_jac_first, dim3(grid), dim3(256), 0, c->stream, n, (const double*)A->d_diag,
                         mg->omega, (const double*)mg->r[l]->colptr(0), mg->z[l]->colptr(0));
      continue;
    }
    // fused sweep on a compressed level: t = z + omega D^-1 (r - A z) in the product's epilogue, then swap
    const int rc = mxg_dev_csr_apply_epi(c, A, mg->z[l], mg->t[l], 1, mg->r[l]->colptr(0), A->d_diag, mg->omega);
    if (rc == 0) { std::swap(mg->z[l], mg->t[l]); continue; }
    if (rc != 1) return 1;
    MXG_TRY(mxg_csr_apply(c, A, mg->z[l], mg->t[l]));
    hipLaunchKernelGGL(k_jac_step, dim3(grid), dim3(256), 0, c->stream, n, (const double*)A->d_diag,
                       mg->omega, (const double*)mg->r[l]->colptr(0), (const double*)mg->t[l]->colptr(0),
                       mg->z[l]->colptr(0));
  }
  MXG_HIP(hipGetLastError());
  return 0;
}

int mg_cycle(mxg_mg* mg, int l);

// last row-partitioned level: gather its right-hand side, run the replicated levels, keep the own segment
int mg_tail(mxg_mg* mg, int l) {
  mxg_ctx* c = mg->ctx;
  mxg_mg* t = mg->tail;
  const int64_t nown = mg->A[l]->nrows;
  if (nown > 0)
    MXG_HIP(hipMemcpyAsync(mg->d_gsend, mg->r[l]->colptr(0), sizeof(double) * nown, hipMemcpyDeviceToDevice, c->stream));
  MXG_TRY(mxg_comm_allgather_dev(c, mg->d_gsend, (size_t)mg->tail_pad, mg->d_grecv, c->stream));
  int64_t off = 0;
  for (size_t r = 0; r < mg->tail_counts.size(); ++r) {
    if (mg->tail_counts[r] > 0)
      MXG_HIP(hipMemcpyAsync(t->r[0]->colptr(0) + off, mg->d_grecv + mg->tail_pad * (int64_t)r,
                             sizeof(double) * mg->tail_counts[r], hipMemcpyDeviceToDevice, c->stream));
    off += mg->tail_counts[r];
  }
  MXG_TRY(mg_cycle(t, 0));
  if (nown > 0)
    MXG_HIP(hipMemcpyAsync(mg->z[l]->colptr(0), t->z[0]->colptr(0) + mg->tail_off, sizeof(double) * nown,
                           hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int mg_cycle(mxg_mg* mg, int l) {
  mxg_ctx* c = mg->ctx;
  if (l == mg->nlevels - 1) return mg->tail ? mg_tail(mg, l) : mg_smooth(mg, l, mg->coarse, true);
  const long long n = mg->A[l]->nrows;
  const int grid = mxg_grid_for(n, 256, 2, 8 * c->num_cu);
  const int sw = l == 0 ? mg->pre : mg->pre * mg->deep_mult;     // variable V-cycle: more steps on the cheap deeper levels
  MXG_TRY(mg_smooth(mg, l, sw, true));
  const int rc = mxg_dev_csr_apply_epi(c, mg->A[l], mg->z[l], mg->t[l], 2, mg->r[l]->colptr(0), nullptr, 0.0);   // t = r - A z
  if (rc > 1) return 1;
  if (rc == 1) {
    MXG_TRY(mxg_csr_apply(c, mg->A[l], mg->z[l], mg->t[l]));                     // t = A z
    hipLaunchKernelGGL(k_resid, dim3(grid), dim3(256), 0, c->stream, n, (const double*)mg->r[l]->colptr(0),
                       mg->t[l]->colptr(0));                                      // t = r - A z
  }
  {                                                                              // restrict (ELL form of R when it has one)
    const int rr = mxg_dev_csr_apply_ell(c, mg->R[l], mg->t[l], mg->r[l + 1]);
    if (rr > 1) return 1;
    if (rr == 1) MXG_TRY(mxg_csr_apply(c, mg->R[l], mg->t[l], mg->r[l + 1]));
  }
  MXG_TRY(mg_cycle(mg, l + 1));
  // z += cscale * P z_c: in the prolongation's epilogue when P is compressed (in place: a row reads only its own z)
  const int rp = mxg_dev_csr_apply_epi(c, mg->P[l], mg->z[l + 1], mg->z[l], 3, nullptr, nullptr, mg->cscale);
  if (rp > 1) return 1;
  if (rp == 1) {
    MXG_TRY(mxg_csr_apply(c, mg->P[l], mg->z[l + 1], mg->t[l]));                  // prolong
    hipLaunchKernelGGL(k_axpy1, dim3(grid), dim3(256), 0, c->stream, n, mg->cscale,
                       (const double*)mg->t[l]->colptr(0), mg->z[l]->colptr(0));
  }
  MXG_HIP(hipGetLastError());
  return mg_smooth(mg, l, sw, false, l == 0);
}

}  // namespace

extern "C" int mxg_mg_create(mxg_ctx* c, int nlevels, mxg_csr* const* A, mxg_csr* const* P,
                             mxg_csr* const* R, int pre, int post, int coarse, double omega,
                             double cscale, mxg_mg** out) {
  for (int l = 0; l < nlevels; ++l) {
    if (A && A[l]) MXG_TRY(mxg_csr_materialize(A[l]));
    if (l + 1 < nlevels && P && P[l]) MXG_TRY(mxg_csr_materialize(P[l]));
    if (l + 1 < nlevels && R && R[l]) MXG_TRY(mxg_csr_materialize(R[l]));
  }
  MXG_CHECK(c && A && out && nlevels >= 1, "mxg_mg_create: bad argument");
  MXG_CHECK(pre == post, "mxg_mg_create: pre and post sweeps must be equal (symmetric cycle for CG)");
  mxg_mg* mg = new mxg_mg();
  mg->ctx = c; mg->nlevels = nlevels; mg->pre = pre; mg->post = post; mg->coarse = coarse;
  mg->omega = omega; mg->cscale = cscale;
  for (int l = 0; l < nlevels; ++l) {
    // levels of a decomposed grid carry ghost columns: their products import them (mxg_csr_apply), the transfer
    // operators stay rank-local because aggregates never straddle two MxGridDomain boxes
    MXG_CHECK(A[l] && !A[l]->is_complex && A[l]->nrows == A[l]->ncols_owned,
              "mxg_mg_create: level %d must be a real matrix, square on the owned columns", l);
    mg->A.push_back(A[l]);
    if (l + 1 < nlevels) {
      MXG_CHECK(P && R && P[l] && R[l], "mxg_mg_create: missing transfer operator at level %d", l);
      MXG_CHECK(P[l]->nrows == A[l]->nrows && P[l]->ncols_owned == A[l + 1]->nrows &&
                R[l]->nrows == A[l + 1]->nrows && R[l]->ncols_owned == A[l]->nrows,
                "mxg_mg_create: transfer operator shapes do not chain at level %d", l);
      mg->P.push_back(P[l]); mg->R.push_back(R[l]);
    }
    mxg_mv *r, *z, *t;
    MXG_TRY(mxg_mv_create(c, A[l]->nrows, 1, 0, &r));
    MXG_TRY(mxg_mv_create(c, A[l]->nrows, 1, 0, &z));
    MXG_TRY(mxg_mv_create(c, A[l]->nrows, 1, 0, &t));
    mg->r.push_back(r); mg->z.push_back(z); mg->t.push_back(t);
  }
  if (const char* e = getenv("MXG_MG_DEEP_MULT")) mg->deep_mult = std::max(1, atoi(e));   // experiments
  if (const char* e = getenv("MXG_MG_CSCALE")) mg->cscale = atof(e);       // experiments (scripts/mg_smoother_sweep.py)

  *out = mg;
  return 0;
}

// Chebyshev smoothing on every level but the coarsest: lmax(D^-1 A) by power iteration (all-reduced norms on a
// decomposed grid), interval [boost * lmax / ratio, boost * lmax]
extern "C" int mxg_mg_set_chebyshev(mxg_mg* mg, double ratio) {
  MXG_CHECK(mg && ratio > 1.0, "mxg_mg_set_chebyshev: bad argument");
  mxg_ctx* c = mg->ctx;
  double boost = 1.1;
  if (const char* e = getenv("MXG_MG_CHEB_BOOST")) boost = atof(e);
  if (const char* e = getenv("MXG_MG_CHEB_RATIO")) ratio = atof(e);        // experiments (scripts/mg_smoother_sweep.py)
  mg->cheb_ratio = ratio;
  mg->lmax.assign(mg->nlevels, 2.0);
  for (int l = 0; l < mg->nlevels; ++l) {
    if ((int)mg->u.size() <= l) {
      mxg_mv* u;
      MXG_TRY(mxg_mv_create(c, mg->A[l]->nrows, 1, 0, &u));
      mg->u.push_back(u);
    }
    mxg_csr* A = mg->A[l];
    const long long n = A->nrows;
    const int grid = mxg_grid_for(n, 256, 2, 8 * c->num_cu);
    mxg_mv *v = mg->z[l], *w = mg->t[l];
    MXG_TRY(mxg_mv_random(v, 777 + l));
    double lam = 1.0, nv = 0.0, nw = 0.0;
    for (int it = 0; it < 15; ++it) {
      MXG_TRY(mxg_mv_norm2(v, &nv));
      if (!(nv > 0.0)) break;
      MXG_TRY(mxg_mv_scale(v, 1.0 / nv, 0.0));
      MXG_TRY(mxg_csr_apply(c, A, v, mg->u[l]));
      hipLaunchKernelGGL(k_jac_first, dim3(grid), dim3(256), 0, c->stream, n, (const double*)A->d_diag, 1.0,
                         (const double*)mg->u[l]->colptr(0), w->colptr(0));      // w = D^-1 A v (0 on inactive rows)
      MXG_HIP(hipGetLastError());
      MXG_TRY(mxg_mv_norm2(w, &nw));
      lam = nw;
      std::swap(v, w);
    }
    mg->lmax[l] = boost * lam;
  }
  // coarsest level ("coarse smoother" = Chebyshev, src/MxProblem.cpp:58): degree 16 on [lmax / 50, lmax] solves the
  // <= 2000-row problem as well as any longer iteration and takes half the launches of 30 Jacobi sweeps
  mg->coarse_cheb_degree = 16;
  mg->coarse_cheb_ratio = 50.0;
  if (const char* e = getenv("MXG_MG_COARSE_CHEB")) {                       // experiments: "degree:ratio", 0 = Jacobi sweeps
    mg->coarse_cheb_degree = atoi(e);
    if (const char* q = strchr(e, ':')) mg->coarse_cheb_ratio = atof(q + 1);
  }
  mg->cheb = true;
  return 0;
}

extern "C" int mxg_mg_set_tail(mxg_mg* mg, mxg_mg* tail, const int64_t* counts) {
  MXG_CHECK(mg && tail && counts, "mxg_mg_set_tail: NULL argument");
  mxg_ctx* c = mg->ctx;
  int64_t total = 0, mx = 0;
  mg->tail_counts.assign(counts, counts + c->nranks);
  for (int r = 0; r < c->nranks; ++r) {
    if (r == c->rank) mg->tail_off = total;
    total += counts[r];
    mx = std::max(mx, counts[r]);
  }
  MXG_CHECK(counts[c->rank] == mg->A.back()->nrows && total == tail->A[0]->nrows,
            "mxg_mg_set_tail: the replicated level has %lld rows, the ranks hold %lld (this rank %lld of %lld)",
            (long long)tail->A[0]->nrows, (long long)total, (long long)counts[c->rank], (long long)mg->A.back()->nrows);
  mg->tail_pad = std::max<int64_t>((mx + 31) & ~(int64_t)31, 32);
  if (mg->d_gsend) hipFree(mg->d_gsend);
  if (mg->d_grecv) hipFree(mg->d_grecv);
  MXG_HIP(hipMalloc(&mg->d_gsend, sizeof(double) * mg->tail_pad));
  MXG_HIP(hipMalloc(&mg->d_grecv, sizeof(double) * mg->tail_pad * c->nranks));
  MXG_HIP(hipMemsetAsync(mg->d_gsend, 0, sizeof(double) * mg->tail_pad, c->stream));
  mg->tail = tail;
  return 0;
}

extern "C" int mxg_mg_destroy(mxg_mg* mg) {
  if (!mg) return 0;
  if (mg->d_gsend) hipFree(mg->d_gsend);
  if (mg->d_grecv) hipFree(mg->d_grecv);
  for (auto* v : mg->r) mxg_mv_destroy(v);
  for (auto* v : mg->z) mxg_mv_destroy(v);
  for (auto* v : mg->t) mxg_mv_destroy(v);
  for (auto* v : mg->u) mxg_mv_destroy(v);
  delete mg;
  return 0;
}

namespace {
// part 0 / 1 of an interleaved complex vector -> real vector, and back
__global__ void k_cplx_take(long long n, const double* __restrict__ zc, int part, double* __restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = zc[2 * i + part];
}
__global__ void k_cplx_put(long long n, const double* __restrict__ in, int part, double* __restrict__ zc) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) zc[2 * i + part] = in[i];
}
}  // namespace

extern "C" int mxg_mg_apply(mxg_ctx* c, mxg_mg* mg, const mxg_mv* r, mxg_mv* z) {
  MXG_CHECK(c && mg && r && z, "mxg_mg_apply: NULL argument");
  MXG_CHECK(r->k == 1 && z->k == 1 && r->nlocal == mg->A[0]->nrows && z->nlocal == r->nlocal &&
            r->is_complex == z->is_complex, "mxg_mg_apply: shape mismatch");
  if (r->is_complex) {
    // real hierarchy (of the real part of a complex operator) on a complex residual: one cycle per part
    const long long n = r->nlocal;
    const int grid = mxg_grid_for(n, 256, 2, 8 * c->num_cu);
    for (int part = 0; part < 2; ++part) {
      hipLaunchKernelGGL(k_cplx_take, dim3(grid), dim3(256), 0, c->stream, n, (const double*)r->colptr(0), part,
                         mg->r[0]->colptr(0));
      MXG_TRY(mg_cycle(mg, 0));
      hipLaunchKernelGGL(k_cplx_put, dim3(grid), dim3(256), 0, c->stream, n, (const double*)mg->z[0]->colptr(0), part,
                         z->colptr(0));
    }
    MXG_HIP(hipGetLastError());
    return 0;
  }
  MXG_TRY(mxg_mv_copy(mg->r[0], r));
  MXG_TRY(mg_cycle(mg, 0));
  return mxg_mv_copy(z, mg->z[0]);
}

namespace {

// One V-cycle with the caller's r as the level-0 right-hand side; the result stays in the hierarchy's own level-0
// vector (*z_out, valid until the next cycle): the multigrid-CG loop saves two vector copies per iteration.
// d_rz (optional): r . z (owned rows, not all-reduced) goes there - from the cycle's last smoother step when the level-0
// operator has the tile form, from a dot kernel otherwise.
int mg_apply_nocopy(mxg_ctx* c, mxg_mg* mg, mxg_mv* r, mxg_mv** z_out, double* d_rz = nullptr) {
  MXG_CHECK(r->k == 1 && !r->is_complex && r->nlocal == mg->A[0]->nrows, "multigrid: shape mismatch");
  mxg_mv* own = mg->r[0];
  mg->r[0] = r;
  static const bool fuse = !(getenv("MXG_MG_FUSED_DOT") && atoi(getenv("MXG_MG_FUSED_DOT")) == 0);   // experiments
  mg->dot_out = fuse ? d_rz : nullptr; mg->dot_done = false;
  const int rc = mg_cycle(mg, 0);
  mg->r[0] = own;
  mg->dot_out = nullptr;
  *z_out = mg->z[0];
  if (rc == 0 && d_rz && !mg->dot_done) return mxg_dev_dot(c, r->desc(), mg->z[0]->desc(), d_rz);
  return rc;
}

// The first smoother step of the next cycle, for the caller to fold into its own pass over the residual: z_0 = omega D^-1 r
// into *z (the hierarchy's level-0 vector).  Returns false when the cycle does not start that way; after writing z the
// caller sets mg->first_done.
bool mg_first_step(mxg_mg* mg, double** z, const double** diag, double* omega) {
  static const bool fuse = !(getenv("MXG_MG_FUSED_FIRST") && atoi(getenv("MXG_MG_FUSED_FIRST")) == 0);   // experiments
  if (!fuse || !mg->cheb || mg->nlevels < 2 || mg->pre < 1 || mg->A[0]->nrows <= CHEB_SMALL_ROWS || !mg->A[0]->d_diag) return false;
  const double lmax = mg->lmax[0], lmin = lmax / mg->cheb_ratio;
  *omega = 1.0 / (0.5 * (lmax + lmin));
  *z = mg->z[0]->colptr(0);
  *diag = (const double*)mg->A[0]->d_diag;
  return true;
}

// z = M^-1 v where the caller only reads the result before the next application: with the multigrid (real
// scalars) *zc points at the hierarchy's own vector, otherwise at z
int prec_view(mxg_ctx* c, const mxg_csr* A, const mxg_solve_opts* o, mxg_mv* v, mxg_mv* z, mxg_mv** zc) {
  if (o->precond == 3 && o->mg && !v->is_complex) return mg_apply_nocopy(c, o->mg, v, zc);
  *zc = z;
  return apply_prec(c, A, o, v, z);
}

bool cg_fused_ok(const mxg_csr* A, const mxg_solve_opts* o) {
  // (must not depend on how many rows THIS rank owns: every rank of a decomposed grid has to take the same path)
  return !A->is_complex && A->pipe_geo == 10 &&
         (o->precond <= 1 || (o->precond == 3 && o->mg != nullptr));
}

// one right-hand side; x: initial guess in, solution out.  r, z, p, q: work vectors.
int cg_fused(mxg_ctx* c, mxg_csr* A, const mxg_solve_opts* o, const mxg_mv* b, mxg_mv* x, mxg_mv* r,
             mxg_mv* z, mxg_mv* p, mxg_mv* q, int max_iters, int* iters, double* res_out) {
  // multigrid-preconditioned CG converges in a few dozen 2-3 ms iterations: test every iteration;
  // Jacobi / plain CG runs thousands of sub-ms iterations: test every 16
  const bool mgprec = o->precond == 3;
  const int CHUNK = mgprec ? 1 : 16;
  const long long n = A->nrows;
  const int grid = mxg_grid_for(n, CGB, 4, 1024);
  double* scal = c->d_result;
  const double* diag = o->precond == 1 ? A->d_diag : nullptr;
  auto allreduce = [&](double* ptr, int count) -> int {
    return mxg_comm_on(c) ? mxg_comm_allreduce(c, ptr, (size_t)count, c->stream) : 0;
  };
  double res = 0.0;
  MXG_TRY(initial_residual(c, A, o, b, x, r, &res));
  int it = 0;
  mxg_mv* zc = z;                        // where the preconditioned residual sits (multigrid: its own level-0 vector)
  // z = V-cycle(r) (in the hierarchy's own vector: -4.6 % at 256^3 against two copies) and r.z into slot `slot`
  auto mgprecond = [&](int slot) -> int { return mg_apply_nocopy(c, o->mg, r, &zc, scal + slot); };
  static const bool xmove = !(getenv("MXG_CG_XMOVE") && atoi(getenv("MXG_CG_XMOVE")) == 0);   // experiments: x += a p in k_cg_dir (1) or k_cg_update (0)
  static const double near_factor = getenv("MXG_CG_NEAR") ? atof(getenv("MXG_CG_NEAR")) : 10.0;   // experiments: 0 = never test early
  while (res > o->tol && it < max_iters) {
    if (mgprec) MXG_TRY(mgprecond(S_RZ0));
    else MXG_TRY(apply_prec(c, A, o, r, z));
    MXG_TRY(mxg_mv_copy(p, zc));
    if (!mgprec) MXG_TRY(mxg_dev_dot(c, r->desc(), zc->desc(), scal + S_RZ0));
    MXG_TRY(allreduce(scal + S_RZ0, 1));
    int cur = S_RZ0;
    double rnorm = res;
    bool stalled = false, synced = false;
    while (rnorm > o->tol && it < max_iters) {
      const int todo = std::min(CHUNK, max_iters - it);
      for (int s = 0; s < todo; ++s) {
        const int nxt = cur == S_RZ0 ? S_RZ1 : S_RZ0;
        MXG_TRY(mxg_dev_csr_apply_dot(c, A, p, q, scal + S_PQ));
        MXG_TRY(allreduce(scal + S_PQ, 1));
        double* zmg = nullptr; const double* dmg = nullptr; double omg = 0.0;
        const bool first = mgprec && mg_first_step(o->mg, &zmg, &dmg, &omg);
        hipLaunchKernelGGL(k_cg_update, dim3(grid), dim3(CGB), 0, c->stream, n, xmove ? (double*)nullptr : x->colptr(0),
                           (const double*)p->colptr(0), r->colptr(0), (const double*)q->colptr(0),
                           mgprec ? (double*)nullptr : z->colptr(0), diag, (const double*)scal, cur, c->d_partial,
                           first ? zmg : (double*)nullptr, dmg, omg);
        if (first) o->mg->first_done = true;
        hipLaunchKernelGGL(k_cg_reduce2, dim3(2), dim3(CGB), 0, c->stream, (const double*)c->d_partial, grid,
                           scal, nxt);
        if (mgprec) {
          // r.r and p.q are final here: send them to the host now and enqueue the V-cycle behind the copy, so that the
          // convergence test below waits for an event the GPU passed a cycle ago and the next iteration is enqueued
          // while this cycle still runs (a stream synchronisation per iteration left the GPU idle ~90 us each)
          if (mxg_comm_on(c)) MXG_TRY(allreduce(scal + S_RR, 1));
          MXG_HIP(hipMemcpyAsync(c->h_result, scal, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
          // ... except close to the tolerance (the last one or two iterations of a solve): there the test is made
          // at once, and a converged solve does not run the cycle, the dot and the direction it would throw away
          // (1.2 ms of a 256^3 vector solve; the synchronisation costs 0.09)
          synced = rnorm <= o->tol * near_factor;
          if (synced) {
            MXG_HIP(hipStreamSynchronize(c->stream));
            const double rr = c->h_result[S_RR];
            if (rr == rr && c->h_result[S_PQ] != 0.0 && sqrt(rr) <= o->tol) {
              if (xmove)
                hipLaunchKernelGGL(k_cg_xstep, dim3(grid), dim3(CGB), 0, c->stream, n, (const double*)p->colptr(0), x->colptr(0),
                                   (const double*)scal, cur);   // x += a p: the part of k_cg_dir that is still owed
              if (first) o->mg->first_done = false;             // the cycle that would have consumed it is not run
              break;
            }
          } else {
            MXG_HIP(hipEventRecord(c->ev_scal, c->stream));
          }
          // z = V-cycle(r), r.z replaces the placeholder the update kernel left in slot nxt
          MXG_TRY(mgprecond(nxt));
          if (mxg_comm_on(c)) MXG_TRY(allreduce(scal + nxt, 1));
        } else if (mxg_comm_on(c)) {
          MXG_TRY(allreduce(scal + std::min(nxt, (int)S_RR), 2));   // r.z (slot nxt) and r.r, adjacent
        }
        hipLaunchKernelGGL(k_cg_dir, dim3(grid), dim3(CGB), 0, c->stream, n, p->colptr(0),
                           (const double*)zc->colptr(0), xmove ? x->colptr(0) : (double*)nullptr, (const double*)scal, cur, nxt);
        cur = nxt;
      }
      MXG_HIP(hipGetLastError());
      it += todo;
      if (mgprec) {
        if (!synced) MXG_HIP(hipEventSynchronize(c->ev_scal));
      } else {
        MXG_HIP(hipMemcpyAsync(c->h_result, scal, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
        MXG_HIP(hipStreamSynchronize(c->stream));
      }
      const double rr = c->h_result[S_RR];
      if (!(rr == rr) || c->h_result[S_PQ] == 0.0) { stalled = true; break; }   // NaN or breakdown
      rnorm = sqrt(rr);
    }
    const double prev = res;
    MXG_TRY(true_residual(c, A, b, x, r, &res));
    if (stalled || rnorm > o->tol || !(res < prev)) break;   // cap reached / no progress: report
  }
  *iters = it;
  *res_out = res;
  return 0;
}

}  // namespace

extern "C" int mxg_solve_cg(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, mxg_mv* x,
                            const mxg_solve_opts* o, mxg_solve_info* info) {
  MXG_TRY(mxg_csr_materialize(const_cast<mxg_csr*>(A)));
  MXG_CHECK(c && A && b && x && o && info, "mxg_solve_cg: NULL argument");
  MXG_CHECK(A->nrows == A->ncols_owned, "mxg_solve_cg: matrix must be square");
  MXG_CHECK(b->k == x->k && b->nlocal == A->nrows && x->nlocal == A->nrows,
            "mxg_solve_cg: shape mismatch");
  // the Krylov loop itself is mxg_krylov.h (shared with the CPU backend of this ABI); the fused path is the GPU's own
  return mxg_krylov::solve_cg(c, A, b, x, o, info,
      [&](const mxg_mv* r, mxg_mv* z) { return apply_prec(c, A, o, r, z); },
      [&](mxg_mv* bj, mxg_mv* xj, mxg_mv* r, mxg_mv* z, mxg_mv* p, mxg_mv* q, int max_iters, int* it, double* res) -> int {
        if (!cg_fused_ok(A, o)) return -1;
        MXG_TRY(mxg_ctx_reserve_reduce(c, 8));
        return cg_fused(c, const_cast<mxg_csr*>(A), o, bj, xj, r, z, p, q, max_iters, it, res);
      });
}

extern "C" int mxg_solve_bicgstab(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, mxg_mv* x,
                                  const mxg_solve_opts* o, mxg_solve_info* info) {
  MXG_TRY(mxg_csr_materialize(const_cast<mxg_csr*>(A)));
  MXG_CHECK(c && A && b && x && o && info, "mxg_solve_bicgstab: NULL argument");
  MXG_CHECK(A->nrows == A->ncols_owned, "mxg_solve_bicgstab: matrix must be square");
  MXG_CHECK(b->k == x->k && b->nlocal == A->nrows && x->nlocal == A->nrows, "mxg_solve_bicgstab: shape mismatch");
  return mxg_krylov::solve_bicgstab(c, A, b, x, o, info, [&](const mxg_mv* r, mxg_mv* z) { return apply_prec(c, A, o, r, z); });
}

extern "C" int mxg_solve_gmres(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, mxg_mv* x,
                               const mxg_solve_opts* o, mxg_solve_info* info) {
  MXG_TRY(mxg_csr_materialize(const_cast<mxg_csr*>(A)));
  MXG_CHECK(c && A && b && x && o && info, "mxg_solve_gmres: NULL argument");
  MXG_CHECK(A->nrows == A->ncols_owned, "mxg_solve_gmres: matrix must be square");
  MXG_CHECK(b->k == x->k && b->nlocal == A->nrows && x->nlocal == A->nrows,
            "mxg_solve_gmres: shape mismatch");
  return mxg_krylov::solve_gmres(c, A, b, x, o, info,
      [&](mxg_mv* v, mxg_mv* z, mxg_mv** zc) { return prec_view(c, A, o, v, z, zc); });
}
