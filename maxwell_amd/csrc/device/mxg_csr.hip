// CSR SpMV / SpMM for the MxMagWaveOp operator chain (K1 of SURVEY section 2.2).
// Replaces Epetra_CrsMatrix::Apply behind MxCrsMatrix<S>::apply (reference
// src/MxCrsMatrix.cpp:347-353).
//
// Kernel design (CDNA4, bandwidth bound, <= ~15 nnz/row, no MFMA):
//   * "CSR-stream": a workgroup owns a contiguous block of rows holding <= NNZ_BLK non-zeros.
//     The block's values / column indices are read with fully coalesced 16-byte / 8-byte
//     non-temporal loads regardless of the row lengths, each lane multiplies its entries with
//     the gathered x values and parks the products in LDS; after one barrier every lane sums
//     the products of one row in ascending order.  The result is therefore bit-identical to
//     the serial CSR row loop (product rounded, then added in storage order).
//   * the matrix stream is marked non-temporal so it does not evict the x vector, whose
//     +-x-plane neighbours are re-used from the XCD's 4 MiB L2;
//   * workgroup -> row-block mapping is XCD aware: the 8 XCDs (workgroups are dealt to them
//     round-robin) each walk one contiguous eighth of the rows, so each L2 only ever sees
//     its own slice of x (plus the stencil halo) instead of the whole vector;
//   * k right-hand sides re-use the register-resident matrix entries of the block.
#include <algorithm>

#include "mxg_internal.h"

namespace {

constexpr int SBLK = 256;        // threads per workgroup
constexpr int NNZ_BLK = 2048;    // non-zeros per row block (16 KiB of real products)
constexpr int ROWS_BLK = 1024;   // rows per row block (bounded separately because of empty rows)
constexpr int EPT = NNZ_BLK / SBLK;  // entries per thread
// the real kernel loads aligned pairs starting at (base & ~1), so a block may hold at most
// NNZ_BLK - 1 entries to stay inside the EPT * SBLK window
constexpr int NNZ_CAP = NNZ_BLK - 1;

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef int int2v __attribute__((ext_vector_type(2)));

template <typename T>
__device__ inline T nt_load(const T* p) {
  return __builtin_nontemporal_load(p);
}

// XCD-aware remap of the hardware workgroup id to a logical row block:
// hardware id w runs on XCD (w % 8); give XCD x the contiguous block range
// [x * per, (x+1) * per).  A pure permutation: correctness never depends on placement.
__device__ inline long long xcd_block(long long w, long long nblocks) {
  const long long per = (nblocks + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  return (w % MXG_NUM_XCD) * per + (w / MXG_NUM_XCD);
}

// One row block per workgroup.  CPLX: values and vectors are (re, im) pairs.
template <bool CPLX>
__global__ __launch_bounds__(SBLK) void k_spmv_stream(
    const long long* __restrict__ rowptr, const int* __restrict__ colidx,
    const double* __restrict__ vals, const int* __restrict__ rowblk, long long nblocks,
    MvDesc X, MvDesc Y) {
  constexpr int W = CPLX ? 2 : 1;
  __shared__ double prod[NNZ_BLK * W];
  __shared__ int off[ROWS_BLK + 1];

  const long long b = xcd_block(blockIdx.x, nblocks);
  if (b >= nblocks) return;
  const int r0 = rowblk[b], r1 = rowblk[b + 1];
  const int nr = r1 - r0;
  const long long base = rowptr[r0];
  const int nnzb = (int)(rowptr[r1] - base);

  for (int t = threadIdx.x; t <= nr; t += SBLK) off[t] = (int)(nt_load(rowptr + r0 + t) - base);

  if (nnzb <= NNZ_CAP) {
    // register-resident copy of this block's entries; pairs are loaded from an even
    // (16-byte aligned) start so that the vector loads never straddle alignment
    const long long lo = CPLX ? base : (base & ~1LL);
    double v[EPT][W];
    int c[EPT];
    bool ok[EPT];
    if (CPLX) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const long long idx = lo + threadIdx.x + (long long)e * SBLK;
        ok[e] = idx < base + nnzb;
        if (ok[e]) {
          dbl2 vv = nt_load(reinterpret_cast<const dbl2*>(vals) + idx);
          v[e][0] = vv.x; v[e][W - 1] = vv.y;
          c[e] = nt_load(colidx + idx);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; e += 2) {
        const long long idx = lo + 2 * (threadIdx.x + (long long)(e / 2) * SBLK);
        const bool in0 = idx >= base && idx < base + nnzb;
        const bool in1 = idx + 1 < base + nnzb;   // idx + 1 >= base always holds
        ok[e] = in0; ok[e + 1] = in1;
        if (in0 || in1) {
          dbl2 vv = nt_load(reinterpret_cast<const dbl2*>(vals + idx));
          int2v cc = nt_load(reinterpret_cast<const int2v*>(colidx + idx));
          v[e][0] = vv.x; v[e + 1][0] = vv.y;
          c[e] = cc.x; c[e + 1] = cc.y;
        }
      }
    }
    for (int j = 0; j < X.k; ++j) {
      const double* x = X.col(j);
      double* y = Y.col(j);
      if (j > 0) __syncthreads();   // previous column's row sums are done with prod[]
      if (CPLX) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          if (ok[e]) {
            const int p = (int)(lo - base) + threadIdx.x + e * SBLK;
            double2 xv = reinterpret_cast<const double2*>(x)[c[e]];
            reinterpret_cast<double2*>(prod)[p] =
                make_double2(v[e][0] * xv.x - v[e][W - 1] * xv.y, v[e][0] * xv.y + v[e][W - 1] * xv.x);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          if (ok[e]) {
            const int p = (int)(lo - base) + 2 * (threadIdx.x + (e / 2) * SBLK) + (e & 1);
            prod[p] = v[e][0] * x[c[e]];
          }
        }
      }
      __syncthreads();
      for (int t = threadIdx.x; t < nr; t += SBLK) {
        const int s = off[t], e = off[t + 1];
        if (CPLX) {
          double sr = 0.0, si = 0.0;
          for (int q = s; q < e; ++q) {
            double2 pv = reinterpret_cast<const double2*>(prod)[q];
            sr += pv.x; si += pv.y;
          }
          dbl2 out2; out2.x = sr; out2.y = si;
          __builtin_nontemporal_store(out2, reinterpret_cast<dbl2*>(y) + r0 + t);
        } else {
          double sum = 0.0;
          for (int q = s; q < e; ++q) sum += prod[q];
          __builtin_nontemporal_store(sum, y + r0 + t);
        }
      }
    }
  } else {
    // a single row longer than NNZ_CAP (row blocks never mix such a row with others):
    // the whole workgroup strides over it; partial sums are combined in a fixed order
    __syncthreads();
    for (int j = 0; j < X.k; ++j) {
      const double* x = X.col(j);
      double* y = Y.col(j);
      double sr = 0.0, si = 0.0;
      for (long long q = base + threadIdx.x; q < base + nnzb; q += SBLK) {
        const int cc = colidx[q];
        if (CPLX) {
          double2 vv = reinterpret_cast<const double2*>(vals)[q];
          double2 xv = reinterpret_cast<const double2*>(x)[cc];
          sr += vv.x * xv.x - vv.y * xv.y;
          si += vv.x * xv.y + vv.y * xv.x;
        } else {
          sr += vals[q] * x[cc];
        }
      }
      if (j > 0) __syncthreads();
      prod[threadIdx.x] = sr;
      if (CPLX) prod[SBLK + threadIdx.x] = si;
      __syncthreads();
      if (threadIdx.x == 0) {
        double tr = 0.0, ti = 0.0;
        for (int q = 0; q < SBLK; ++q) { tr += prod[q]; if (CPLX) ti += prod[SBLK + q]; }
        if (CPLX) reinterpret_cast<double2*>(y)[r0] = make_double2(tr, ti);
        else y[r0] = tr;
      }
    }
  }
}

// Reference-style kernel for A/B experiments: one row per group of G lanes, shuffle reduction.
template <int G>
__global__ __launch_bounds__(SBLK) void k_spmv_vector(
    const long long* __restrict__ rowptr, const int* __restrict__ colidx,
    const double* __restrict__ vals, long long nrows, MvDesc X, MvDesc Y) {
  const long long gid = (long long)blockIdx.x * SBLK + threadIdx.x;
  const long long row = gid / G;
  const int sub = threadIdx.x % G;
  for (int j = 0; j < X.k; ++j) {
    const double* x = X.col(j);
    double* y = Y.col(j);
    double s = 0.0;
    if (row < nrows) {
      const long long a = rowptr[row], e = rowptr[row + 1];
      for (long long q = a + sub; q < e; q += G) s += vals[q] * x[colidx[q]];
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) s += __shfl_down(s, o, G);
    if (row < nrows && sub == 0) y[row] = s;
  }
}

// pack x rows into the send buffer / copy owned part into the extended vector
__global__ void k_pack(const double* __restrict__ x, const long long* __restrict__ idx,
                       long long n, int w, double* __restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long r = idx[i];
    for (int c = 0; c < w; ++c) out[i * w + c] = x[r * w + c];
  }
}

__global__ void k_diag(const long long* __restrict__ rowptr, const int* __restrict__ colidx,
                       const double* __restrict__ vals, long long nrows, int w,
                       double* __restrict__ diag) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  double dr = 0.0, di = 0.0;
  for (long long q = rowptr[r]; q < rowptr[r + 1]; ++q)
    if (colidx[q] == r) { dr += vals[q * w]; if (w == 2) di += vals[q * w + 1]; }
  diag[r * w] = dr;
  if (w == 2) diag[r * w + 1] = di;
}

}  // namespace

extern "C" int mxg_csr_create(mxg_ctx* c, int64_t nrows, int64_t ncols_owned, int64_t nghost,
                              const int64_t* rowptr, const int32_t* colidx, const double* vals,
                              int is_complex, const mxg_halo_plan* plan, mxg_csr** out) {
  MXG_CHECK(c && rowptr && out, "mxg_csr_create: NULL argument");
  MXG_CHECK(nrows >= 0 && ncols_owned >= 0 && nghost >= 0, "mxg_csr_create: negative size");
  MXG_CHECK(nrows < (int64_t)1 << 31 && ncols_owned + nghost < (int64_t)1 << 31,
            "mxg_csr_create: local sizes must fit int32 column / row-block indices");
  MXG_CHECK(rowptr[0] == 0, "mxg_csr_create: rowptr[0] must be 0");
  MXG_CHECK(nghost == 0 || plan != nullptr, "mxg_csr_create: ghosts without a halo plan");
  const int64_t nnz = rowptr[nrows];
  const int64_t ncols = ncols_owned + nghost;
  for (int64_t r = 0; r < nrows; ++r)
    MXG_CHECK(rowptr[r + 1] >= rowptr[r], "mxg_csr_create: rowptr not monotone at row %lld",
              (long long)r);
  for (int64_t q = 0; q < nnz; ++q)
    MXG_CHECK(colidx[q] >= 0 && colidx[q] < ncols,
              "mxg_csr_create: column index %d out of range [0,%lld) at entry %lld", colidx[q],
              (long long)ncols, (long long)q);
  MXG_HIP(hipSetDevice(c->device));
  mxg_csr* a = new mxg_csr();
  a->ctx = c;
  a->nrows = nrows; a->ncols_owned = ncols_owned; a->nghost = nghost; a->nnz = nnz;
  a->is_complex = is_complex ? 1 : 0;
  const int w = is_complex ? 2 : 1;

  // row blocks: greedy, <= NNZ_BLK non-zeros and <= ROWS_BLK rows; an over-long row stands alone
  std::vector<int32_t> blk;
  blk.push_back(0);
  int64_t r = 0;
  while (r < nrows) {
    int64_t e = r;
    const int64_t start = rowptr[r];
    while (e < nrows && e - r < ROWS_BLK && rowptr[e + 1] - start <= NNZ_CAP) ++e;
    if (e == r) e = r + 1;  // single long row
    blk.push_back((int32_t)e);
    r = e;
  }
  a->nblocks = (int64_t)blk.size() - 1;

  // +2 entries of slack: the real kernel loads aligned pairs and may touch one element past nnz
  MXG_HIP(hipMalloc(&a->d_rowptr, sizeof(int64_t) * (nrows + 1)));
  MXG_HIP(hipMalloc(&a->d_colidx, sizeof(int32_t) * (nnz + 2)));
  MXG_HIP(hipMalloc(&a->d_vals, sizeof(double) * (nnz + 2) * w));
  MXG_HIP(hipMalloc(&a->d_rowblk, sizeof(int32_t) * blk.size()));
  MXG_HIP(hipMemsetAsync(a->d_colidx, 0, sizeof(int32_t) * (nnz + 2), c->stream));
  MXG_HIP(hipMemsetAsync(a->d_vals, 0, sizeof(double) * (nnz + 2) * w, c->stream));
  MXG_HIP(hipMemcpyAsync(a->d_rowptr, rowptr, sizeof(int64_t) * (nrows + 1), hipMemcpyHostToDevice,
                         c->stream));
  if (nnz > 0) {
    MXG_HIP(hipMemcpyAsync(a->d_colidx, colidx, sizeof(int32_t) * nnz, hipMemcpyHostToDevice,
                           c->stream));
    MXG_HIP(hipMemcpyAsync(a->d_vals, vals, sizeof(double) * nnz * w, hipMemcpyHostToDevice,
                           c->stream));
  }
  MXG_HIP(hipMemcpyAsync(a->d_rowblk, blk.data(), sizeof(int32_t) * blk.size(),
                         hipMemcpyHostToDevice, c->stream));
  // diagonal for the Jacobi / Chebyshev preconditioners (square part only)
  MXG_HIP(hipMalloc(&a->d_diag, sizeof(double) * std::max<int64_t>(nrows, 1) * w));
  if (nrows > 0) {
    hipLaunchKernelGGL(k_diag, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream,
                       (const long long*)a->d_rowptr, a->d_colidx, a->d_vals, (long long)nrows, w,
                       a->d_diag);
    MXG_HIP(hipGetLastError());
  }
  if (plan && plan->nneigh > 0) {
    a->nneigh = plan->nneigh;
    a->neigh.assign(plan->neigh, plan->neigh + plan->nneigh);
    a->send_off.assign(plan->send_off, plan->send_off + plan->nneigh + 1);
    a->recv_off.assign(plan->recv_off, plan->recv_off + plan->nneigh + 1);
    a->nsend = a->send_off[a->nneigh];
    MXG_CHECK(a->recv_off[a->nneigh] == nghost, "mxg_csr_create: halo plan receives %lld ghosts, matrix has %lld",
              (long long)a->recv_off[a->nneigh], (long long)nghost);
    for (int64_t i = 0; i < a->nsend; ++i)
      MXG_CHECK(plan->send_idx[i] >= 0 && plan->send_idx[i] < ncols_owned,
                "mxg_csr_create: send index out of range");
    MXG_HIP(hipMalloc(&a->d_send_idx, sizeof(int64_t) * std::max<int64_t>(a->nsend, 1)));
    if (a->nsend > 0)
      MXG_HIP(hipMemcpyAsync(a->d_send_idx, plan->send_idx, sizeof(int64_t) * a->nsend,
                             hipMemcpyHostToDevice, c->stream));
  }
  MXG_HIP(hipStreamSynchronize(c->stream));
  *out = a;
  return 0;
}

extern "C" int mxg_csr_destroy(mxg_csr* a) {
  if (!a) return 0;
  hipSetDevice(a->ctx->device);
  hipStreamSynchronize(a->ctx->stream);
  hipFree(a->d_rowptr); hipFree(a->d_colidx); hipFree(a->d_vals); hipFree(a->d_rowblk);
  hipFree(a->d_diag);
  if (a->d_send_idx) hipFree(a->d_send_idx);
  if (a->d_sendbuf) hipFree(a->d_sendbuf);
  if (a->d_xext) hipFree(a->d_xext);
  delete a;
  return 0;
}

extern "C" int mxg_csr_info(const mxg_csr* a, int64_t* nrows, int64_t* ncols_owned, int64_t* nghost,
                            int64_t* nnz, int* is_complex) {
  if (nrows) *nrows = a->nrows;
  if (ncols_owned) *ncols_owned = a->ncols_owned;
  if (nghost) *nghost = a->nghost;
  if (nnz) *nnz = a->nnz;
  if (is_complex) *is_complex = a->is_complex;
  return 0;
}

extern "C" int64_t mxg_csr_apply_bytes(const mxg_csr* a, int k) {
  const int64_t sb = a->is_complex ? 16 : 8;
  return a->nnz * (sb + 4) + a->nrows * (8 + sb * k) + (a->ncols_owned + a->nghost) * sb * k;
}

extern "C" int mxg_csr_set_kernel(mxg_csr* a, int which) {
  a->kernel = which;
  return 0;
}

// C1: ghost import.  Packs the rows each neighbour needs, exchanges them with one grouped
// ncclSend/ncclRecv per neighbour, and assembles [owned | ghosts] per column in d_xext.
static int halo_exchange(mxg_ctx* c, mxg_csr* a, const mxg_mv* x, MvDesc* xext) {
  const int w = a->is_complex ? 2 : 1;
  const int k = x->k;
  const int64_t next = a->ncols_owned + a->nghost;
  const int64_t ldx = ((next * w) + 31) & ~(int64_t)31;
  if (k > a->halo_kcap) {
    MXG_HIP(hipStreamSynchronize(c->stream));
    if (a->d_sendbuf) hipFree(a->d_sendbuf);
    if (a->d_xext) hipFree(a->d_xext);
    MXG_HIP(hipMalloc(&a->d_sendbuf, sizeof(double) * std::max<int64_t>(a->nsend, 1) * w * k));
    MXG_HIP(hipMalloc(&a->d_xext, sizeof(double) * ldx * k));
    a->halo_kcap = k;
  }
  for (int j = 0; j < k; ++j) {
    double* dst = a->d_xext + ldx * j;
    MXG_HIP(hipMemcpyAsync(dst, x->colptr(j), sizeof(double) * a->ncols_owned * w,
                           hipMemcpyDeviceToDevice, c->stream));
    if (a->nsend > 0) {
      const int grid = mxg_grid_for(a->nsend, 256, 1, 4 * c->num_cu);
      hipLaunchKernelGGL(k_pack, dim3(grid), dim3(256), 0, c->stream, x->colptr(j),
                         (const long long*)a->d_send_idx, (long long)a->nsend, w,
                         a->d_sendbuf + a->nsend * w * j);
    }
  }
  MXG_HIP(hipGetLastError());
  MXG_CHECK(c->comm != nullptr, "halo exchange needs an RCCL communicator (nranks > 1)");
  MXG_NCCL(ncclGroupStart());
  for (int j = 0; j < k; ++j) {
    for (int p = 0; p < a->nneigh; ++p) {
      const int64_t ns = (a->send_off[p + 1] - a->send_off[p]) * w;
      const int64_t nr = (a->recv_off[p + 1] - a->recv_off[p]) * w;
      if (ns > 0)
        MXG_NCCL(ncclSend(a->d_sendbuf + (a->nsend * j + a->send_off[p]) * w, ns, ncclDouble,
                          a->neigh[p], c->comm, c->stream));
      if (nr > 0)
        MXG_NCCL(ncclRecv(a->d_xext + ldx * j + (a->ncols_owned + a->recv_off[p]) * w, nr,
                          ncclDouble, a->neigh[p], c->comm, c->stream));
    }
  }
  MXG_NCCL(ncclGroupEnd());
  xext->base = a->d_xext; xext->ld = ldx; xext->n = next; xext->k = k; xext->cplx = a->is_complex;
  xext->col0 = 0; xext->cols = nullptr;
  return 0;
}

extern "C" int mxg_csr_apply(mxg_ctx* c, const mxg_csr* a_, const mxg_mv* x, mxg_mv* y) {
  mxg_csr* a = const_cast<mxg_csr*>(a_);
  MXG_CHECK(c && a && x && y, "mxg_csr_apply: NULL argument");
  MXG_CHECK(x->k == y->k, "mxg_csr_apply: x has %d columns, y has %d", x->k, y->k);
  MXG_CHECK(x->nlocal == a->ncols_owned, "mxg_csr_apply: x has %lld rows, matrix has %lld owned columns",
            (long long)x->nlocal, (long long)a->ncols_owned);
  MXG_CHECK(y->nlocal == a->nrows, "mxg_csr_apply: y has %lld rows, matrix has %lld rows",
            (long long)y->nlocal, (long long)a->nrows);
  MXG_CHECK(x->is_complex == a->is_complex && y->is_complex == a->is_complex,
            "mxg_csr_apply: scalar type mismatch");
  MvDesc X = x->desc();
  const MvDesc Y = y->desc();
  if (a->nghost > 0 || a->nneigh > 0) {
    MXG_TRY(halo_exchange(c, a, x, &X));
  } else if (x->base == y->base) {
    // possible aliasing (apply(x, x), src/MxMagWaveOp.cpp:650,1249): work from a copy of x,
    // like Epetra_CrsMatrix::Multiply does when X and Y share storage
    bool overlap = false;
    for (int i = 0; i < x->k && !overlap; ++i)
      for (int j = 0; j < y->k; ++j)
        if (x->hcols[i] == y->hcols[j]) { overlap = true; break; }
    if (overlap) {
      MXG_TRY(mxg_ctx_reserve_scratch(c, (size_t)x->ld * x->k));
      for (int j = 0; j < x->k; ++j)
        MXG_HIP(hipMemcpyAsync(c->d_scratch + (size_t)x->ld * j, x->colptr(j),
                               sizeof(double) * x->ld, hipMemcpyDeviceToDevice, c->stream));
      X.base = c->d_scratch; X.col0 = 0; X.cols = nullptr;
    }
  }
  if (a->nrows == 0) return 0;
  if (a->kernel == 1 && !a->is_complex) {
    const long long threads = (long long)a->nrows * 8;
    hipLaunchKernelGGL((k_spmv_vector<8>), dim3((unsigned)((threads + SBLK - 1) / SBLK)), dim3(SBLK),
                       0, c->stream, (const long long*)a->d_rowptr, a->d_colidx, a->d_vals,
                       (long long)a->nrows, X, Y);
  } else {
    const long long per = (a->nblocks + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
    const unsigned grid = (unsigned)(per * MXG_NUM_XCD);
    if (a->is_complex)
      hipLaunchKernelGGL((k_spmv_stream<true>), dim3(grid), dim3(SBLK), 0, c->stream,
                         (const long long*)a->d_rowptr, a->d_colidx, a->d_vals, a->d_rowblk,
                         (long long)a->nblocks, X, Y);
    else
      hipLaunchKernelGGL((k_spmv_stream<false>), dim3(grid), dim3(SBLK), 0, c->stream,
                         (const long long*)a->d_rowptr, a->d_colidx, a->d_vals, a->d_rowblk,
                         (long long)a->nblocks, X, Y);
  }
  MXG_HIP(hipGetLastError());
  return 0;
}
