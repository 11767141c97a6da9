// Setup-time sparse algebra on the device: the operator chain of MxMagWaveOp::initMatrices
//   curlCurl = curlE dmL [invEps] curlB,   gradDiv = dmA gradPsi dmVInv divB dmA,   vecLapl = purge(curlCurl - gradDiv),
//   shifted = vecLapl - sigma mRhs                                  (src/MxMagWaveOp.cpp:138-273)
// is formed from the single Yee operators without a host round trip.  Replaces
//   MxCrsMatrix::matrixMatrixMultiply  (src/MxCrsMatrix.cpp:361-382, EpetraExt::MatrixMatrix::Multiply)
//   MxCrsMatrix::matrixMatrixAdd       (src/MxCrsMatrix.cpp:404-457, EpetraExt::MatrixMatrix::Add)
//   MxCrsMatrix::purgeZeros            (src/MxCrsMatrix.cpp:84-116)
// with exactly the arithmetic of the host versions in include/mx/mx_sparse.hpp, so that the device chain is
// BIT-IDENTICAL to the host chain (and to the oracle):
//   multiply  Gustavson per row, one thread per row: partial products of an output entry are accumulated in the order
//             they are met (row of A in storage order, rows of B in storage order), entries emitted by ascending column;
//             a diagonal factor is a row / column scaling on the other factor's pattern;
//   add       union pattern, sa * a + sb * b where both are present;
//   purge     keeps |a| > tol.
// Real and complex scalars (both operands of one kind).
// These kernels run once per problem; they are latency-tolerant row loops, not bandwidth kernels.
#include <algorithm>

#include "mxg_internal.h"

namespace {

constexpr int SA_BLOCK = 256;

// ---- exclusive scan of int64 counts (n + 1 outputs): per-block sums, a serial pass over the block sums, fix-up
__global__ __launch_bounds__(SA_BLOCK) void k_scan_block(const long long* __restrict__ in, long long n,
                                                          long long* __restrict__ out, long long* __restrict__ bsum) {
  __shared__ long long s[SA_BLOCK];
  const long long i = (long long)blockIdx.x * SA_BLOCK + threadIdx.x;
  s[threadIdx.x] = i < n ? in[i] : 0;
  __syncthreads();
  for (int off = 1; off < SA_BLOCK; off <<= 1) {            // Hillis-Steele inclusive scan
    const long long v = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
    __syncthreads();
    s[threadIdx.x] += v;
    __syncthreads();
  }
  if (i < n) out[i + 1] = s[threadIdx.x];
  if (threadIdx.x == SA_BLOCK - 1) bsum[blockIdx.x] = s[threadIdx.x];
}
// exclusive scan of the block sums in place, one workgroup: a contiguous chunk per thread, the chunk totals scanned in LDS
// (a single thread used to walk the 80 000 sums of a 256^3 operator: 7 ms per scan, nine scans per chain)
__global__ __launch_bounds__(1024) void k_scan_sums(long long* bsum, long long nb) {
  __shared__ long long tot[1024];
  const long long per = (nb + 1023) / 1024, b0 = (long long)threadIdx.x * per, b1 = b0 + per < nb ? b0 + per : nb;
  long long acc = 0;
  for (long long b = b0; b < b1; ++b) acc += bsum[b];
  tot[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const long long v = (int)threadIdx.x >= off ? tot[threadIdx.x - off] : 0;
    __syncthreads();
    tot[threadIdx.x] += v;
    __syncthreads();
  }
  acc = threadIdx.x > 0 ? tot[threadIdx.x - 1] : 0;
  for (long long b = b0; b < b1; ++b) { const long long v = bsum[b]; bsum[b] = acc; acc += v; }
}
__global__ __launch_bounds__(SA_BLOCK) void k_scan_fix(long long* __restrict__ out, long long n,
                                                        const long long* __restrict__ bsum) {
  const long long i = (long long)blockIdx.x * SA_BLOCK + threadIdx.x;
  if (i < n) out[i + 1] += bsum[blockIdx.x];
  if (i == 0) out[0] = 0;
}

int exclusive_scan(mxg_ctx* c, const long long* d_in, long long n, long long* d_out /* n + 1 */) {
  const long long nb = std::max<long long>((n + SA_BLOCK - 1) / SA_BLOCK, 1);
  struct Buf { long long* p = nullptr; ~Buf() { if (p) hipFree(p); } } hold;   // freed on every exit path
  MXG_HIP(hipMalloc(&hold.p, sizeof(long long) * nb));
  long long* bsum = hold.p;
  hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nb), dim3(SA_BLOCK), 0, c->stream, d_in, n, d_out, bsum);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, c->stream, bsum, nb);
  hipLaunchKernelGGL(k_scan_fix, dim3((unsigned)nb), dim3(SA_BLOCK), 0, c->stream, d_out, n, (const long long*)bsum);
  MXG_HIP(hipGetLastError());
  MXG_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- multiply
__global__ __launch_bounds__(SA_BLOCK) void k_mm_upper(const long long* __restrict__ arp, const int* __restrict__ aci,
                                                        const long long* __restrict__ brp, long long nrows,
                                                        long long* __restrict__ ub) {
  const long long i = (long long)blockIdx.x * SA_BLOCK + threadIdx.x;
  if (i >= nrows) return;
  long long n = 0;
  for (long long p = arp[i]; p < arp[i + 1]; ++p) n += brp[aci[p] + 1] - brp[aci[p]];
  ub[i] = n;
}

// values of W doubles (1: real, 2: complex as (re, im)); the complex product and sum are those std::complex<double>
// forms on the host for finite operands - (ac - bd, ad + bc), no contraction - so that both chains agree bit for bit
template <int W> struct Val;
template <> struct Val<1> {
  double x;
  __device__ static Val ld(const double* p, long long i) { Val v; v.x = p[i]; return v; }
  __device__ void st(double* p, long long i) const { p[i] = x; }
  __device__ Val mul(const Val& o) const { Val v; v.x = x * o.x; return v; }
  __device__ Val add(const Val& o) const { Val v; v.x = x + o.x; return v; }
  __device__ double mag() const { return fabs(x); }
};
template <> struct Val<2> {
  double x, y;
  __device__ static Val ld(const double* p, long long i) { Val v; v.x = p[2 * i]; v.y = p[2 * i + 1]; return v; }
  __device__ void st(double* p, long long i) const { p[2 * i] = x; p[2 * i + 1] = y; }
  __device__ Val mul(const Val& o) const { Val v; v.x = x * o.x - y * o.y; v.y = x * o.y + y * o.x; return v; }
  __device__ Val add(const Val& o) const { Val v; v.x = x + o.x; v.y = y + o.y; return v; }
  __device__ double mag() const { return hypot(x, y); }          // std::abs(std::complex)
};

// Gustavson row: accumulate in meeting order (linear search of the row's own slice), then insertion sort by column
template <int W>
__global__ __launch_bounds__(SA_BLOCK) void k_mm_rows(const long long* __restrict__ arp, const int* __restrict__ aci,
                                                       const double* __restrict__ ava, const long long* __restrict__ brp,
                                                       const int* __restrict__ bci, const double* __restrict__ bva,
                                                       long long nrows, const long long* __restrict__ ub,
                                                       int* __restrict__ tc, double* __restrict__ tv,
                                                       long long* __restrict__ cnt) {
  const long long i = (long long)blockIdx.x * SA_BLOCK + threadIdx.x;
  if (i >= nrows) return;
  int* c = tc + ub[i];
  double* v = tv + ub[i] * W;
  long long n = 0;
  for (long long p = arp[i]; p < arp[i + 1]; ++p) {
    const int k = aci[p];
    const Val<W> a = Val<W>::ld(ava, p);
    for (long long q = brp[k]; q < brp[k + 1]; ++q) {
      const int j = bci[q];
      const Val<W> prod = a.mul(Val<W>::ld(bva, q));
      long long f = 0;
      while (f < n && c[f] != j) ++f;
      if (f < n) Val<W>::ld(v, f).add(prod).st(v, f); else { c[n] = j; prod.st(v, n); ++n; }
    }
  }
  for (long long x = 1; x < n; ++x) {
    const int cc = c[x];
    const Val<W> vv = Val<W>::ld(v, x);
    long long y = x - 1;
    while (y >= 0 && c[y] > cc) { c[y + 1] = c[y]; Val<W>::ld(v, y).st(v, y + 1); --y; }
    c[y + 1] = cc; vv.st(v, y + 1);
  }
  cnt[i] = n;
}

template <int W>
__global__ __launch_bounds__(SA_BLOCK) void k_compact(const long long* __restrict__ ub, const int* __restrict__ tc,
                                                       const double* __restrict__ tv, long long nrows,
                                                       const long long* __restrict__ crp, int* __restrict__ cci,
                                                       double* __restrict__ cva) {
  const long long i = (long long)blockIdx.x * SA_BLOCK + threadIdx.x;
  if (i >= nrows) return;
  const long long n = crp[i + 1] - crp[i];
  for (long long x = 0; x < n; ++x) { cci[crp[i] + x] = tc[ub[i] + x]; Val<W>::ld(tv, ub[i] + x).st(cva, crp[i] + x); }
}

// diagonal factor: rows of B scaled by d[i] (A diagonal) or entries of A scaled by d[col] (B diagonal)
template <int W>
__global__ __launch_bounds__(SA_BLOCK) void k_scale_rows(const long long* __restrict__ rp, const double* __restrict__ va,
                                                          const double* __restrict__ d, long long nrows,
                                                          double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * SA_BLOCK + threadIdx.x;
  if (i >= nrows) return;
  const Val<W> di = Val<W>::ld(d, i);
  for (long long p = rp[i]; p < rp[i + 1]; ++p) di.mul(Val<W>::ld(va, p)).st(out, p);
}
template <int W>
__global__ __launch_bounds__(SA_BLOCK) void k_scale_cols(const int* __restrict__ ci, const double* __restrict__ va,
                                                          const double* __restrict__ d, long long nnz,
                                                          double* __restrict__ out) {
  const long long stride = (long long)gridDim.x * SA_BLOCK;
  for (long long p = (long long)blockIdx.x * SA_BLOCK + threadIdx.x; p < nnz; p += stride)
    Val<W>::ld(va, p).mul(Val<W>::ld(d, ci[p])).st(out, p);
}

// ---- add: count, then fill (two-pointer merge of the sorted rows)
__global__ __launch_bounds__(SA_BLOCK) void k_add_count(const long long* __restrict__ arp, const int* __restrict__ aci,
                                                         const long long* __restrict__ brp, const int* __restrict__ bci,
                                                         long long nrows, long long* __restrict__ cnt) {
  const long long i = (long long)blockIdx.x * SA_BLOCK + threadIdx.x;
  if (i >= nrows) return;
  long long p = arp[i], pe = arp[i + 1], q = brp[i], qe = brp[i + 1], n = 0;
  while (p < pe || q < qe) {
    if (q >= qe || (p < pe && aci[p] < bci[q])) ++p;
    else if (p >= pe || bci[q] < aci[p]) ++q;
    else { ++p; ++q; }
    ++n;
  }
  cnt[i] = n;
}
template <int W>
__global__ __launch_bounds__(SA_BLOCK) void k_add_fill(const long long* __restrict__ arp, const int* __restrict__ aci,
                                                        const double* __restrict__ ava, Val<W> sa,
                                                        const long long* __restrict__ brp, const int* __restrict__ bci,
                                                        const double* __restrict__ bva, Val<W> sb, long long nrows,
                                                        const long long* __restrict__ crp, int* __restrict__ cci,
                                                        double* __restrict__ cva) {
  const long long i = (long long)blockIdx.x * SA_BLOCK + threadIdx.x;
  if (i >= nrows) return;
  long long p = arp[i], pe = arp[i + 1], q = brp[i], qe = brp[i + 1], w = crp[i];
  while (p < pe || q < qe) {
    if (q >= qe || (p < pe && aci[p] < bci[q])) { cci[w] = aci[p]; sa.mul(Val<W>::ld(ava, p)).st(cva, w); ++p; }
    else if (p >= pe || bci[q] < aci[p]) { cci[w] = bci[q]; sb.mul(Val<W>::ld(bva, q)).st(cva, w); ++q; }
    else { cci[w] = aci[p]; sa.mul(Val<W>::ld(ava, p)).add(sb.mul(Val<W>::ld(bva, q))).st(cva, w); ++p; ++q; }
    ++w;
  }
}

// ---- purge
template <int W>
__global__ __launch_bounds__(SA_BLOCK) void k_purge_count(const long long* __restrict__ rp, const double* __restrict__ va,
                                                           double tol, long long nrows, long long* __restrict__ cnt) {
  const long long i = (long long)blockIdx.x * SA_BLOCK + threadIdx.x;
  if (i >= nrows) return;
  long long n = 0;
  for (long long p = rp[i]; p < rp[i + 1]; ++p) if (Val<W>::ld(va, p).mag() > tol) ++n;
  cnt[i] = n;
}
template <int W>
__global__ __launch_bounds__(SA_BLOCK) void k_purge_fill(const long long* __restrict__ rp, const int* __restrict__ ci,
                                                          const double* __restrict__ va, double tol, long long nrows,
                                                          const long long* __restrict__ crp, int* __restrict__ cci,
                                                          double* __restrict__ cva) {
  const long long i = (long long)blockIdx.x * SA_BLOCK + threadIdx.x;
  if (i >= nrows) return;
  long long w = crp[i];
  for (long long p = rp[i]; p < rp[i + 1]; ++p) {
    const Val<W> v = Val<W>::ld(va, p);
    if (v.mag() > tol) { cci[w] = ci[p]; v.st(cva, w); ++w; }
  }
}

__global__ void k_is_diagonal(const long long* __restrict__ rp, const int* __restrict__ ci, long long nrows, int* flag) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  if (rp[i + 1] - rp[i] != 1 || ci[rp[i]] != i) *flag = 0;
}

// device temporaries of one call: freed on every exit path (an allocation failure at 256^3 / 512^3 used to leak the GBs
// already allocated); keep() hands a finished array over to the result matrix
struct DevPool {
  std::vector<void*> p;
  ~DevPool() { for (void* q : p) hipFree(q); }
  template <typename T>
  int alloc(T** out, size_t bytes) {
    void* q = nullptr;
    MXG_HIP(hipMalloc(&q, bytes));
    p.push_back(q);
    *out = static_cast<T*>(q);
    return 0;
  }
  void keep(void* q) { p.erase(std::remove(p.begin(), p.end(), q), p.end()); }
};

inline unsigned rows_grid(long long nrows) { return (unsigned)std::max<long long>((nrows + SA_BLOCK - 1) / SA_BLOCK, 1); }

int is_diagonal(mxg_ctx* c, const mxg_csr* a, bool* out) {
  DevPool pool;
  *out = false;
  if (a->nrows != a->ncols_owned + a->nghost || a->nnz != a->nrows) return 0;
  int* d_flag = nullptr;
  int h = 1;
  MXG_TRY(pool.alloc(&d_flag, sizeof(int)));
  MXG_HIP(hipMemcpyAsync(d_flag, &h, sizeof(int), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_is_diagonal, dim3(rows_grid(a->nrows)), dim3(SA_BLOCK), 0, c->stream,
                     (const long long*)a->d_rowptr, (const int*)a->d_colidx, (long long)a->nrows, d_flag);
  MXG_HIP(hipMemcpyAsync(&h, d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  *out = h != 0;
  return 0;
}

int check_operands(mxg_ctx* c, const mxg_csr* a, const mxg_csr* b, const char* what) {
  MXG_CHECK(c && a && b, "%s: NULL argument", what);
  MXG_CHECK(a->is_complex == b->is_complex, "%s: one operand is real, the other complex", what);
  MXG_CHECK(a->nghost == 0 && b->nghost == 0 && a->nneigh == 0 && b->nneigh == 0,
            "%s: row-partitioned matrices are formed on the window by the host assembler", what);
  return 0;
}

}  // namespace

// defined in mxg_csr.hip: wrap finished device CSR arrays (rowptr nrows + 1, colidx / vals with 8 entries of slack)
int mxg_csr_adopt_device(mxg_ctx* c, int64_t nrows, int64_t ncols, int64_t nnz, int64_t* d_rowptr, int32_t* d_colidx,
                         double* d_vals, int is_complex, mxg_csr** out);

extern "C" int mxg_csr_multiply(mxg_ctx* c, const mxg_csr* a, const mxg_csr* b, mxg_csr** out) {
  DevPool pool;
  MXG_TRY(check_operands(c, a, b, "mxg_csr_multiply"));
  MXG_CHECK(out, "mxg_csr_multiply: NULL argument");
  MxgSetupTrace trace("mxg_csr_multiply (nnz a)", a->nnz);
  MXG_CHECK(a->ncols_owned == b->nrows, "mxg_csr_multiply: A has %lld columns, B has %lld rows", (long long)a->ncols_owned,
            (long long)b->nrows);
  MXG_HIP(hipSetDevice(c->device));
  const long long nrows = a->nrows;
  const int cx = a->is_complex ? 1 : 0, w = cx ? 2 : 1;
  bool adiag = false, bdiag = false;
  MXG_TRY(is_diagonal(c, a, &adiag));
  if (!adiag) MXG_TRY(is_diagonal(c, b, &bdiag));
  if (adiag || bdiag) {
    // a diagonal factor: the other factor's pattern, each value one product (what the general loop would form)
    const mxg_csr* pat = adiag ? b : a;
    int64_t* rp = nullptr; int32_t* ci = nullptr; double* va = nullptr;
    MXG_TRY(pool.alloc(&rp, sizeof(int64_t) * (pat->nrows + 1)));
    MXG_TRY(pool.alloc(&ci, sizeof(int32_t) * (pat->nnz + 8)));
    MXG_TRY(pool.alloc(&va, sizeof(double) * (pat->nnz + 8) * w));
    MXG_HIP(hipMemsetAsync(ci, 0, sizeof(int32_t) * (pat->nnz + 8), c->stream));
    MXG_HIP(hipMemsetAsync(va, 0, sizeof(double) * (pat->nnz + 8) * w, c->stream));
    MXG_HIP(hipMemcpyAsync(rp, pat->d_rowptr, sizeof(int64_t) * (pat->nrows + 1), hipMemcpyDeviceToDevice, c->stream));
    if (pat->nnz > 0) MXG_HIP(hipMemcpyAsync(ci, pat->d_colidx, sizeof(int32_t) * pat->nnz, hipMemcpyDeviceToDevice, c->stream));
    const unsigned gcols = (unsigned)std::min<long long>((pat->nnz + SA_BLOCK - 1) / SA_BLOCK, 65535);
    if (adiag) {
      if (cx) hipLaunchKernelGGL(k_scale_rows<2>, dim3(rows_grid(pat->nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)b->d_rowptr,
                                 (const double*)b->d_vals, (const double*)a->d_vals, (long long)b->nrows, va);
      else hipLaunchKernelGGL(k_scale_rows<1>, dim3(rows_grid(pat->nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)b->d_rowptr,
                              (const double*)b->d_vals, (const double*)a->d_vals, (long long)b->nrows, va);
    } else if (pat->nnz > 0) {
      if (cx) hipLaunchKernelGGL(k_scale_cols<2>, dim3(gcols), dim3(SA_BLOCK), 0, c->stream, (const int*)a->d_colidx,
                                 (const double*)a->d_vals, (const double*)b->d_vals, (long long)a->nnz, va);
      else hipLaunchKernelGGL(k_scale_cols<1>, dim3(gcols), dim3(SA_BLOCK), 0, c->stream, (const int*)a->d_colidx,
                              (const double*)a->d_vals, (const double*)b->d_vals, (long long)a->nnz, va);
    }
    MXG_HIP(hipGetLastError());
    pool.keep(rp); pool.keep(ci); pool.keep(va);
    return mxg_csr_adopt_device(c, pat->nrows, b->ncols_owned, pat->nnz, rp, ci, va, cx, out);
  }
  long long *ub_cnt = nullptr, *ub = nullptr, *cnt = nullptr;
  MXG_TRY(pool.alloc(&ub_cnt, sizeof(long long) * std::max<long long>(nrows, 1)));
  MXG_TRY(pool.alloc(&ub, sizeof(long long) * (nrows + 1)));
  MXG_TRY(pool.alloc(&cnt, sizeof(long long) * std::max<long long>(nrows, 1)));
  hipLaunchKernelGGL(k_mm_upper, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)a->d_rowptr,
                     (const int*)a->d_colidx, (const long long*)b->d_rowptr, nrows, ub_cnt);
  MXG_TRY(exclusive_scan(c, ub_cnt, nrows, ub));
  long long total = 0;
  MXG_HIP(hipMemcpy(&total, ub + nrows, sizeof(long long), hipMemcpyDeviceToHost));
  int* tc = nullptr; double* tv = nullptr;
  MXG_TRY(pool.alloc(&tc, sizeof(int) * std::max<long long>(total, 1)));
  MXG_TRY(pool.alloc(&tv, sizeof(double) * std::max<long long>(total, 1) * w));
  if (cx) hipLaunchKernelGGL(k_mm_rows<2>, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)a->d_rowptr,
                             (const int*)a->d_colidx, (const double*)a->d_vals, (const long long*)b->d_rowptr, (const int*)b->d_colidx,
                             (const double*)b->d_vals, nrows, (const long long*)ub, tc, tv, cnt);
  else hipLaunchKernelGGL(k_mm_rows<1>, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)a->d_rowptr,
                          (const int*)a->d_colidx, (const double*)a->d_vals, (const long long*)b->d_rowptr, (const int*)b->d_colidx,
                          (const double*)b->d_vals, nrows, (const long long*)ub, tc, tv, cnt);
  MXG_HIP(hipGetLastError());
  int64_t* rp = nullptr;
  MXG_TRY(pool.alloc(&rp, sizeof(int64_t) * (nrows + 1)));
  MXG_TRY(exclusive_scan(c, cnt, nrows, (long long*)rp));
  long long nnz = 0;
  MXG_HIP(hipMemcpy(&nnz, rp + nrows, sizeof(long long), hipMemcpyDeviceToHost));
  int32_t* ci = nullptr; double* va = nullptr;
  MXG_TRY(pool.alloc(&ci, sizeof(int32_t) * (nnz + 8)));
  MXG_TRY(pool.alloc(&va, sizeof(double) * (nnz + 8) * w));
  MXG_HIP(hipMemsetAsync(ci, 0, sizeof(int32_t) * (nnz + 8), c->stream));
  MXG_HIP(hipMemsetAsync(va, 0, sizeof(double) * (nnz + 8) * w, c->stream));
  if (cx) hipLaunchKernelGGL(k_compact<2>, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)ub, (const int*)tc,
                             (const double*)tv, nrows, (const long long*)rp, ci, va);
  else hipLaunchKernelGGL(k_compact<1>, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)ub, (const int*)tc,
                          (const double*)tv, nrows, (const long long*)rp, ci, va);
  MXG_HIP(hipGetLastError());
  MXG_HIP(hipStreamSynchronize(c->stream));
  pool.keep(rp); pool.keep(ci); pool.keep(va);
  return mxg_csr_adopt_device(c, nrows, b->ncols_owned, nnz, rp, ci, va, cx, out);
}

extern "C" int mxg_csr_add_z(mxg_ctx* c, const mxg_csr* a, const double sa[2], const mxg_csr* b, const double sb[2], mxg_csr** out) {
  DevPool pool;
  MXG_CHECK(a && b && sa && sb, "mxg_csr_add: NULL argument");
  MxgSetupTrace trace("mxg_csr_add (nnz a)", a->nnz);
  MXG_TRY(check_operands(c, a, b, "mxg_csr_add"));
  MXG_CHECK(out, "mxg_csr_add: NULL argument");
  MXG_CHECK(a->nrows == b->nrows && a->ncols_owned == b->ncols_owned, "mxg_csr_add: shapes differ");
  const int cx = a->is_complex ? 1 : 0, w = cx ? 2 : 1;
  MXG_CHECK(cx || (sa[1] == 0.0 && sb[1] == 0.0), "mxg_csr_add: complex scalars on real matrices");
  MXG_HIP(hipSetDevice(c->device));
  const long long nrows = a->nrows;
  long long* cnt = nullptr;
  MXG_TRY(pool.alloc(&cnt, sizeof(long long) * std::max<long long>(nrows, 1)));
  hipLaunchKernelGGL(k_add_count, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)a->d_rowptr,
                     (const int*)a->d_colidx, (const long long*)b->d_rowptr, (const int*)b->d_colidx, nrows, cnt);
  int64_t* rp = nullptr;
  MXG_TRY(pool.alloc(&rp, sizeof(int64_t) * (nrows + 1)));
  MXG_TRY(exclusive_scan(c, cnt, nrows, (long long*)rp));
  long long nnz = 0;
  MXG_HIP(hipMemcpy(&nnz, rp + nrows, sizeof(long long), hipMemcpyDeviceToHost));
  int32_t* ci = nullptr; double* va = nullptr;
  MXG_TRY(pool.alloc(&ci, sizeof(int32_t) * (nnz + 8)));
  MXG_TRY(pool.alloc(&va, sizeof(double) * (nnz + 8) * w));
  MXG_HIP(hipMemsetAsync(ci, 0, sizeof(int32_t) * (nnz + 8), c->stream));
  MXG_HIP(hipMemsetAsync(va, 0, sizeof(double) * (nnz + 8) * w, c->stream));
  if (cx) {
    Val<2> za, zb; za.x = sa[0]; za.y = sa[1]; zb.x = sb[0]; zb.y = sb[1];
    hipLaunchKernelGGL(k_add_fill<2>, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)a->d_rowptr,
                       (const int*)a->d_colidx, (const double*)a->d_vals, za, (const long long*)b->d_rowptr,
                       (const int*)b->d_colidx, (const double*)b->d_vals, zb, nrows, (const long long*)rp, ci, va);
  } else {
    Val<1> ra, rb; ra.x = sa[0]; rb.x = sb[0];
    hipLaunchKernelGGL(k_add_fill<1>, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)a->d_rowptr,
                       (const int*)a->d_colidx, (const double*)a->d_vals, ra, (const long long*)b->d_rowptr,
                       (const int*)b->d_colidx, (const double*)b->d_vals, rb, nrows, (const long long*)rp, ci, va);
  }
  MXG_HIP(hipGetLastError());
  MXG_HIP(hipStreamSynchronize(c->stream));
  pool.keep(rp); pool.keep(ci); pool.keep(va);
  return mxg_csr_adopt_device(c, nrows, a->ncols_owned, nnz, rp, ci, va, cx, out);
}

extern "C" int mxg_csr_add(mxg_ctx* c, const mxg_csr* a, double sa, const mxg_csr* b, double sb, mxg_csr** out) {
  const double za[2] = {sa, 0.0}, zb[2] = {sb, 0.0};
  return mxg_csr_add_z(c, a, za, b, zb, out);
}

extern "C" int mxg_csr_purge(mxg_ctx* c, const mxg_csr* a, double tol, mxg_csr** out) {
  DevPool pool;
  MxgSetupTrace trace("mxg_csr_purge (nnz)", a->nnz);
  MXG_TRY(check_operands(c, a, a, "mxg_csr_purge"));
  MXG_CHECK(out, "mxg_csr_purge: NULL argument");
  MXG_HIP(hipSetDevice(c->device));
  const long long nrows = a->nrows;
  long long* cnt = nullptr;
  MXG_TRY(pool.alloc(&cnt, sizeof(long long) * std::max<long long>(nrows, 1)));
  const int cx = a->is_complex ? 1 : 0, w = cx ? 2 : 1;
  if (cx) hipLaunchKernelGGL(k_purge_count<2>, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)a->d_rowptr,
                             (const double*)a->d_vals, tol, nrows, cnt);
  else hipLaunchKernelGGL(k_purge_count<1>, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)a->d_rowptr,
                          (const double*)a->d_vals, tol, nrows, cnt);
  int64_t* rp = nullptr;
  MXG_TRY(pool.alloc(&rp, sizeof(int64_t) * (nrows + 1)));
  MXG_TRY(exclusive_scan(c, cnt, nrows, (long long*)rp));
  long long nnz = 0;
  MXG_HIP(hipMemcpy(&nnz, rp + nrows, sizeof(long long), hipMemcpyDeviceToHost));
  int32_t* ci = nullptr; double* va = nullptr;
  MXG_TRY(pool.alloc(&ci, sizeof(int32_t) * (nnz + 8)));
  MXG_TRY(pool.alloc(&va, sizeof(double) * (nnz + 8) * w));
  MXG_HIP(hipMemsetAsync(ci, 0, sizeof(int32_t) * (nnz + 8), c->stream));
  MXG_HIP(hipMemsetAsync(va, 0, sizeof(double) * (nnz + 8) * w, c->stream));
  if (cx) hipLaunchKernelGGL(k_purge_fill<2>, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)a->d_rowptr,
                             (const int*)a->d_colidx, (const double*)a->d_vals, tol, nrows, (const long long*)rp, ci, va);
  else hipLaunchKernelGGL(k_purge_fill<1>, dim3(rows_grid(nrows)), dim3(SA_BLOCK), 0, c->stream, (const long long*)a->d_rowptr,
                          (const int*)a->d_colidx, (const double*)a->d_vals, tol, nrows, (const long long*)rp, ci, va);
  MXG_HIP(hipGetLastError());
  MXG_HIP(hipStreamSynchronize(c->stream));
  pool.keep(rp); pool.keep(ci); pool.keep(va);
  return mxg_csr_adopt_device(c, nrows, a->ncols_owned, nnz, rp, ci, va, cx, out);
}

// host copies of the CSR arrays (rowptr nrows + 1, colidx / vals nnz entries; NULL = not wanted)
extern "C" int mxg_csr_download(const mxg_csr* a, int64_t* rowptr, int32_t* colidx, double* vals) {
  MxgSetupTrace trace("mxg_csr_download (nnz)", a->nnz);
  MXG_CHECK(a, "mxg_csr_download: NULL argument");
  mxg_ctx* c = a->ctx;
  MXG_HIP(hipSetDevice(c->device));
  const int w = a->is_complex ? 2 : 1;
  if (rowptr) MXG_HIP(hipMemcpyAsync(rowptr, a->d_rowptr, sizeof(int64_t) * (a->nrows + 1), hipMemcpyDeviceToHost, c->stream));
  if (colidx && a->nnz > 0) MXG_HIP(hipMemcpyAsync(colidx, a->d_colidx, sizeof(int32_t) * a->nnz, hipMemcpyDeviceToHost, c->stream));
  if (vals && a->nnz > 0) MXG_HIP(hipMemcpyAsync(vals, a->d_vals, sizeof(double) * a->nnz * w, hipMemcpyDeviceToHost, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  return 0;
}
