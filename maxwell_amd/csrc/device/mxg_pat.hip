// Pattern-dictionary SpMV ("pat"): the product kernel of the eigen-solver's inner loops.
//
// The square operators MxMagWaveOp iterates on (shifted vector Laplacian, scalar Laplacian, the multigrid levels)
// are stencils: almost every row repeats one of a few hundred sequences of (column - row, value) - the bulk 7-point
// stencil with the offsets of its z-line - and only rows next to the cavity wall are individual.  mxg_csr_compress
// therefore also builds
//   dict[npat]      the distinct sequences of up to 8 entries (exact value bit patterns), 128 B each,
//   rowpid[nrows]   uint16 id per row (0xFFFF: the row is individual),
//   grows[ngrows]   the individual rows, which keep reading the CSR arrays,
// and a product is two launches with no LDS, no barrier and no staging:
//   k_spmv_pat   one thread per row: 2 B of row id, a dictionary entry that is the same address for (almost) every
//                lane of a wave (L1 / L2 resident: the whole dictionary is a few hundred KiB), the gathers, the
//                products in storage order;
//   k_spmv_gen   one thread per individual row over the CSR arrays.
// Per row the kernel moves 2 + 8 (x) + 8 (y) bytes instead of 3 bytes per ENTRY (k_spmv_cs) or 12 (CSR).  Values,
// columns and the order of the row sum are those of the CSR arrays: results are bit-identical to the serial row loop.
// The multigrid epilogues (EPI, see mxg_csr.hip) and the fused p.Ap of CG (DOT) ride along as in k_spmv_cs.
#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>

#include "mxg_internal.h"

namespace {

constexpr int PAT_LEN = 8;
struct PatEntry {                 // 128 bytes
  int delta[PAT_LEN];             // column - row
  double val[PAT_LEN];
  int len, pad[7];
};
static_assert(sizeof(PatEntry) == 128, "dictionary entry layout");
constexpr unsigned short PAT_NONE = 0xFFFF;
constexpr int PAT_BLOCK = 256;

typedef unsigned int uint4p __attribute__((ext_vector_type(4)));
typedef double dbl2p __attribute__((ext_vector_type(2)));

// the epilogues of mxg_csr.hip's cs_epilogue, restated (same arithmetic, same order)
template <int EPI>
__device__ inline double pat_epilogue(double ax, long long row, const double* __restrict__ x,
                                      const double* __restrict__ rhs, const double* __restrict__ diag, double omega,
                                      double beta, const double* y) {
  if (EPI == 1) {
    const double dg = diag[row], xr = x[row];
    return dg != 0.0 ? xr + omega * (rhs[row] - ax) / dg : xr;
  }
  if (EPI == 5) {
    const double dg = diag[row], xr = x[row];
    return dg != 0.0 ? xr + beta * xr + omega * (rhs[row] - ax) / dg : xr;
  }
  if (EPI == 4) {
    const double dg = diag[row], xr = x[row];
    return dg != 0.0 ? xr + beta * (xr - y[row]) + omega * (rhs[row] - ax) / dg : xr;
  }
  if (EPI == 2) return rhs[row] - ax;
  if (EPI == 3) return y[row] + omega * ax;
  return ax;
}

__device__ inline long long pat_xcd_block(long long w, long long nblocks) {
  const long long per = (nblocks + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  return (w % MXG_NUM_XCD) * per + (w / MXG_NUM_XCD);
}

// block sum of `part` in a fixed order (wave shuffles, then the four wave sums in wave order) -> partial[b]
__device__ inline void pat_block_sum(double part, double* __restrict__ partial, long long b) {
  __shared__ double sred[PAT_BLOCK / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < PAT_BLOCK / 64; ++w) tot += sred[w];
    partial[b] = tot;
  }
}

// HALO: columns >= nowned are ghosts and live in xg[] (the receive buffer of the exchange)
template <bool DOT, int EPI, bool HALO>
__global__ __launch_bounds__(PAT_BLOCK) void k_spmv_pat(
    const unsigned short* __restrict__ rowpid, const PatEntry* __restrict__ dict, long long nrows, long long nblocks,
    const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ xdot, double* __restrict__ partial,
    const double* __restrict__ rhs, const double* __restrict__ diag, double omega, double beta,
    const double* __restrict__ xg, long long nowned) {
  const long long b = pat_xcd_block(blockIdx.x, nblocks);
  if (b >= nblocks) return;
  const long long row = b * PAT_BLOCK + threadIdx.x;
  double part = 0.0;
  if (row < nrows) {
    const unsigned pid = rowpid[row];
    if (pid != PAT_NONE) {
      const PatEntry* e = dict + pid;
      const uint4p d0 = *reinterpret_cast<const uint4p*>(e->delta), d1 = *reinterpret_cast<const uint4p*>(e->delta + 4);
      const int len = e->len;
      const int dl[PAT_LEN] = {(int)d0.x, (int)d0.y, (int)d0.z, (int)d0.w, (int)d1.x, (int)d1.y, (int)d1.z, (int)d1.w};
      double xv[PAT_LEN];
#pragma unroll
      for (int i = 0; i < PAT_LEN; ++i) {                   // every gather in flight before the first product
        const long long cc = row + (i < len ? dl[i] : dl[0]);   // (entries past the end re-read the first column)
        xv[i] = (HALO && cc >= nowned) ? xg[cc - nowned] : x[cc];
      }
      const dbl2p v01 = *reinterpret_cast<const dbl2p*>(e->val), v23 = *reinterpret_cast<const dbl2p*>(e->val + 2),
                  v45 = *reinterpret_cast<const dbl2p*>(e->val + 4), v67 = *reinterpret_cast<const dbl2p*>(e->val + 6);
      const double v[PAT_LEN] = {v01.x, v01.y, v23.x, v23.y, v45.x, v45.y, v67.x, v67.y};
      double sum = 0.0;
#pragma unroll
      for (int i = 0; i < PAT_LEN; ++i) {
        const double s2 = sum + v[i] * xv[i];
        sum = i < len ? s2 : sum;
      }
      __builtin_nontemporal_store(pat_epilogue<EPI>(sum, row, x, rhs, diag, omega, beta, y), y + row);
      if (DOT) part = sum * xdot[row];
    }
  }
  if (DOT) pat_block_sum(part, partial, b);
}

// the individual rows: one thread per row over the CSR arrays, entries in storage order
template <bool DOT, int EPI, bool HALO>
__global__ __launch_bounds__(PAT_BLOCK) void k_spmv_gen(
    const int* __restrict__ grows, long long ngrows, const long long* __restrict__ rowptr,
    const int* __restrict__ colidx, const double* __restrict__ vals,
    const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ xdot, double* __restrict__ partial,
    const double* __restrict__ rhs, const double* __restrict__ diag, double omega, double beta,
    const double* __restrict__ xg, long long nowned) {
  const long long i = (long long)blockIdx.x * PAT_BLOCK + threadIdx.x;
  double part = 0.0;
  if (i < ngrows) {
    const long long row = grows[i];
    const long long q0 = rowptr[row], q1 = rowptr[row + 1];
    double sum = 0.0;
    for (long long q = q0; q < q1; q += 4) {              // gathers in batches of four, products in storage order
      double vv[4], xx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long qq = q + j < q1 ? q + j : q1 - 1;
        const long long cc = colidx[qq];
        vv[j] = vals[qq];
        xx[j] = (HALO && cc >= nowned) ? xg[cc - nowned] : x[cc];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double s2 = sum + vv[j] * xx[j];
        sum = q + j < q1 ? s2 : sum;
      }
    }
    y[row] = pat_epilogue<EPI>(sum, row, x, rhs, diag, omega, beta, y);
    if (DOT) part = sum * xdot[row];
  }
  if (DOT) pat_block_sum(part, partial, blockIdx.x);
}

template <bool DOT, int EPI, bool HALO>
void launch_pat_t(mxg_ctx* c, mxg_csr* a, const double* x, double* y, const double* xdot, double* partial,
                  const double* rhs, const double* diag, double omega, double beta, const double* xg, long long nowned,
                  bool pattern_rows, bool generic_rows) {
  const long long nb = (a->nrows + PAT_BLOCK - 1) / PAT_BLOCK;
  if (pattern_rows && nb > 0) {
    const unsigned grid = (unsigned)(((nb + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD);
    hipLaunchKernelGGL((k_spmv_pat<DOT, EPI, HALO>), dim3(grid), dim3(PAT_BLOCK), 0, c->stream,
                       (const unsigned short*)a->d_rowpid, (const PatEntry*)a->d_patdict, (long long)a->nrows, nb, x, y,
                       xdot, partial, rhs, diag, omega, beta, xg, nowned);
  }
  const long long ng = a->pat_ngrows;
  if (generic_rows && ng > 0) {
    const unsigned grid = (unsigned)((ng + PAT_BLOCK - 1) / PAT_BLOCK);
    hipLaunchKernelGGL((k_spmv_gen<DOT, EPI, HALO>), dim3(grid), dim3(PAT_BLOCK), 0, c->stream,
                       (const int*)a->d_grows, ng, (const long long*)a->d_rowptr, (const int*)a->d_colidx,
                       (const double*)a->d_vals, x, y, xdot, DOT ? partial + nb : partial, rhs, diag, omega, beta, xg, nowned);
  }
}

}  // namespace

int64_t mxg_pat_partials(const mxg_csr* a) {
  return (a->nrows + PAT_BLOCK - 1) / PAT_BLOCK + (a->pat_ngrows + PAT_BLOCK - 1) / PAT_BLOCK;
}

// y = epilogue(A x) over the pattern dictionary.  halo: 0 none; 1 both launches read ghosts from xg (the caller has
// made the stream wait for the exchange).  dot: per-block partial sums of (A x) . xdot go to partial[0 .. mxg_pat_partials).
int mxg_pat_launch(mxg_ctx* c, mxg_csr* a, const double* x, double* y, int epi, int dot, const double* xdot,
                   double* partial, const double* rhs, const double* diag, double omega, double beta, int halo,
                   const double* xg) {
  const long long nowned = a->ncols_owned;
#define MXG_PAT(D, E)                                                                                        \
  do {                                                                                                       \
    if (halo) launch_pat_t<D, E, true>(c, a, x, y, xdot, partial, rhs, diag, omega, beta, xg, nowned, true, true);  \
    else launch_pat_t<D, E, false>(c, a, x, y, xdot, partial, rhs, diag, omega, beta, xg, nowned, true, true);      \
  } while (0)
  if (dot) { MXG_PAT(true, 0); }
  else if (epi == 1) { MXG_PAT(false, 1); }
  else if (epi == 2) { MXG_PAT(false, 2); }
  else if (epi == 3) { MXG_PAT(false, 3); }
  else if (epi == 4) { MXG_PAT(false, 4); }
  else if (epi == 5) { MXG_PAT(false, 5); }
  else { MXG_PAT(false, 0); }
#undef MXG_PAT
  if (hipGetLastError() != hipSuccess) { mxg_set_error("mxg_pat_launch: launch failed"); return 1; }
  return 0;
}

// Build the dictionary, the per-row ids and the list of individual rows from host CSR arrays.
int mxg_pat_build(mxg_csr* a, const int64_t* rp, const int32_t* ci, const double* va) {
  mxg_ctx* c = a->ctx;
  const int64_t n = a->nrows;
  std::vector<unsigned short> rowpid((size_t)std::max<int64_t>(n, 1), PAT_NONE);
  std::vector<PatEntry> dict;
  std::unordered_map<std::string, int> global;
  std::mutex mu;
  auto work = [&](int64_t r0, int64_t r1) {
    std::unordered_map<std::string, int> local;           // signatures this thread has already resolved
    std::string sg;
    for (int64_t r = r0; r < r1; ++r) {
      const int64_t q0 = rp[r], q1 = rp[r + 1];
      const int len = (int)(q1 - q0);
      if (len == 0 || len > PAT_LEN) continue;
      sg.assign((size_t)len * 12, '\0');
      for (int i = 0; i < len; ++i) {
        const int32_t dl = (int32_t)((int64_t)ci[q0 + i] - r);
        std::memcpy(&sg[(size_t)i * 12], &dl, 4);
        std::memcpy(&sg[(size_t)i * 12 + 4], &va[q0 + i], 8);
      }
      auto it = local.find(sg);
      int id;
      if (it != local.end()) {
        id = it->second;
      } else {
        std::lock_guard<std::mutex> lk(mu);
        auto g = global.find(sg);
        if (g != global.end()) {
          id = g->second;
        } else if (dict.size() < (size_t)PAT_NONE) {
          PatEntry e{};
          e.len = len;
          for (int i = 0; i < len; ++i) {
            std::memcpy(&e.delta[i], &sg[(size_t)i * 12], 4);
            std::memcpy(&e.val[i], &sg[(size_t)i * 12 + 4], 8);
          }
          id = (int)dict.size();
          dict.push_back(e);
          global.emplace(sg, id);
        } else {
          id = -1;                                          // dictionary full: the row stays individual
        }
        local.emplace(sg, id);
      }
      if (id >= 0) rowpid[(size_t)r] = (unsigned short)id;
    }
  };
  unsigned nthreads = std::thread::hardware_concurrency();
  if (const char* e = getenv("OMP_NUM_THREADS")) { const int v = atoi(e); if (v > 0) nthreads = (unsigned)v; }
  if (nthreads > 32) nthreads = 32;
  if (nthreads < 1 || n < 4096) nthreads = 1;
  std::vector<std::thread> pool;
  for (unsigned i = 0; i < nthreads; ++i) pool.emplace_back(work, n * i / nthreads, n * (i + 1) / nthreads);
  for (auto& th : pool) th.join();
  std::vector<int32_t> grows;
  for (int64_t r = 0; r < n; ++r)
    if (rowpid[(size_t)r] == PAT_NONE) grows.push_back((int32_t)r);
  if (a->d_rowpid) hipFree(a->d_rowpid);
  if (a->d_patdict) hipFree(a->d_patdict);
  if (a->d_grows) hipFree(a->d_grows);
  a->d_rowpid = nullptr; a->d_patdict = nullptr; a->d_grows = nullptr;
  MXG_HIP(hipMalloc(&a->d_rowpid, sizeof(unsigned short) * rowpid.size()));
  MXG_HIP(hipMalloc(&a->d_patdict, sizeof(PatEntry) * std::max<size_t>(dict.size(), 1)));
  MXG_HIP(hipMalloc(&a->d_grows, sizeof(int32_t) * std::max<size_t>(grows.size(), 1)));
  MXG_HIP(hipMemcpyAsync(a->d_rowpid, rowpid.data(), sizeof(unsigned short) * rowpid.size(), hipMemcpyHostToDevice, c->stream));
  if (!dict.empty())
    MXG_HIP(hipMemcpyAsync(a->d_patdict, dict.data(), sizeof(PatEntry) * dict.size(), hipMemcpyHostToDevice, c->stream));
  if (!grows.empty())
    MXG_HIP(hipMemcpyAsync(a->d_grows, grows.data(), sizeof(int32_t) * grows.size(), hipMemcpyHostToDevice, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  a->pat_npat = (int64_t)dict.size();
  a->pat_ngrows = (int64_t)grows.size();
  int64_t gnnz = 0;
  for (int32_t r : grows) gnnz += rp[r + 1] - rp[r];
  a->pat_gnnz = gnnz;
  return 0;
}
