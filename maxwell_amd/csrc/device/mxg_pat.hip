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

// ---------------------------------------------------------------------------------------------------------------
// Wave-persistent form of the same product ("patw").  scripts/ubench_gather.hip showed that a bare 7-point stencil
// over this numbering runs in 0.077 ms for 20.6 M rows (0.054 ms with 16-byte gathers) - the x gathers are not what
// costs 0.2 ms in k_spmv_cs / k_spmv_pat; the per-row vector-memory and LDS instructions that fetch and decode the
// matrix are (k_spmv_pat: 7 dictionary loads + 8 gathers per row).  On the Yee lattice a row's pattern is that of the
// row three before it (same component, next cell of the grid line) for hundreds of rows, so: each wavefront owns a
// contiguous range of rows and walks it in tiles of 63 rows (a multiple of the three interleaved components, lane l
// always meets the same component); a lane keeps its pattern - column offsets and values - in registers and reloads
// it only when the 2-byte id of its next row differs from the cached one (line ends, the cavity wall).  Per row that
// leaves the id, the gathers (as many as the longest row of the tile has entries) and the store.
// Values, offsets and the order of the row sum are those of the dictionary = of the CSR arrays: bit-identical.
constexpr int PATW_TILE = 63;
constexpr int PATW_WAVES = 4;          // wavefronts per workgroup

__device__ inline long long patw_rows_per_wave(long long nrows, long long nwaves) {
  const long long tiles = (nrows + PATW_TILE - 1) / PATW_TILE;
  return ((tiles + nwaves - 1) / nwaves) * PATW_TILE;
}

template <bool DOT, int EPI, bool HALO>
__global__ __launch_bounds__(64 * PATW_WAVES) void k_spmv_patw(
    const unsigned short* __restrict__ rowpid, const PatEntry* __restrict__ dict, long long nrows, long long nwgs,
    const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ xdot, double* __restrict__ partial,
    const double* __restrict__ rhs, const double* __restrict__ diag, double omega, double beta,
    const double* __restrict__ xg, long long nowned) {
  const long long wg = pat_xcd_block(blockIdx.x, nwgs);          // XCD w % 8 walks a contiguous eighth of the rows
  if (wg >= nwgs) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gw = wg * PATW_WAVES + wave;
  const long long per = patw_rows_per_wave(nrows, nwgs * PATW_WAVES);
  const long long r_begin = gw * per, r_end = min(nrows, r_begin + per);
  unsigned cpid = PAT_NONE;                                      // id of the pattern held in dl / v / len
  int dl[PAT_LEN];
  double v[PAT_LEN];
  int len = 0;
#pragma unroll
  for (int i = 0; i < PAT_LEN; ++i) { dl[i] = 0; v[i] = 0.0; }
  double part = 0.0;
  const bool live = lane < PATW_TILE;
  long long row = r_begin + lane;
  unsigned pid = (live && row < r_end) ? (unsigned)rowpid[row] : (unsigned)PAT_NONE;
  for (long long base = r_begin; base < r_end; base += PATW_TILE) {
    row = base + lane;
    const bool act = live && row < r_end && pid != PAT_NONE;
    // the id of this lane's row in the next tile: in flight while this tile is processed
    const long long nrow = row + PATW_TILE;
    const unsigned npid = (live && nrow < r_end) ? (unsigned)rowpid[nrow] : (unsigned)PAT_NONE;
    if (act && pid != cpid) {                                    // exec-masked reload: rare
      const PatEntry* e = dict + pid;
      const uint4p d0 = *reinterpret_cast<const uint4p*>(e->delta), d1 = *reinterpret_cast<const uint4p*>(e->delta + 4);
      const dbl2p v01 = *reinterpret_cast<const dbl2p*>(e->val), v23 = *reinterpret_cast<const dbl2p*>(e->val + 2),
                  v45 = *reinterpret_cast<const dbl2p*>(e->val + 4), v67 = *reinterpret_cast<const dbl2p*>(e->val + 6);
      len = e->len;
      dl[0] = (int)d0.x; dl[1] = (int)d0.y; dl[2] = (int)d0.z; dl[3] = (int)d0.w;
      dl[4] = (int)d1.x; dl[5] = (int)d1.y; dl[6] = (int)d1.z; dl[7] = (int)d1.w;
      v[0] = v01.x; v[1] = v01.y; v[2] = v23.x; v[3] = v23.y; v[4] = v45.x; v[5] = v45.y; v[6] = v67.x; v[7] = v67.y;
      cpid = pid;
    }
    const int mylen = act ? len : 0;
    int maxlen = mylen;                                          // wave-uniform trip count of the gather loop
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
    if (maxlen > 0) {
      double xv[PAT_LEN];
#pragma unroll
      for (int i = 0; i < PAT_LEN; ++i) {
        if (i < maxlen) {                                        // uniform branch; lanes past their end re-read entry 0
          const long long cc = act ? row + (i < mylen ? dl[i] : dl[0]) : 0;
          xv[i] = (HALO && cc >= nowned) ? xg[cc - nowned] : x[cc];
        }
      }
      if (act) {
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < PAT_LEN; ++i) {
          if (i < maxlen) {
            const double s2 = sum + v[i] * xv[i];
            sum = i < mylen ? s2 : sum;
          }
        }
        __builtin_nontemporal_store(pat_epilogue<EPI>(sum, row, x, rhs, diag, omega, beta, y), y + row);
        if (DOT) part += sum * xdot[row];
      }
    }
    pid = npid;
  }
  if (DOT) {                                                     // one partial per wavefront, lanes added in lane order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if (lane == 0) partial[gw] = part;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Row-pair form ("pair", MXG_PAT=2).  scripts/ubench_gather.hip: the floor of these kernels is the number of vector-
// memory instructions per row (a wave64 load or store costs a CU ~18 cycles whatever it hits), and a bare stencil
// with 16-byte gathers runs at the HBM rate of x + y.  Two consecutive rows (2m, 2m+1: different components of
// the same or the next cell) have the SAME column offsets wherever the lattice is regular, so one 16-byte gather
// serves both: the dictionary holds pair patterns (shared offsets, two value lists), a lane handles PAIR_U pairs at
// stride 63 pairs (a multiple of the three interleaved components: the same pattern for hundreds of rows along a
// grid line), loads the pattern once and issues all PAIR_U x len gathers before the first product.  Pairs whose id
// differs from the lane's first one take a second, masked pass; rows that do not pair stay individual (k_spmv_gen).
// Per row: (PAIR_U + 10 + PAIR_U len + PAIR_U) / (2 PAIR_U) = 5.75 vector-memory instructions at len 7 instead of
// 17 (k_spmv_pat).  Values, offsets and the order of each row sum are those of the CSR arrays: bit-identical.
struct PairEntry {                // 256 bytes
  int delta[PAT_LEN];             // column - row, the same for both rows of the pair
  double va[PAT_LEN];             // values of row 2m
  double vb[PAT_LEN];             // values of row 2m + 1
  int len, pad[23];
};
static_assert(sizeof(PairEntry) == 256, "pair dictionary entry layout");
constexpr int PAIR_TILE = 63, PAIR_WAVES = 4;
typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));   // 16-byte access at 8-byte alignment

struct PairRegs { int dl[PAT_LEN]; double va[PAT_LEN], vb[PAT_LEN]; int len; };

__device__ inline void pair_load(const PairEntry* __restrict__ e, PairRegs& R) {
  const uint4p d0 = *reinterpret_cast<const uint4p*>(e->delta), d1 = *reinterpret_cast<const uint4p*>(e->delta + 4);
  R.dl[0] = (int)d0.x; R.dl[1] = (int)d0.y; R.dl[2] = (int)d0.z; R.dl[3] = (int)d0.w;
  R.dl[4] = (int)d1.x; R.dl[5] = (int)d1.y; R.dl[6] = (int)d1.z; R.dl[7] = (int)d1.w;
#pragma unroll
  for (int i = 0; i < PAT_LEN; i += 2) {
    const dbl2p a = *reinterpret_cast<const dbl2p*>(e->va + i), b = *reinterpret_cast<const dbl2p*>(e->vb + i);
    R.va[i] = a.x; R.va[i + 1] = a.y; R.vb[i] = b.x; R.vb[i + 1] = b.y;
  }
  R.len = e->len;
}

// epilogue of both rows of a pair (row = 2m, 16-byte aligned accesses); same arithmetic as pat_epilogue
template <int EPI>
__device__ inline dbl2p pair_epilogue(dbl2p ax, long long row, const double* __restrict__ x, const double* __restrict__ rhs,
                                      const double* __restrict__ diag, double omega, double beta, const double* y) {
  dbl2p o;
  if (EPI == 0) return ax;
  if (EPI == 2) { const dbl2p r = *reinterpret_cast<const dbl2p*>(rhs + row); o.x = r.x - ax.x; o.y = r.y - ax.y; return o; }
  if (EPI == 3) { const dbl2p yy = *reinterpret_cast<const dbl2p*>(y + row); o.x = yy.x + omega * ax.x; o.y = yy.y + omega * ax.y; return o; }
  const dbl2p dg = *reinterpret_cast<const dbl2p*>(diag + row), xr = *reinterpret_cast<const dbl2p*>(x + row),
              r = *reinterpret_cast<const dbl2p*>(rhs + row);
  if (EPI == 1) {
    o.x = dg.x != 0.0 ? xr.x + omega * (r.x - ax.x) / dg.x : xr.x;
    o.y = dg.y != 0.0 ? xr.y + omega * (r.y - ax.y) / dg.y : xr.y;
  } else if (EPI == 5) {
    o.x = dg.x != 0.0 ? xr.x + beta * xr.x + omega * (r.x - ax.x) / dg.x : xr.x;
    o.y = dg.y != 0.0 ? xr.y + beta * xr.y + omega * (r.y - ax.y) / dg.y : xr.y;
  } else {   // EPI == 4
    const dbl2p yy = *reinterpret_cast<const dbl2p*>(y + row);
    o.x = dg.x != 0.0 ? xr.x + beta * (xr.x - yy.x) + omega * (r.x - ax.x) / dg.x : xr.x;
    o.y = dg.y != 0.0 ? xr.y + beta * (xr.y - yy.y) + omega * (r.y - ax.y) / dg.y : xr.y;
  }
  return o;
}

template <bool DOT, int EPI, int PAIR_U>
__global__ __launch_bounds__(64 * PAIR_WAVES) void k_spmv_pair(
    const unsigned short* __restrict__ pairid, const PairEntry* __restrict__ dict, long long npairs, long long nwgs,
    const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ xdot, double* __restrict__ partial,
    const double* __restrict__ rhs, const double* __restrict__ diag, double omega, double beta) {
  const long long wg = pat_xcd_block(blockIdx.x, nwgs);          // XCD w % 8 walks a contiguous eighth of the rows
  if (wg >= nwgs) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gw = wg * PAIR_WAVES + wave;
  const long long m0 = gw * (long long)(PAIR_TILE * PAIR_U);
  const bool live = lane < PAIR_TILE;
  long long m[PAIR_U];
  unsigned pid[PAIR_U];
#pragma unroll
  for (int u = 0; u < PAIR_U; ++u) {
    m[u] = m0 + (long long)u * PAIR_TILE + lane;
    pid[u] = (live && m[u] < npairs) ? (unsigned)pairid[m[u]] : (unsigned)PAT_NONE;
  }
  double part = 0.0;
  PairRegs R;
  R.len = 0;
  // ---- pass 1: every pair of this lane that carries the id of its first one
  const unsigned cur = pid[0];
  if (cur != PAT_NONE) pair_load(dict + cur, R);
  const int mylen = cur != PAT_NONE ? R.len : 0;
  int maxlen = mylen;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
  bool fast[PAIR_U];
#pragma unroll
  for (int u = 0; u < PAIR_U; ++u) fast[u] = cur != PAT_NONE && pid[u] == cur;
  {
    dbl2p xv[PAIR_U][PAT_LEN];
#pragma unroll
    for (int u = 0; u < PAIR_U; ++u)
#pragma unroll
      for (int i = 0; i < PAT_LEN; ++i)
        if (i < maxlen) {                                        // wave-uniform; lanes past their end re-read entry 0
          const long long cc = (fast[u] && mylen > 0) ? 2 * m[u] + (i < mylen ? R.dl[i] : R.dl[0]) : 0;   // empty pair: x[0]
          const dbl2u t = *reinterpret_cast<const dbl2u*>(x + cc);
          xv[u][i].x = t.x; xv[u][i].y = t.y;
        }
#pragma unroll
    for (int u = 0; u < PAIR_U; ++u)
      if (fast[u]) {
        dbl2p sum; sum.x = 0.0; sum.y = 0.0;
#pragma unroll
        for (int i = 0; i < PAT_LEN; ++i)
          if (i < maxlen) {
            const double sa = sum.x + R.va[i] * xv[u][i].x, sb = sum.y + R.vb[i] * xv[u][i].y;
            sum.x = i < mylen ? sa : sum.x;
            sum.y = i < mylen ? sb : sum.y;
          }
        const long long row = 2 * m[u];
        __builtin_nontemporal_store(pair_epilogue<EPI>(sum, row, x, rhs, diag, omega, beta, y), reinterpret_cast<dbl2p*>(y + row));
        if (DOT) { const dbl2p xd = *reinterpret_cast<const dbl2p*>(xdot + row); part += sum.x * xd.x; part += sum.y * xd.y; }
      }
  }
  // ---- pass 2 (line ends, the cavity wall): pairs with another id, one at a time under the lane mask
#pragma unroll
  for (int u = 1; u < PAIR_U; ++u) {
    const bool need = pid[u] != PAT_NONE && !fast[u];
    if (__any(need)) {
      if (need) {
        pair_load(dict + pid[u], R);
        dbl2p xv[PAT_LEN];
#pragma unroll
        for (int i = 0; i < PAT_LEN; ++i) {
          const long long cc = R.len > 0 ? 2 * m[u] + (i < R.len ? R.dl[i] : R.dl[0]) : 0;
          const dbl2u t = *reinterpret_cast<const dbl2u*>(x + cc);
          xv[i].x = t.x; xv[i].y = t.y;
        }
        dbl2p sum; sum.x = 0.0; sum.y = 0.0;
#pragma unroll
        for (int i = 0; i < PAT_LEN; ++i) {
          const double sa = sum.x + R.va[i] * xv[i].x, sb = sum.y + R.vb[i] * xv[i].y;
          sum.x = i < R.len ? sa : sum.x;
          sum.y = i < R.len ? sb : sum.y;
        }
        const long long row = 2 * m[u];
        __builtin_nontemporal_store(pair_epilogue<EPI>(sum, row, x, rhs, diag, omega, beta, y), reinterpret_cast<dbl2p*>(y + row));
        if (DOT) { const dbl2p xd = *reinterpret_cast<const dbl2p*>(xdot + row); part += sum.x * xd.x; part += sum.y * xd.y; }
      }
    }
  }
  if (DOT) {                                                     // one partial per wavefront, lanes added in lane order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if (lane == 0) partial[gw] = part;
  }
}

// Wave-persistent form of the pair product (MXG_PAIRW=1; the default is k_spmv_pair): a wavefront owns a contiguous
// range of pairs and walks it PAIRW_U tiles of 63 pairs at a time; a lane keeps its pair pattern in registers and
// reloads it only when an id differs from the cached one; the ids of the next step are fetched while the gathers of
// this one are in flight, so a step is ONE memory round trip (the gathers) instead of three (id -> pattern -> gathers).
constexpr int PAIRW_U = 2;

__device__ inline long long pairw_pairs_per_wave(long long npairs, long long nwaves) {
  const long long step = (long long)PAIR_TILE * PAIRW_U;
  const long long steps = (npairs + step - 1) / step;
  return ((steps + nwaves - 1) / nwaves) * step;
}

template <bool DOT, int EPI>
__global__ __launch_bounds__(64 * PAIR_WAVES) void k_spmv_pairw(
    const unsigned short* __restrict__ pairid, const PairEntry* __restrict__ dict, long long npairs, long long nwgs,
    const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ xdot, double* __restrict__ partial,
    const double* __restrict__ rhs, const double* __restrict__ diag, double omega, double beta) {
  const long long wg = pat_xcd_block(blockIdx.x, nwgs);
  if (wg >= nwgs) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gw = wg * PAIR_WAVES + wave;
  const long long per = pairw_pairs_per_wave(npairs, nwgs * PAIR_WAVES);
  const long long m_begin = gw * per, m_end = min(npairs, m_begin + per);
  const bool live = lane < PAIR_TILE;
  PairRegs R;
  R.len = 0;
#pragma unroll
  for (int i = 0; i < PAT_LEN; ++i) { R.dl[i] = 0; R.va[i] = 0.0; R.vb[i] = 0.0; }
  unsigned cpid = PAT_NONE;
  double part = 0.0;
  unsigned pid[PAIRW_U];
#pragma unroll
  for (int u = 0; u < PAIRW_U; ++u) {
    const long long mm = m_begin + (long long)u * PAIR_TILE + lane;
    pid[u] = (live && mm < m_end) ? (unsigned)pairid[mm] : (unsigned)PAT_NONE;
  }
  for (long long base = m_begin; base < m_end; base += (long long)PAIR_TILE * PAIRW_U) {
    long long m[PAIRW_U];
    unsigned npid[PAIRW_U];
#pragma unroll
    for (int u = 0; u < PAIRW_U; ++u) {
      m[u] = base + (long long)u * PAIR_TILE + lane;
      const long long nm = m[u] + (long long)PAIR_TILE * PAIRW_U;
      npid[u] = (live && nm < m_end) ? (unsigned)pairid[nm] : (unsigned)PAT_NONE;   // in flight during this step
    }
    // the lane's pattern: that of its first live pair of this step; reload only on a change
    unsigned want = PAT_NONE;
#pragma unroll
    for (int u = PAIRW_U - 1; u >= 0; --u) want = pid[u] != PAT_NONE ? pid[u] : want;
    if (want != PAT_NONE && want != cpid) { pair_load(dict + want, R); cpid = want; }
    bool fast[PAIRW_U];
    bool anyfast = false;
#pragma unroll
    for (int u = 0; u < PAIRW_U; ++u) { fast[u] = pid[u] != PAT_NONE && pid[u] == cpid; anyfast = anyfast || fast[u]; }
    const int mylen = anyfast ? R.len : 0;
    int maxlen = mylen;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
    if (maxlen > 0) {
      dbl2p xv[PAIRW_U][PAT_LEN];
#pragma unroll
      for (int u = 0; u < PAIRW_U; ++u)
#pragma unroll
        for (int i = 0; i < PAT_LEN; ++i)
          if (i < maxlen) {
            const long long cc = (fast[u] && mylen > 0) ? 2 * m[u] + (i < mylen ? R.dl[i] : R.dl[0]) : 0;   // empty pair: x[0]
            const dbl2u t = *reinterpret_cast<const dbl2u*>(x + cc);
            xv[u][i].x = t.x; xv[u][i].y = t.y;
          }
#pragma unroll
      for (int u = 0; u < PAIRW_U; ++u)
        if (fast[u]) {
          dbl2p sum; sum.x = 0.0; sum.y = 0.0;
#pragma unroll
          for (int i = 0; i < PAT_LEN; ++i)
            if (i < maxlen) {
              const double sa = sum.x + R.va[i] * xv[u][i].x, sb = sum.y + R.vb[i] * xv[u][i].y;
              sum.x = i < mylen ? sa : sum.x;
              sum.y = i < mylen ? sb : sum.y;
            }
          const long long row = 2 * m[u];
          __builtin_nontemporal_store(pair_epilogue<EPI>(sum, row, x, rhs, diag, omega, beta, y), reinterpret_cast<dbl2p*>(y + row));
          if (DOT) { const dbl2p xd = *reinterpret_cast<const dbl2p*>(xdot + row); part += sum.x * xd.x; part += sum.y * xd.y; }
        }
    } else if (anyfast) {                                      // empty pairs: the epilogue of a zero product
#pragma unroll
      for (int u = 0; u < PAIRW_U; ++u)
        if (fast[u]) {
          dbl2p sum; sum.x = 0.0; sum.y = 0.0;
          const long long row = 2 * m[u];
          __builtin_nontemporal_store(pair_epilogue<EPI>(sum, row, x, rhs, diag, omega, beta, y), reinterpret_cast<dbl2p*>(y + row));
        }
    }
    // pairs of this step with another id than the cached one (line ends, the wall): one at a time under the lane mask
#pragma unroll
    for (int u = 1; u < PAIRW_U; ++u) {
      const bool need = pid[u] != PAT_NONE && !fast[u];
      if (__any(need)) {
        if (need) {
          pair_load(dict + pid[u], R);
          cpid = pid[u];
          dbl2p xw[PAT_LEN];
#pragma unroll
          for (int i = 0; i < PAT_LEN; ++i) {
            const long long cc = R.len > 0 ? 2 * m[u] + (i < R.len ? R.dl[i] : R.dl[0]) : 0;
            const dbl2u t = *reinterpret_cast<const dbl2u*>(x + cc);
            xw[i].x = t.x; xw[i].y = t.y;
          }
          dbl2p sum; sum.x = 0.0; sum.y = 0.0;
#pragma unroll
          for (int i = 0; i < PAT_LEN; ++i) {
            const double sa = sum.x + R.va[i] * xw[i].x, sb = sum.y + R.vb[i] * xw[i].y;
            sum.x = i < R.len ? sa : sum.x;
            sum.y = i < R.len ? sb : sum.y;
          }
          const long long row = 2 * m[u];
          __builtin_nontemporal_store(pair_epilogue<EPI>(sum, row, x, rhs, diag, omega, beta, y), reinterpret_cast<dbl2p*>(y + row));
          if (DOT) { const dbl2p xd = *reinterpret_cast<const dbl2p*>(xdot + row); part += sum.x * xd.x; part += sum.y * xd.y; }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < PAIRW_U; ++u) pid[u] = npid[u];
  }
  if (DOT) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if (lane == 0) partial[gw] = part;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Tile form ("tile", MXG_PAT=3): patterns that are uniform per wavefront, fetched through SCALAR loads.
// A run is a stretch of consecutive rows that all have the same length and column offsets and in which every row
// repeats the values of the row three before it (same component, next cell of the grid line) - the interior of a grid
// line.  Runs are cut into tiles of <= 126 rows (63 lanes x 2 rows; a multiple of three, so every tile of a run has
// the same pattern): 16 bytes of descriptor {first row, rows, pattern} per tile, and a pattern = {length, offsets,
// values of row offsets 0, 1, 2 mod 3}.  A wavefront reads descriptor and pattern with s_load, so the gather addresses
// are SGPR base + lane * 16 with no vector arithmetic and no per-lane dependent load in front of them, two rows per
// lane share each 16-byte gather, and a lane picks its two value lists by (2 lane) mod 3 with v_cndmask.  This is the
// form of scripts/ubench_gather.hip's v2 kernel (0.054 ms for 20.6 M rows).  Rows outside tiles (the cavity wall, line
// ends, runs shorter than TILE_MIN_RUN) are walked over the CSR arrays by the first workgroups of the SAME launch.
struct TileDesc { int start, nrows, pid, pad; };                     // 16 bytes
struct TileEntry {                                                   // 256 bytes
  int len;
  int dslot;                                                         // index of the diagonal entry (offset 0), -1 without one
  int pad0[6];
  int delta[PAT_LEN];
  double v[3][PAT_LEN];
};
static_assert(sizeof(TileEntry) == 256, "tile dictionary entry layout");
constexpr int TILE_ROWS = 126, TILE_WAVES = 4, TILE_MIN_RUN = 24, TILE_TPW = 2;

template <bool DOT, int EPI>
__global__ __launch_bounds__(64 * TILE_WAVES) void k_spmv_tile(
    const TileDesc* __restrict__ desc, const TileEntry* __restrict__ dict, long long ntiles, long long ntb, long long ngb,
    const int* __restrict__ grows, long long ngrows, const long long* __restrict__ rowptr, const int* __restrict__ colidx,
    const double* __restrict__ vals, const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ xdot,
    double* __restrict__ partial, const double* __restrict__ rhs, const double* __restrict__ diag, double omega, double beta,
    bool skipgen, bool spread) {
  double part = 0.0;
  const long long pb = blockIdx.x;                                   // partial slot
  // The first ngb8 workgroups walk the irregular rows (measured: better than spreading them among the tile groups,
  // 0.143 against 0.173 ms - while they are resident they take wave slots from the streaming tiles either way)
  const long long ngb8 = ((ngb + 7) / 8) * 8;
  // spread: groups of 8 workgroups (one per XCD); the ngg groups of irregular rows are placed evenly among the tile groups,
  // i.e. next to the tiles of the same part of the matrix
  const long long ngg = ngb8 / 8, ntg = (ntb + 7) / 8, G = blockIdx.x / 8;
  const long long gs = (spread && ngg > 0) ? max(1LL, (ngg + ntg) / ngg) : 1;
  const long long gq = G / gs;
  const bool gen_group = spread ? (ngg > 0 && G % gs == 0 && gq < ngg) : (long long)blockIdx.x < ngb8;
  const long long gen_block = spread ? gq * 8 + blockIdx.x % 8 : (long long)blockIdx.x;
  const long long tile_block = spread ? (G - (ngg > 0 ? min(ngg, gq + 1) : 0)) * 8 + blockIdx.x % 8 : (long long)blockIdx.x - ngb8;
  if (gen_group) {
    // ---- irregular rows: one thread per row over the CSR arrays
    const long long i = gen_block * (64 * TILE_WAVES) + threadIdx.x;
    if (i < ngrows && !skipgen) {
      const long long row = grows[i];
      const long long q0 = rowptr[row], q1 = rowptr[row + 1];
      double sum = 0.0;
      for (long long q = q0; q < q1; q += 8) {
        double vv[8], xx[8];
        int cc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long long qq = q + j < q1 ? q + j : q1 - 1;
          cc[j] = colidx[qq];
          vv[j] = vals[qq];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) xx[j] = x[cc[j]];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const double s2 = sum + vv[j] * xx[j];
          sum = q + j < q1 ? s2 : sum;
        }
      }
      __builtin_nontemporal_store(pat_epilogue<EPI>(sum, row, x, rhs, diag, omega, beta, y), y + row);
      if (DOT) part = sum * xdot[row];
    }
  } else {
    const long long tb = pat_xcd_block(tile_block, ntb);   // XCD w % 8 walks a contiguous eighth of the tiles
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long long t0 = (tb * TILE_WAVES + wave) * TILE_TPW;       // TILE_TPW consecutive tiles per wavefront
    if (tb < ntb && t0 < ntiles) {
      const int ca = (2 * lane) % 3, cb = (2 * lane + 1) % 3;
      TileDesc d[TILE_TPW];
      const TileEntry* __restrict__ e[TILE_TPW];
      int len[TILE_TPW];
      bool on[TILE_TPW];
#pragma unroll
      for (int u = 0; u < TILE_TPW; ++u) {                          // wave-uniform: scalar loads
        const long long t = min(t0 + u, ntiles - 1);
        d[u] = desc[t];
        e[u] = dict + d[u].pid;
        len[u] = e[u]->len;
        on[u] = t0 + u < ntiles && 2 * lane < d[u].nrows;           // nrows is even
      }
      dbl2p xv[TILE_TPW][PAT_LEN];
#pragma unroll
      for (int u = 0; u < TILE_TPW; ++u)
        if (on[u]) {
#pragma unroll
          for (int i = 0; i < PAT_LEN; ++i)
            if (i < len[u]) {                                       // uniform
              const double* xp = x + ((long long)d[u].start + e[u]->delta[i]);   // scalar base
              const dbl2u tt = *reinterpret_cast<const dbl2u*>(xp + 2 * lane);
              xv[u][i].x = tt.x; xv[u][i].y = tt.y;
            }
        }
#pragma unroll
      for (int u = 0; u < TILE_TPW; ++u)
        if (on[u]) {
          const long long row = (long long)d[u].start + 2 * lane;
          dbl2p sum; sum.x = 0.0; sum.y = 0.0;
#pragma unroll
          for (int i = 0; i < PAT_LEN; ++i)
            if (i < len[u]) {
              const double v0 = e[u]->v[0][i], v1 = e[u]->v[1][i], v2 = e[u]->v[2][i];
              const double va = ca == 0 ? v0 : (ca == 1 ? v1 : v2), vb = cb == 0 ? v0 : (cb == 1 ? v1 : v2);
              sum.x = sum.x + va * xv[u][i].x;
              sum.y = sum.y + vb * xv[u][i].y;
            }
          dbl2p o;
          if (EPI == 0) {
            o = sum;
          } else {
            dbl2p rr, dg, xr, yy;
            if (EPI != 3) { const dbl2u q = *reinterpret_cast<const dbl2u*>(rhs + row); rr.x = q.x; rr.y = q.y; }
            if (EPI == 1 || EPI == 4 || EPI == 5) {
              const dbl2u q2 = *reinterpret_cast<const dbl2u*>(x + row);
              xr.x = q2.x; xr.y = q2.y;
              const int ds = e[u]->dslot;                             // uniform
              if (ds >= 0) {                                          // the diagonal is an entry of the pattern: no load
                const double d0 = e[u]->v[0][ds], d1 = e[u]->v[1][ds], d2 = e[u]->v[2][ds];
                dg.x = ca == 0 ? d0 : (ca == 1 ? d1 : d2);
                dg.y = cb == 0 ? d0 : (cb == 1 ? d1 : d2);
              } else {
                const dbl2u q1 = *reinterpret_cast<const dbl2u*>(diag + row);
                dg.x = q1.x; dg.y = q1.y;
              }
            }
            if (EPI == 3 || EPI == 4) { const dbl2u q = *reinterpret_cast<const dbl2u*>(y + row); yy.x = q.x; yy.y = q.y; }
            if (EPI == 2) { o.x = rr.x - sum.x; o.y = rr.y - sum.y; }
            else if (EPI == 3) { o.x = yy.x + omega * sum.x; o.y = yy.y + omega * sum.y; }
            else if (EPI == 1) {
              o.x = dg.x != 0.0 ? xr.x + omega * (rr.x - sum.x) / dg.x : xr.x;
              o.y = dg.y != 0.0 ? xr.y + omega * (rr.y - sum.y) / dg.y : xr.y;
            } else if (EPI == 5) {
              o.x = dg.x != 0.0 ? xr.x + beta * xr.x + omega * (rr.x - sum.x) / dg.x : xr.x;
              o.y = dg.y != 0.0 ? xr.y + beta * xr.y + omega * (rr.y - sum.y) / dg.y : xr.y;
            } else {
              o.x = dg.x != 0.0 ? xr.x + beta * (xr.x - yy.x) + omega * (rr.x - sum.x) / dg.x : xr.x;
              o.y = dg.y != 0.0 ? xr.y + beta * (xr.y - yy.y) + omega * (rr.y - sum.y) / dg.y : xr.y;
            }
          }
          // non-temporal: a plain store allocates the line in L2 and evicts the x windows the neighbouring tiles are
          // about to gather - measured 0.105 against 0.054 ms on the tile rows (scripts/ubench_gather.hip, v6)
          dbl2u ou; ou.x = o.x; ou.y = o.y;
          __builtin_nontemporal_store(ou, reinterpret_cast<dbl2u*>(y + row));
          if (DOT) { const dbl2u xd = *reinterpret_cast<const dbl2u*>(xdot + row); part += sum.x * xd.x; part += sum.y * xd.y; }
        }
    }
  }
  if (DOT) pat_block_sum(part, partial, pb);
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
    // 8 entries per trip: all columns and values first, then all gathers (two memory round trips per 8 entries),
    // products in storage order
    for (long long q = q0; q < q1; q += 8) {
      double vv[8], xx[8];
      int cc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long long qq = q + j < q1 ? q + j : q1 - 1;
        cc[j] = colidx[qq];
        vv[j] = vals[qq];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) xx[j] = (HALO && cc[j] >= nowned) ? xg[cc[j] - nowned] : x[cc[j]];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double s2 = sum + vv[j] * xx[j];
        sum = q + j < q1 ? s2 : sum;
      }
    }
    y[row] = pat_epilogue<EPI>(sum, row, x, rhs, diag, omega, beta, y);
    if (DOT) part = sum * xdot[row];
  }
  if (DOT) pat_block_sum(part, partial, blockIdx.x);
}

// MXG_PATW=0 selects the one-thread-per-row kernel (k_spmv_pat)
bool patw_on() { const char* e = getenv("MXG_PATW"); return !(e && atoi(e) == 0); }   // read per launch: tests switch it
// workgroups of the wave-persistent kernel: MXG_PATW_WGS per CU (default 4 = 16 wavefronts per CU), but never so many
// that a wavefront would own fewer than 8 tiles
long long patw_wgs(const mxg_ctx* c, const mxg_csr* a) {
  static const int per_cu = getenv("MXG_PATW_WGS") ? std::max(1, atoi(getenv("MXG_PATW_WGS"))) : 4;
  const long long tiles = (a->nrows + PATW_TILE - 1) / PATW_TILE;
  long long wgs = (long long)c->num_cu * per_cu;
  wgs = std::min(wgs, std::max<long long>(1, tiles / (8 * PATW_WAVES)));
  return wgs;
}
// pairs per lane of the one-shot pair kernel (MXG_PAIR_U = 2, 3 or 4)
int pair_u() { const char* e = getenv("MXG_PAIR_U"); return e ? std::min(4, std::max(2, atoi(e))) : 2; }
bool pairw_on() { const char* e = getenv("MXG_PAIRW"); return e && atoi(e) != 0; }
long long pair_wgs(const mxg_ctx* c, const mxg_csr* a) {
  if (pairw_on()) {          // MXG_PAIRW_WGS workgroups per CU (default 4), at least 8 steps per wavefront
    static const int per_cu = getenv("MXG_PAIRW_WGS") ? std::max(1, atoi(getenv("MXG_PAIRW_WGS"))) : 4;
    const long long steps = (a->nrows / 2 + PAIR_TILE * PAIRW_U - 1) / (PAIR_TILE * PAIRW_U);
    return std::min<long long>((long long)c->num_cu * per_cu, std::max<long long>(1, steps / (8 * PAIR_WAVES)));
  }
  const long long per_wg = (long long)PAIR_TILE * pair_u() * PAIR_WAVES;
  return (a->nrows / 2 + per_wg - 1) / per_wg;
}
// partial sums the pattern kernel writes (the individual-row kernel appends its own behind them)
long long pat_first_partials(const mxg_ctx* c, const mxg_csr* a) {
  if (a->d_pairid) return pair_wgs(c, a) * PAIR_WAVES;
  return patw_on() ? patw_wgs(c, a) * PATW_WAVES : (a->nrows + PAT_BLOCK - 1) / PAT_BLOCK;
}

template <bool DOT, int EPI, bool HALO>
void launch_pat_t(mxg_ctx* c, mxg_csr* a, const double* x, double* y, const double* xdot, double* partial,
                  const double* rhs, const double* diag, double omega, double beta, const double* xg, long long nowned,
                  bool pattern_rows, bool generic_rows) {
  if (a->d_tiledesc) {
    if (HALO) { mxg_set_error("mxg_pat_launch: the tile product does not read ghosts"); return; }
    const long long ntb = (a->pat_ntile + TILE_WAVES * TILE_TPW - 1) / (TILE_WAVES * TILE_TPW), ngb = (a->pat_ngrows + 64 * TILE_WAVES - 1) / (64 * TILE_WAVES);
    const long long ntb8 = ((ntb + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD, ngb8 = ((ngb + 7) / 8) * 8;
    if (ntb8 + ngb8 > 0)
      hipLaunchKernelGGL((k_spmv_tile<DOT, EPI>), dim3((unsigned)(ngb8 + ntb8)), dim3(64 * TILE_WAVES), 0, c->stream,
                         (const TileDesc*)a->d_tiledesc, (const TileEntry*)a->d_tiledict, (long long)a->pat_ntile, ntb, ngb,
                         (const int*)a->d_grows, (long long)a->pat_ngrows, (const long long*)a->d_rowptr, (const int*)a->d_colidx,
                         (const double*)a->d_vals, x, y, xdot, partial, rhs, diag, omega, beta,
                         getenv("MXG_TILE_DIAG_SKIP_IRREGULAR") != nullptr,    // diagnostic: wrong results
                         getenv("MXG_TILE_SPREAD") != nullptr && atoi(getenv("MXG_TILE_SPREAD")) != 0);
    return;
  }
  const long long nb = a->d_pairid ? pair_wgs(c, a) : patw_on() ? patw_wgs(c, a) : (a->nrows + PAT_BLOCK - 1) / PAT_BLOCK;
  if (pattern_rows && nb > 0 && a->d_pairid) {
    if (HALO) { mxg_set_error("mxg_pat_launch: the row-pair product does not read ghosts"); return; }
    const unsigned grid = (unsigned)(((nb + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD);
    if (pairw_on())
      hipLaunchKernelGGL((k_spmv_pairw<DOT, EPI>), dim3(grid), dim3(64 * PAIR_WAVES), 0, c->stream,
                         (const unsigned short*)a->d_pairid, (const PairEntry*)a->d_pairdict, (long long)(a->nrows / 2), nb, x, y,
                         xdot, partial, rhs, diag, omega, beta);
    else if (pair_u() == 4)
      hipLaunchKernelGGL((k_spmv_pair<DOT, EPI, 4>), dim3(grid), dim3(64 * PAIR_WAVES), 0, c->stream,
                         (const unsigned short*)a->d_pairid, (const PairEntry*)a->d_pairdict, (long long)(a->nrows / 2), nb, x, y,
                         xdot, partial, rhs, diag, omega, beta);
    else if (pair_u() == 3)
      hipLaunchKernelGGL((k_spmv_pair<DOT, EPI, 3>), dim3(grid), dim3(64 * PAIR_WAVES), 0, c->stream,
                         (const unsigned short*)a->d_pairid, (const PairEntry*)a->d_pairdict, (long long)(a->nrows / 2), nb, x, y,
                         xdot, partial, rhs, diag, omega, beta);
    else
      hipLaunchKernelGGL((k_spmv_pair<DOT, EPI, 2>), dim3(grid), dim3(64 * PAIR_WAVES), 0, c->stream,
                         (const unsigned short*)a->d_pairid, (const PairEntry*)a->d_pairdict, (long long)(a->nrows / 2), nb, x, y,
                         xdot, partial, rhs, diag, omega, beta);
  } else if (pattern_rows && nb > 0 && patw_on()) {
    const unsigned grid = (unsigned)(((nb + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD);
    hipLaunchKernelGGL((k_spmv_patw<DOT, EPI, HALO>), dim3(grid), dim3(64 * PATW_WAVES), 0, c->stream,
                       (const unsigned short*)a->d_rowpid, (const PatEntry*)a->d_patdict, (long long)a->nrows, nb, x, y,
                       xdot, partial, rhs, diag, omega, beta, xg, nowned);
  } else if (pattern_rows && nb > 0) {
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
                       (const double*)a->d_vals, x, y, xdot, DOT ? partial + pat_first_partials(c, a) : partial, rhs, diag, omega, beta, xg, nowned);
  }
}

}  // namespace

int64_t mxg_pat_partials(const mxg_csr* a) {
  if (a->d_tiledesc) {
    const int64_t ntb = (a->pat_ntile + TILE_WAVES * TILE_TPW - 1) / (TILE_WAVES * TILE_TPW), ngb = (a->pat_ngrows + 64 * TILE_WAVES - 1) / (64 * TILE_WAVES);
    return ((ngb + 7) / 8) * 8 + ((ntb + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD;
  }
  return pat_first_partials(a->ctx, a) + (a->pat_ngrows + PAT_BLOCK - 1) / PAT_BLOCK;
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
      // on the Yee lattice a row repeats the pattern of the row three (same component, previous cell of the grid line) or
      // one before it for hundreds of rows: compare the CSR entries directly and skip the hash for those
      bool reused = false;
      for (int back = 3; back >= 1 && !reused; back -= 2) {
        const int64_t pr = r - back;
        if (pr < r0 || rowpid[(size_t)pr] == PAT_NONE) continue;
        const int64_t p0 = rp[pr];
        if (rp[pr + 1] - p0 != len) continue;
        bool same = true;
        for (int i = 0; i < len && same; ++i)
          same = ((int64_t)ci[q0 + i] - r == (int64_t)ci[p0 + i] - pr) && std::memcmp(&va[q0 + i], &va[p0 + i], 8) == 0;
        if (same) { rowpid[(size_t)r] = rowpid[(size_t)pr]; reused = true; }
      }
      if (reused) continue;
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

// Row-pair dictionary: pairs (2m, 2m+1) of rows with the same length (<= 8) and the same column offsets share a
// pattern; everything else (and the last row of an odd matrix) goes to the list of individual rows.
int mxg_pair_build(mxg_csr* a, const int64_t* rp, const int32_t* ci, const double* va) {
  mxg_ctx* c = a->ctx;
  const int64_t n = a->nrows, npairs = n / 2;
  std::vector<unsigned short> pairid((size_t)std::max<int64_t>(npairs, 1), PAT_NONE);
  std::vector<PairEntry> dict;
  std::unordered_map<std::string, int> global;
  std::mutex mu;
  auto pairable = [&](int64_t m) -> int {          // length of the shared pattern, -1 if the rows do not pair
    const int64_t r = 2 * m, qa = rp[r], qb = rp[r + 1];
    const int64_t len = qb - qa;
    if (rp[r + 2] - qb != len || len > PAT_LEN) return -1;
    for (int64_t i = 0; i < len; ++i)
      if ((int64_t)ci[qa + i] - r != (int64_t)ci[qb + i] - (r + 1)) return -1;
    return (int)len;
  };
  auto work = [&](int64_t m_lo, int64_t m_hi) {
    std::unordered_map<std::string, int> local;
    std::string sg;
    for (int64_t m = m_lo; m < m_hi; ++m) {
      const int len = pairable(m);
      if (len < 0) continue;
      const int64_t r = 2 * m, qa = rp[r], qb = rp[r + 1];
      // the pair three before (same components, previous cells of the grid line) usually has the same pattern
      bool reused = false;
      for (int back = 3; back >= 1 && !reused; back -= 2) {
        const int64_t pm = m - back;
        if (pm < m_lo || pairid[(size_t)pm] == PAT_NONE) continue;
        const int64_t pr = 2 * pm, pa = rp[pr], pb = rp[pr + 1];
        if (pb - pa != len || rp[pr + 2] - pb != len) continue;
        bool same = true;
        for (int i = 0; i < len && same; ++i)
          same = ((int64_t)ci[qa + i] - r == (int64_t)ci[pa + i] - pr) && std::memcmp(&va[qa + i], &va[pa + i], 8) == 0 &&
                 std::memcmp(&va[qb + i], &va[pb + i], 8) == 0;
        if (same) { pairid[(size_t)m] = pairid[(size_t)pm]; reused = true; }
      }
      if (reused) continue;
      sg.assign(4 + (size_t)len * 20, '\0');
      std::memcpy(&sg[0], &len, 4);
      for (int i = 0; i < len; ++i) {
        const int32_t dl = (int32_t)((int64_t)ci[qa + i] - r);
        std::memcpy(&sg[4 + (size_t)i * 20], &dl, 4);
        std::memcpy(&sg[4 + (size_t)i * 20 + 4], &va[qa + i], 8);
        std::memcpy(&sg[4 + (size_t)i * 20 + 12], &va[qb + i], 8);
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
          PairEntry e{};
          e.len = len;
          for (int i = 0; i < len; ++i) {
            e.delta[i] = (int32_t)((int64_t)ci[qa + i] - r);
            e.va[i] = va[qa + i];
            e.vb[i] = va[qb + i];
          }
          id = (int)dict.size();
          dict.push_back(e);
          global.emplace(sg, id);
        } else {
          id = -1;
        }
        local.emplace(sg, id);
      }
      if (id >= 0) pairid[(size_t)m] = (unsigned short)id;
    }
  };
  unsigned nthreads = std::thread::hardware_concurrency();
  if (const char* e = getenv("OMP_NUM_THREADS")) { const int v = atoi(e); if (v > 0) nthreads = (unsigned)v; }
  if (nthreads > 32) nthreads = 32;
  if (nthreads < 1 || npairs < 4096) nthreads = 1;
  std::vector<std::thread> pool;
  for (unsigned i = 0; i < nthreads; ++i) pool.emplace_back(work, npairs * i / nthreads, npairs * (i + 1) / nthreads);
  for (auto& th : pool) th.join();
  std::vector<int32_t> grows;
  for (int64_t m = 0; m < npairs; ++m)
    if (pairid[(size_t)m] == PAT_NONE) { grows.push_back((int32_t)(2 * m)); grows.push_back((int32_t)(2 * m + 1)); }
  if (n & 1) grows.push_back((int32_t)(n - 1));
  if (a->d_pairid) hipFree(a->d_pairid);
  if (a->d_pairdict) hipFree(a->d_pairdict);
  if (a->d_grows) hipFree(a->d_grows);
  a->d_pairid = nullptr; a->d_pairdict = nullptr; a->d_grows = nullptr;
  MXG_HIP(hipMalloc(&a->d_pairid, sizeof(unsigned short) * pairid.size()));
  MXG_HIP(hipMalloc(&a->d_pairdict, sizeof(PairEntry) * std::max<size_t>(dict.size(), 1)));
  MXG_HIP(hipMalloc(&a->d_grows, sizeof(int32_t) * std::max<size_t>(grows.size(), 1)));
  MXG_HIP(hipMemcpyAsync(a->d_pairid, pairid.data(), sizeof(unsigned short) * pairid.size(), hipMemcpyHostToDevice, c->stream));
  if (!dict.empty())
    MXG_HIP(hipMemcpyAsync(a->d_pairdict, dict.data(), sizeof(PairEntry) * dict.size(), hipMemcpyHostToDevice, c->stream));
  if (!grows.empty())
    MXG_HIP(hipMemcpyAsync(a->d_grows, grows.data(), sizeof(int32_t) * grows.size(), hipMemcpyHostToDevice, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  a->pat_npair = (int64_t)dict.size();
  a->pat_ngrows = (int64_t)grows.size();
  int64_t gnnz = 0;
  for (int32_t r : grows) gnnz += rp[r + 1] - rp[r];
  a->pat_gnnz = gnnz;
  if (getenv("MXG_PAT_STATS"))
    fprintf(stderr, "[mxg_pat] %lld rows: %lld pair patterns, %lld individual rows (%.2f %%, %lld entries)\n", (long long)n,
            (long long)dict.size(), (long long)grows.size(), 100.0 * grows.size() / std::max<int64_t>(n, 1), (long long)gnnz);
  return 0;
}

// Tile form: runs of rows with one length and one list of column offsets in which each row repeats the values of the row
// three before it, cut into tiles of <= TILE_ROWS rows (a multiple of 6 except at the end of a run, always even).
int mxg_tile_build(mxg_csr* a, const int64_t* rp, const int32_t* ci, const double* va) {
  mxg_ctx* c = a->ctx;
  const int64_t n = a->nrows;
  // step[r] bit 0: row r has the length and column offsets of row r - 1; bit 1: ... and the values of row r - 3
  std::vector<unsigned char> step((size_t)std::max<int64_t>(n, 1), 0);
  auto same_shape = [&](int64_t r, int64_t p) {
    const int64_t q = rp[r], t = rp[p], len = rp[r + 1] - q;
    if (rp[p + 1] - t != len || len > PAT_LEN) return false;
    for (int64_t i = 0; i < len; ++i)
      if ((int64_t)ci[q + i] - r != (int64_t)ci[t + i] - p) return false;
    return true;
  };
  auto same_vals = [&](int64_t r, int64_t p) {
    const int64_t q = rp[r], t = rp[p], len = rp[r + 1] - q;
    return std::memcmp(va + q, va + t, sizeof(double) * (size_t)len) == 0;
  };
  {
    unsigned nthreads = std::thread::hardware_concurrency();
    if (const char* e = getenv("OMP_NUM_THREADS")) { const int v = atoi(e); if (v > 0) nthreads = (unsigned)v; }
    if (nthreads > 32) nthreads = 32;
    if (nthreads < 1 || n < 4096) nthreads = 1;
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < nthreads; ++i)
      pool.emplace_back([&, i] {
        for (int64_t r = std::max<int64_t>(1, n * i / nthreads); r < n * (i + 1) / nthreads; ++r) {
          unsigned char f = 0;
          if (rp[r + 1] - rp[r] <= PAT_LEN && same_shape(r, r - 1)) {
            f = 1;
            if (r >= 3 && same_shape(r, r - 3) && same_vals(r, r - 3)) f = 3;
          }
          step[(size_t)r] = f;
        }
      });
    for (auto& th : pool) th.join();
  }
  std::vector<TileDesc> tiles;
  std::vector<TileEntry> dict;
  std::unordered_map<std::string, int> ids;
  std::vector<int32_t> grows;
  std::string sg;
  int64_t covered = 0;
  int64_t r = 0;
  while (r < n) {
    // the run that starts at r: rows r+1, r+2 need the shape of their predecessor, later rows also the values of r-3
    int64_t e = r + 1;
    if (rp[r + 1] - rp[r] <= PAT_LEN) {
      while (e < n && (step[(size_t)e] & 1) && (e - r < 3 || (step[(size_t)e] & 2))) ++e;
    }
    int64_t len_run = e - r;
    if (len_run & 1) --len_run;                               // tiles hold an even number of rows
    if (len_run >= TILE_MIN_RUN) {
      const int len = (int)(rp[r + 1] - rp[r]);
      sg.assign(4 + (size_t)len * 4 + (size_t)len * 24, '\0');
      std::memcpy(&sg[0], &len, 4);
      TileEntry te{};
      te.len = len;
      te.dslot = -1;
      int ndiag = 0;
      for (int i = 0; i < len; ++i) {
        te.delta[i] = (int32_t)((int64_t)ci[rp[r] + i] - r);
        if (te.delta[i] == 0) { te.dslot = i; ++ndiag; }
        for (int k = 0; k < 3; ++k) te.v[k][i] = va[rp[r + k] + i];     // len_run >= 24: rows r, r+1, r+2 exist in the run
        std::memcpy(&sg[4 + (size_t)i * 28], &te.delta[i], 4);
        for (int k = 0; k < 3; ++k) std::memcpy(&sg[4 + (size_t)i * 28 + 4 + 8 * k], &te.v[k][i], 8);
      }
      if (ndiag != 1) te.dslot = -1;                            // d_diag sums repeated diagonal entries: read it then
      auto it = ids.find(sg);
      int id;
      if (it != ids.end()) id = it->second;
      else { id = (int)dict.size(); dict.push_back(te); ids.emplace(sg, id); }
      for (int64_t s0 = r; s0 < r + len_run; s0 += TILE_ROWS) {
        TileDesc d;
        d.start = (int)s0; d.nrows = (int)std::min<int64_t>(TILE_ROWS, r + len_run - s0); d.pid = id; d.pad = 0;
        tiles.push_back(d);
      }
      covered += len_run;
      for (int64_t q = r + len_run; q < e; ++q) grows.push_back((int32_t)q);   // the odd row at the end of the run
    } else {
      for (int64_t q = r; q < e; ++q) grows.push_back((int32_t)q);
    }
    r = e;
  }
  if (a->d_tiledesc) hipFree(a->d_tiledesc);
  if (a->d_tiledict) hipFree(a->d_tiledict);
  if (a->d_grows) hipFree(a->d_grows);
  a->d_tiledesc = nullptr; a->d_tiledict = nullptr; a->d_grows = nullptr;
  a->pat_tile_coverage = n > 0 ? (double)covered / (double)n : 0.0;
  if (getenv("MXG_PAT_STATS"))
    fprintf(stderr, "[mxg_pat] %lld rows: %lld tiles of %lld tile patterns cover %.2f %%, %lld irregular rows\n", (long long)n,
            (long long)tiles.size(), (long long)dict.size(), 100.0 * a->pat_tile_coverage, (long long)grows.size());
  const double min_cov = getenv("MXG_TILE_MIN_COVERAGE") ? atof(getenv("MXG_TILE_MIN_COVERAGE")) : 0.80;
  if (a->pat_tile_coverage < min_cov) { a->pat_ntile = 0; a->pat_ntilepat = 0; return 0; }   // not a lattice operator: keep k_spmv_cs
  MXG_HIP(hipMalloc(&a->d_tiledesc, sizeof(TileDesc) * std::max<size_t>(tiles.size(), 1)));
  MXG_HIP(hipMalloc(&a->d_tiledict, sizeof(TileEntry) * std::max<size_t>(dict.size(), 1)));
  MXG_HIP(hipMalloc(&a->d_grows, sizeof(int32_t) * std::max<size_t>(grows.size(), 1)));
  if (!tiles.empty()) MXG_HIP(hipMemcpyAsync(a->d_tiledesc, tiles.data(), sizeof(TileDesc) * tiles.size(), hipMemcpyHostToDevice, c->stream));
  if (!dict.empty()) MXG_HIP(hipMemcpyAsync(a->d_tiledict, dict.data(), sizeof(TileEntry) * dict.size(), hipMemcpyHostToDevice, c->stream));
  if (!grows.empty()) MXG_HIP(hipMemcpyAsync(a->d_grows, grows.data(), sizeof(int32_t) * grows.size(), hipMemcpyHostToDevice, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  a->pat_ntile = (int64_t)tiles.size();
  a->pat_ntilepat = (int64_t)dict.size();
  a->pat_ngrows = (int64_t)grows.size();
  return 0;
}
