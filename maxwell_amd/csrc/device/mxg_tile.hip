// Tile product ("tile"): the product kernel of the eigen-solver's inner loops on lattice operators.
//
// The operators MxMagWaveOp iterates on (shifted vector Laplacian, scalar Laplacian, the multigrid levels; reference
// src/MxMagWaveOp.cpp:826-943 -> MxCrsMatrix::apply, src/MxCrsMatrix.cpp:347-353) are stencils on the Yee lattice:
// along a grid line (z fastest, three components interleaved) consecutive rows have the same length and the same
// column offsets, and every row repeats the values of the row three before it.  mxg_csr_compress re-expresses such a
// matrix as
//   tiles      a run of such rows cut into pieces of <= 126 rows (63 lanes x 2 rows): 16 bytes of descriptor
//              {first row, rows, pattern id}; a pattern = {length, offsets, values of row offsets 0, 1, 2 mod 3}.
//              A wavefront reads descriptor and pattern with SCALAR loads, so its gather addresses are SGPR base +
//              lane * 16 with no per-lane dependent load in front of them, two rows per lane share each 16-byte gather
//              and a lane picks its two value lists by (2 lane) mod 3.  Patterns of <= 8 entries (the 7-point bulk
//              stencil) and of 9..16 entries (13-point rows inside dielectrics) are two classes of tiles with their own
//              dictionaries, walked by different workgroups of the same launch.
//   groups     rows outside tiles (the cavity wall, line ends, rows that read ghost columns) in ELL form, 64 rows per
//              group: row ids [64], columns [width][64], values [width][64].  One wavefront per group; every load of
//              the matrix is coalesced and independent of every other (the CSR walk it replaces had three dependent,
//              strided round trips per row and cost 0.06 of the 0.13 ms of a 256^3 product for 5 % of the rows).
//   long rows  rows with more than 40 entries: one thread per row over the CSR arrays (never on a lattice operator).
// Values, columns and the order of every row sum are those of the CSR arrays: results are bit-identical to the serial
// row loop of the oracle (oracle/spmv_ref.c), for real and complex scalars, for k columns per pass (the pattern is
// loaded once for all columns of a block of vectors), and on a row-partitioned matrix: tiles never read ghost
// columns (rows that do are group rows), so tiles and ghost-free groups run while the halo is in flight and the
// ghost-reading groups follow in a second launch behind the halo event.
// The multigrid epilogues (EPI, see mxg_csr.hip) and the fused p.Ap of CG (DOT) ride along (real scalars, one column).
#include <algorithm>
#include <string>
#include <thread>
#include <unordered_map>

#include "mxg_internal.h"

namespace {

constexpr int TL_SHORT = 8, TL_LONG = 16;      // entries of a short / long pattern
constexpr int TILE_ROWS = 126, TILE_ROWS_C = 63, TILE_WAVES = 4, TILE_MIN_RUN = 24, TILE_TPW = 2;
constexpr int TILE_BLOCK = 64 * TILE_WAVES;
constexpr int TILE_ORDER = 2;                  // placement of the group octets in the grid (k_spmv_tile): spread, 0.116 against 0.122 ms first / last at 256^3
constexpr int GEN_SORT_WINDOW = 4096;          // irregular rows are sorted by length within windows of this many before grouping
constexpr int GEN_MAXW = 40;                   // widest ELL group (35: dielectric interface rows of the FIT stencil); longer rows are "long rows"

typedef double dbl2p __attribute__((ext_vector_type(2)));
typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));   // 16-byte access at 8-byte alignment

struct TileDesc { int start, nrows, pid, pad; };                       // 16 bytes
struct TileEntry8 {                                                  // short pattern: 256 bytes
  int len;
  int dslot;                                                         // index of the diagonal entry (offset 0), -1 without one
  int uniform;                                                       // 1: the three components carry the same values (v[0] = v[1] = v[2] bit for bit)
  int pad0[5];
  int delta[8];
  double v[3][8];
};
struct TileEntry16 {                                                 // long pattern: 512 bytes
  int len, dslot, uniform, pad0[5];
  int delta[16];
  double v[3][16];
  int pad1[8];
};
static_assert(sizeof(TileEntry8) == 256 && sizeof(TileEntry16) == 512, "tile dictionary entry layout");
struct TileEntryC {                                                  // complex: 1024 bytes, read with vector loads by the lanes
  int maxlen, pad0[3];
  int len[4];                                                        // per row offset 0, 1, 2 mod 3: the three components of a
  int delta[3][16];                                                  // cell may have different shapes (in complex elements)
  double v[3][16][2];
  int pad1[8];
};
static_assert(sizeof(TileEntryC) == 1024, "complex tile dictionary entry layout");
struct GenGroup { long long off; int width, pad; };                  // off: first slot of the group in gcol / gval (64 per entry slot)
static_assert(sizeof(GenGroup) == 16, "group descriptor layout");

// What a launch needs beyond its pointer parameters, passed by value.  The descriptors and dictionaries a wavefront
// reads with scalar loads are separate `const T* __restrict__` kernel parameters: the compiler only emits s_load for
// memory it can prove read-only, which it cannot for pointers that arrive inside a struct (measured: the tile rows ran
// 2.3 x slower with every pattern value fetched by a vector load).
struct TileArgs {
  long long ntA, ntbA;                                                // short-pattern tiles, their workgroups
  long long ntB, ntbB;                                                // long-pattern tiles
  const int* grow; const int* gcol; const double* gval; long long ng0, ngn;   // groups [ng0, ng0 + ngn)
  const int* lrows; long long nl0, nln;                               // long rows [nl0, nl0 + nln)
  const long long* rowptr; const int* colidx; const double* vals;
  long long ngbG, ngbL, ngb;                                          // workgroups of groups / long rows; ngb = each rounded up to 8, summed
  long long ldx, ldy; int order;                                      // order: placement of the group octets (kernel comment)
  const double* xdot; double* partial; const double* rhs; const double* diag; double omega, beta;
  const double* xg; long long ldg; long long nowned;
};

// the epilogues of mxg_csr.hip's cs_epilogue, restated (same arithmetic, same order)
template <int EPI>
__device__ inline double tile_epilogue(double ax, long long row, const double* __restrict__ x,
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

__device__ inline long long tile_xcd_block(long long w, long long nblocks) {
  const long long per = (nblocks + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  return (w % MXG_NUM_XCD) * per + (w / MXG_NUM_XCD);
}

// block sum of `part` in a fixed order (wave shuffles, then the four wave sums in wave order) -> partial[b]
__device__ inline void tile_block_sum(double part, double* __restrict__ partial, long long b) {
  __shared__ double sred[TILE_WAVES];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < TILE_WAVES; ++w) tot += sred[w];
    partial[b] = tot;
  }
}

// ---- real tiles: two rows per lane, pattern in SGPRs ---------------------------------------------------------------
template <typename ENTRY, int NL, int TPW, bool DOT, int EPI>
__device__ inline void tile_rows_real(const TileDesc* __restrict__ desc, const ENTRY* __restrict__ dict, long long ntiles,
                                      long long ntb, long long tile_block, const double* __restrict__ x, double* __restrict__ y,
                                      const TileArgs& A, double& part) {
  const long long tb = tile_xcd_block(tile_block, ntb);            // XCD w % 8 walks a contiguous eighth of the tiles
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const long long t0 = (tb * TILE_WAVES + wave) * TPW;             // TPW consecutive tiles per wavefront
  if (!(tb < ntb && t0 < ntiles)) return;
  const int ca = (2 * lane) % 3, cb = (2 * lane + 1) % 3;
  TileDesc d[TPW];
  const ENTRY* __restrict__ e[TPW];
  int len[TPW];
  bool on[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {                                  // wave-uniform: scalar loads
    const long long t = min(t0 + u, ntiles - 1);
    d[u] = desc[t];
    e[u] = dict + d[u].pid;
    len[u] = e[u]->len;
    on[u] = t0 + u < ntiles && 2 * lane < d[u].nrows;              // nrows is even
  }
  dbl2p xv[TPW][NL];
#pragma unroll
  for (int u = 0; u < TPW; ++u)
    if (on[u]) {
#pragma unroll
      for (int i = 0; i < NL; ++i)
        if (i < len[u]) {                                          // uniform
          const double* xp = x + ((long long)d[u].start + e[u]->delta[i]);   // scalar base
          const dbl2u tt = *reinterpret_cast<const dbl2u*>(xp + 2 * lane);
          xv[u][i].x = tt.x; xv[u][i].y = tt.y;
        }
    }
#pragma unroll
  for (int u = 0; u < TPW; ++u)
    if (on[u]) {
      const long long row = (long long)d[u].start + 2 * lane;
      dbl2p sum; sum.x = 0.0; sum.y = 0.0;
      const bool uni = e[u]->uniform != 0;                           // wave-uniform: a scalar branch
      if (uni) {
        // one value per entry for all three components (the interior of a uniform lattice): no per-lane selection -
        // 4 v_cndmask per value and row pair less, a third of the kernel's vector instructions
#pragma unroll
        for (int i = 0; i < NL; ++i)
          if (i < len[u]) {
            const double v0 = e[u]->v[0][i];
            sum.x = sum.x + v0 * xv[u][i].x;
            sum.y = sum.y + v0 * xv[u][i].y;
          }
      } else {
#pragma unroll
        for (int i = 0; i < NL; ++i)
          if (i < len[u]) {
            const double v0 = e[u]->v[0][i], v1 = e[u]->v[1][i], v2 = e[u]->v[2][i];
            const double va = ca == 0 ? v0 : (ca == 1 ? v1 : v2), vb = cb == 0 ? v0 : (cb == 1 ? v1 : v2);
            sum.x = sum.x + va * xv[u][i].x;
            sum.y = sum.y + vb * xv[u][i].y;
          }
      }
      dbl2p o, rr;
      if (EPI == 0) {
        o = sum;
      } else {
        dbl2p dg, xr, yy;
        if (EPI != 3) { const dbl2u q = *reinterpret_cast<const dbl2u*>(A.rhs + row); rr.x = q.x; rr.y = q.y; }
        if (EPI == 1 || EPI == 4 || EPI == 5) {
          const dbl2u q2 = *reinterpret_cast<const dbl2u*>(x + row);
          xr.x = q2.x; xr.y = q2.y;
          const int ds = e[u]->dslot;                              // uniform
          if (ds >= 0) {                                           // the diagonal is an entry of the pattern: no load
            const double d0 = e[u]->v[0][ds], d1 = e[u]->v[1][ds], d2 = e[u]->v[2][ds];
            dg.x = ca == 0 ? d0 : (ca == 1 ? d1 : d2);
            dg.y = cb == 0 ? d0 : (cb == 1 ? d1 : d2);
          } else {
            const dbl2u q1 = *reinterpret_cast<const dbl2u*>(A.diag + row);
            dg.x = q1.x; dg.y = q1.y;
          }
        }
        if (EPI == 3 || EPI == 4) { const dbl2u q = *reinterpret_cast<const dbl2u*>(y + row); yy.x = q.x; yy.y = q.y; }
        const double omega = A.omega, beta = A.beta;
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
      if (DOT) {
        if (EPI == 0) { const dbl2u xd = *reinterpret_cast<const dbl2u*>(A.xdot + row); part += sum.x * xd.x; part += sum.y * xd.y; }
        else { part += o.x * rr.x; part += o.y * rr.y; }           // smoother step: (new iterate) . rhs
      }
    }
}

// ---- complex tiles: one row per lane (16 bytes of x per entry); the lane reads the shape and the values of its row
// offset mod 3 from the dictionary with vector loads (three distinct addresses per instruction), so the three
// components of a cell may have different shapes: the lossy-dielectric operator of config 5 has 15-entry rows whose
// cross-component entries are complex rounding residue (kept by purgeZeros as in the reference), shaped per component
template <int TPW>
__device__ inline void tile_rows_cplx(const TileDesc* __restrict__ desc, const TileEntryC* __restrict__ dict, long long ntiles,
                                      long long ntb, long long tile_block, const double* __restrict__ x0, double* __restrict__ y0) {
  const long long tb = tile_xcd_block(tile_block, ntb);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const long long t0 = (tb * TILE_WAVES + wave) * TPW;
  if (!(tb < ntb && t0 < ntiles)) return;
  const int cr = lane % 3;                                         // tiles start at a multiple of three rows of their run
  const dbl2p* __restrict__ x = reinterpret_cast<const dbl2p*>(x0);
  dbl2p* __restrict__ y = reinterpret_cast<dbl2p*>(y0);
  for (int u = 0; u < TPW; ++u) {
    if (t0 + u >= ntiles) break;
    const TileDesc d = desc[t0 + u];
    const TileEntryC* __restrict__ e = dict + d.pid;
    const int maxlen = e->maxlen;                                  // uniform
    const bool on = lane < d.nrows;
    const int mylen = on ? e->len[cr] : 0;
    const int* __restrict__ dp = e->delta[cr];
    const dbl2p* __restrict__ vp = reinterpret_cast<const dbl2p*>(&e->v[cr][0][0]);
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int c = 0; c < TL_LONG / 8; ++c)
      if (c * 8 < maxlen) {
        dbl2p xv[8], vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (c * 8 + i < maxlen) {
            const bool live = c * 8 + i < mylen;
            xv[i] = x[live ? (long long)d.start + dp[c * 8 + i] + lane : 0];   // dead slot: any valid address, the product is masked
            vv[i] = vp[c * 8 + i];
          }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (c * 8 + i < maxlen) {
            const double pr = vv[i].x * xv[i].x - vv[i].y * xv[i].y;
            const double pi = vv[i].x * xv[i].y + vv[i].y * xv[i].x;
            const double r2 = sr + pr, i2 = si + pi;
            sr = c * 8 + i < mylen ? r2 : sr;
            si = c * 8 + i < mylen ? i2 : si;
          }
      }
    if (on) {
      dbl2p o; o.x = sr; o.y = si;
      __builtin_nontemporal_store(o, y + ((long long)d.start + lane));
    }
  }
}

// ---- group rows: ELL slabs of 64 rows, one wavefront per group ----------------------------------------------------
template <bool CPLX, bool DOT, int EPI, bool HALO>
__device__ inline void group_rows(long long wg, const GenGroup* __restrict__ ggrp, const double* __restrict__ x,
                                  double* __restrict__ y, const double* __restrict__ xg, const TileArgs& A, double& part) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const long long gb = tile_xcd_block(wg, A.ngbG);                 // groups are in row order: XCD w % 8 walks a contiguous eighth,
  const long long gi = gb * TILE_WAVES + wave;                     // the part of x its tile workgroups stream through its L2
  if (gb >= A.ngbG || gi >= A.ngn) return;
  const long long g = A.ng0 + gi;
  const GenGroup gd = ggrp[g];                                     // uniform: scalar load
  const int row = A.grow[g * 64 + lane];                           // -1: padding lane of the last group
  const int* __restrict__ cp = A.gcol + gd.off + lane;
  constexpr int W = CPLX ? 2 : 1;
  const double* __restrict__ vp = A.gval + (gd.off + lane) * W;
  double sr = 0.0, si = 0.0;
  for (int s0 = 0; s0 < gd.width; s0 += 8) {                       // uniform trip count
    int cc[8];
    double vr[8], vi[8], xr[8], xi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int s = s0 + i < gd.width ? s0 + i : gd.width - 1;
      cc[i] = cp[(long long)s * 64];
      if (CPLX) { const dbl2p t = *reinterpret_cast<const dbl2p*>(vp + (long long)s * 128); vr[i] = t.x; vi[i] = t.y; }
      else vr[i] = vp[(long long)s * 64];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long c0 = cc[i] >= 0 ? cc[i] : 0;                 // padding slot: any valid address, the product is masked
      const double* src = (HALO && c0 >= A.nowned) ? xg + (c0 - A.nowned) * W : x + c0 * W;
      if (CPLX) { const dbl2p t = *reinterpret_cast<const dbl2p*>(src); xr[i] = t.x; xi[i] = t.y; }
      else xr[i] = *src;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool live = s0 + i < gd.width && cc[i] >= 0;
      if (CPLX) {
        const double pr = vr[i] * xr[i] - vi[i] * xi[i];
        const double pi = vr[i] * xi[i] + vi[i] * xr[i];
        const double r2 = sr + pr, i2 = si + pi;
        sr = live ? r2 : sr;
        si = live ? i2 : si;
      } else {
        const double s2 = sr + vr[i] * xr[i];
        sr = live ? s2 : sr;
      }
    }
  }
  if (row >= 0) {
    if (CPLX) {
      dbl2p o; o.x = sr; o.y = si;
      __builtin_nontemporal_store(o, reinterpret_cast<dbl2p*>(y) + row);
    } else {
      const double out = tile_epilogue<EPI>(sr, row, x, A.rhs, A.diag, A.omega, A.beta, y);
      __builtin_nontemporal_store(out, y + row);
      if (DOT) part += EPI == 0 ? sr * A.xdot[row] : out * A.rhs[row];
    }
  }
}

// ---- long rows: one thread per row over the CSR arrays, entries in storage order ----------------------------------
template <bool CPLX, bool DOT, int EPI, bool HALO>
__device__ inline void long_rows(long long wg, const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ xg,
                                 const TileArgs& A, double& part) {
  const long long i = wg * TILE_BLOCK + threadIdx.x;
  if (i >= A.nln) return;
  const long long row = A.lrows[A.nl0 + i];
  const long long q0 = A.rowptr[row], q1 = A.rowptr[row + 1];
  constexpr int W = CPLX ? 2 : 1;
  double sr = 0.0, si = 0.0;
  for (long long q = q0; q < q1; ++q) {
    const long long c0 = A.colidx[q];
    const double* src = (HALO && c0 >= A.nowned) ? xg + (c0 - A.nowned) * W : x + c0 * W;
    if (CPLX) {
      const double ar = A.vals[2 * q], ai = A.vals[2 * q + 1], xr = src[0], xi = src[1];
      const double pr = ar * xr - ai * xi;
      const double pi = ar * xi + ai * xr;
      sr = sr + pr;
      si = si + pi;
    } else {
      sr = sr + A.vals[q] * src[0];
    }
  }
  if (CPLX) { y[2 * row] = sr; y[2 * row + 1] = si; }
  else {
    const double out = tile_epilogue<EPI>(sr, row, x, A.rhs, A.diag, A.omega, A.beta, y);
    y[row] = out;
    if (DOT) part += EPI == 0 ? sr * A.xdot[row] : out * A.rhs[row];
  }
}

// One launch.  blockIdx.y is the column of the block of vectors (every workgroup runs straight-line single-column
// code; a loop over the columns inside the kernel cost more registers than the shared pattern loads saved).
// blockIdx.x: the workgroups of the group rows and long rows (ngb of them, a multiple of 8) and those of the
// short-pattern tiles, dealt in octets - one workgroup per XCD - so that workgroup w of either kind runs on XCD w % 8.
// A.order: 0 group octets first, 1 last, 2 spread evenly among the tile octets.
// LONGP: the kernel of the long-pattern tiles (9..16 entries) - a launch of its own, made only for matrices that have
// such tiles: their 48 pattern values per tile would otherwise set the register budget of every product (SGPR spills
// on the 7-point operators the solver iterates on).
template <bool CPLX, bool DOT, int EPI, bool HALO, bool LONGP = false>
__global__ __launch_bounds__(TILE_BLOCK) void k_spmv_tile(
    const TileDesc* __restrict__ descA, const void* __restrict__ dictA, const GenGroup* __restrict__ ggrp,
    const double* __restrict__ x0, double* __restrict__ y0, const TileArgs A) {
  double part = 0.0;
  const long long bid = blockIdx.x;
  const double* __restrict__ x = x0 + (long long)blockIdx.y * A.ldx;
  double* __restrict__ y = y0 + (long long)blockIdx.y * A.ldy;
  const double* __restrict__ xg = HALO ? A.xg + (long long)blockIdx.y * A.ldg : nullptr;
  if (LONGP) {
    tile_rows_real<TileEntry16, TL_LONG, 1, DOT, EPI>(descA, (const TileEntry16*)dictA, A.ntB, A.ntbB, bid, x, y, A, part);
  } else {
    // octet o of the grid is a group octet or a tile octet
    const long long G = A.ngb / MXG_NUM_XCD, o = bid / MXG_NUM_XCD, T = gridDim.x / MXG_NUM_XCD;
    long long gbefore;                                             // group octets in front of octet o
    bool is_group;
    if (A.order == 0) { is_group = o < G; gbefore = min(o, G); }
    else if (A.order == 1) { is_group = o >= T - G; gbefore = max(0LL, o - (T - G)); }
    else {                                                         // group octet i sits at position floor(i T / G)
      gbefore = G > 0 ? min(G, (o * G + T - 1) / T) : 0;
      is_group = gbefore < G && (gbefore * T) / G == o;
    }
    const long long lane8 = bid % MXG_NUM_XCD;
    if (is_group) {
      const long long wg = gbefore * MXG_NUM_XCD + lane8, ngbG8 = ((A.ngbG + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD;
      if (wg < ngbG8) group_rows<CPLX, DOT, EPI, HALO>(wg, ggrp, x, y, xg, A, part);      // (XCD-mapped inside: octets, not single workgroups)
      else if (wg - ngbG8 < A.ngbL) long_rows<CPLX, DOT, EPI, HALO>(wg - ngbG8, x, y, xg, A, part);
    } else {
      const long long tile_block = (o - gbefore) * MXG_NUM_XCD + lane8;
      if (CPLX) tile_rows_cplx<TILE_TPW>(descA, (const TileEntryC*)dictA, A.ntA, A.ntbA, tile_block, x, y);
      else tile_rows_real<TileEntry8, TL_SHORT, TILE_TPW, DOT, EPI>(descA, (const TileEntry8*)dictA, A.ntA, A.ntbA, tile_block, x, y, A, part);
    }
  }
  if (DOT) tile_block_sum(part, A.partial, bid);
}

// ---------------------------------------------------------------------------------------------------------------
// host side: the plan (pure host code), its upload, the launch
// ---------------------------------------------------------------------------------------------------------------
struct TilePlan {
  bool cplx = false;
  std::vector<TileDesc> tilesA, tilesB;
  std::vector<TileEntry8> dictA;
  std::vector<TileEntry16> dictB;
  std::vector<TileEntryC> dictC;
  std::vector<int32_t> grow;                 // 64 per group, -1 padding
  std::vector<GenGroup> ggrp;
  std::vector<int32_t> gcol;                 // [group][width][64], -1 = no entry
  std::vector<double> gval;                  // the same shape (x 2 for complex)
  int64_t ngrp_interior = 0;                 // groups [0, ngrp_interior) read no ghost column
  std::vector<int32_t> lrows;
  int64_t nl_interior = 0;
  int64_t covered = 0, nirregular = 0;
};

unsigned plan_threads(int64_t n) {
  unsigned nthreads = std::thread::hardware_concurrency();
  if (const char* e = getenv("OMP_NUM_THREADS")) { const int v = atoi(e); if (v > 0) nthreads = (unsigned)v; }
  if (nthreads > 32) nthreads = 32;
  if (nthreads < 1 || n < 4096) nthreads = 1;
  return nthreads;
}

// Runs of rows of <= 16 entries, none of them in a ghost column, in which each row repeats the column offsets and the
// values of the row three before it - and, for real scalars, the column offsets of the row before it (two rows of a
// lane share their gathers; a complex lane holds one row, so the three components may be shaped differently) - cut
// into tiles (real: <= 126 rows, even; complex: <= 63 rows; every tile starts a multiple of three rows into its run).
// All other rows become group rows or long rows, the ones that read no ghost column first.
void tile_plan(const int64_t* rp, const int32_t* ci, const double* va, int64_t n, int64_t nowned, bool cplx, TilePlan& P) {
  P = TilePlan();
  P.cplx = cplx;
  const int w = cplx ? 2 : 1;
  const int maxlen = TL_LONG;
  // step[r] bit 0: row r has the length and column offsets of row r - 1; bit 1: ... and the values of row r - 3;
  // bit 2: row r reads a ghost column
  std::vector<unsigned char> step((size_t)std::max<int64_t>(n, 1), 0);
  auto same_shape = [&](int64_t r, int64_t p) {
    const int64_t q = rp[r], t = rp[p], len = rp[r + 1] - q;
    if (rp[p + 1] - t != len || len > maxlen) return false;
    for (int64_t i = 0; i < len; ++i)
      if ((int64_t)ci[q + i] - r != (int64_t)ci[t + i] - p) return false;
    return true;
  };
  auto same_vals = [&](int64_t r, int64_t p) {
    const int64_t q = rp[r], t = rp[p], len = rp[r + 1] - q;
    return std::memcmp(va + q * w, va + t * w, sizeof(double) * (size_t)len * w) == 0;
  };
  {
    const unsigned nthreads = plan_threads(n);
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < nthreads; ++i)
      pool.emplace_back([&, i] {
        for (int64_t r = n * i / nthreads; r < n * (i + 1) / nthreads; ++r) {
          unsigned char f = 0;
          if (rp[r + 1] - rp[r] <= maxlen) {
            if (r >= 1 && same_shape(r, r - 1)) f |= 1;
            if (r >= 3 && same_shape(r, r - 3) && same_vals(r, r - 3)) f |= 2;
          }
          for (int64_t q = rp[r]; q < rp[r + 1]; ++q)
            if (ci[q] >= nowned) { f |= 4; break; }
          step[(size_t)r] = f;
        }
      });
    for (auto& th : pool) th.join();
  }
  static const bool uniform_on = !(getenv("MXG_TILE_UNIFORM") && atoi(getenv("MXG_TILE_UNIFORM")) == 0);   // experiments
  std::unordered_map<std::string, int> idsA, idsB;
  std::vector<int32_t> irregular;
  std::string sg;
  const int tile_rows = cplx ? TILE_ROWS_C : TILE_ROWS;
  int64_t r = 0;
  while (r < n) {
    // the run that starts at r: rows r+1, r+2 need the shape of their predecessor, later rows also the values of r-3
    int64_t e = r + 1;
    const int len = (int)(rp[r + 1] - rp[r]);
    if (len <= maxlen && !(step[(size_t)r] & 4)) {
      if (cplx)
        while (e < n && !(step[(size_t)e] & 4) && (e - r < 3 ? rp[e + 1] - rp[e] <= maxlen : (step[(size_t)e] & 2) != 0)) ++e;
      else
        while (e < n && (step[(size_t)e] & 1) && !(step[(size_t)e] & 4) && (e - r < 3 || (step[(size_t)e] & 2))) ++e;
    }
    int64_t len_run = e - r;
    if (!cplx && (len_run & 1)) --len_run;                          // real tiles hold an even number of rows
    if (len_run >= TILE_MIN_RUN) {
      const bool is_long = !cplx && len > TL_SHORT;
      // signature: the shapes and values of the rows r, r+1, r+2 (len_run >= 24: they exist in the run)
      sg.clear();
      int dslot = -1, ndiag = 0;
      for (int k = 0; k < 3; ++k) {
        const int64_t q0 = rp[r + k];
        const int lk = (int)(rp[r + k + 1] - q0);
        sg.append(reinterpret_cast<const char*>(&lk), 4);
        for (int i = 0; i < lk; ++i) {
          const int32_t dl = (int32_t)((int64_t)ci[q0 + i] - (r + k));
          if (k == 0 && dl == 0) { dslot = i; ++ndiag; }
          sg.append(reinterpret_cast<const char*>(&dl), 4);
        }
        sg.append(reinterpret_cast<const char*>(va + q0 * w), sizeof(double) * (size_t)lk * w);
      }
      if (ndiag != 1) dslot = -1;                                   // d_diag sums repeated diagonal entries: read it then
      auto& ids = is_long ? idsB : idsA;
      auto it = ids.find(sg);
      int id;
      if (it != ids.end()) {
        id = it->second;
      } else {
        if (cplx) {
          TileEntryC te{};
          for (int k = 0; k < 3; ++k) {
            const int64_t q0 = rp[r + k];
            te.len[k] = (int)(rp[r + k + 1] - q0);
            te.maxlen = std::max(te.maxlen, te.len[k]);
            for (int i = 0; i < te.len[k]; ++i) {
              te.delta[k][i] = (int32_t)((int64_t)ci[q0 + i] - (r + k));
              te.v[k][i][0] = va[(q0 + i) * 2]; te.v[k][i][1] = va[(q0 + i) * 2 + 1];
            }
          }
          id = (int)P.dictC.size(); P.dictC.push_back(te);
        } else if (is_long) {
          TileEntry16 te{};
          te.len = len; te.dslot = dslot;
          for (int i = 0; i < len; ++i) {
            te.delta[i] = (int32_t)((int64_t)ci[rp[r] + i] - r);
            for (int k = 0; k < 3; ++k) te.v[k][i] = va[rp[r + k] + i];
          }
          te.uniform = uniform_on && std::memcmp(te.v[0], te.v[1], sizeof(te.v[0])) == 0 && std::memcmp(te.v[0], te.v[2], sizeof(te.v[0])) == 0;
          id = (int)P.dictB.size(); P.dictB.push_back(te);
        } else {
          TileEntry8 te{};
          te.len = len; te.dslot = dslot;
          for (int i = 0; i < len; ++i) {
            te.delta[i] = (int32_t)((int64_t)ci[rp[r] + i] - r);
            for (int k = 0; k < 3; ++k) te.v[k][i] = va[rp[r + k] + i];
          }
          te.uniform = uniform_on && std::memcmp(te.v[0], te.v[1], sizeof(te.v[0])) == 0 && std::memcmp(te.v[0], te.v[2], sizeof(te.v[0])) == 0;
          id = (int)P.dictA.size(); P.dictA.push_back(te);
        }
        ids.emplace(sg, id);
      }
      auto& tiles = is_long ? P.tilesB : P.tilesA;
      for (int64_t s0 = r; s0 < r + len_run; s0 += tile_rows) {
        TileDesc d;
        d.start = (int)s0; d.nrows = (int)std::min<int64_t>(tile_rows, r + len_run - s0); d.pid = id; d.pad = 0;
        tiles.push_back(d);
      }
      P.covered += len_run;
      for (int64_t q = r + len_run; q < e; ++q) irregular.push_back((int32_t)q);   // the odd row at the end of the run
    } else {
      for (int64_t q = r; q < e; ++q) irregular.push_back((int32_t)q);
    }
    r = e;
  }
  P.nirregular = (int64_t)irregular.size();
  // group rows (ghost-free first) and long rows
  std::vector<int32_t> gi, gg, li, lg;
  for (int32_t q : irregular) {
    const bool ghost = (step[(size_t)q] & 4) != 0, lng = rp[q + 1] - rp[q] > GEN_MAXW;
    (lng ? (ghost ? lg : li) : (ghost ? gg : gi)).push_back(q);
  }
  P.lrows = li; P.nl_interior = (int64_t)li.size();
  P.lrows.insert(P.lrows.end(), lg.begin(), lg.end());
  // A group is as wide as its longest row: within windows of GEN_SORT_WINDOW consecutive irregular rows the rows are
  // dealt to groups in order of length (stable: equal lengths stay in row order), which brings the slots from 1.30 x
  // to 1.02 x the entries on the pillbox operators (tile product 0.124 -> 0.116 ms at 256^3); the windows keep a group's rows near each other in x.
  auto add_groups = [&](std::vector<int32_t> rows) {
    static const int window = getenv("MXG_TILE_SORT_WINDOW") ? atoi(getenv("MXG_TILE_SORT_WINDOW")) : GEN_SORT_WINDOW;   // experiments
    if (window > 64)
      for (size_t b = 0; b < rows.size(); b += (size_t)window)
        std::stable_sort(rows.begin() + (long)b, rows.begin() + (long)std::min(rows.size(), b + (size_t)window),
                         [&](int32_t a, int32_t c2) { return rp[a + 1] - rp[a] < rp[c2 + 1] - rp[c2]; });
    for (size_t b = 0; b < rows.size(); b += 64) {
      const size_t cnt = std::min<size_t>(64, rows.size() - b);
      int width = 0;
      for (size_t l = 0; l < cnt; ++l) width = std::max<int>(width, (int)(rp[rows[b + l] + 1] - rp[rows[b + l]]));
      GenGroup g; g.off = (long long)P.gcol.size(); g.width = width; g.pad = 0;
      P.ggrp.push_back(g);
      const size_t base = P.gcol.size();
      P.gcol.resize(base + (size_t)width * 64, -1);
      P.gval.resize((base + (size_t)width * 64) * w, 0.0);
      for (size_t l = 0; l < 64; ++l) {
        if (l >= cnt) { P.grow.push_back(-1); continue; }
        const int64_t row = rows[b + l], q0 = rp[row], ln = rp[row + 1] - q0;
        P.grow.push_back((int32_t)row);
        for (int64_t s = 0; s < ln; ++s) {
          P.gcol[base + (size_t)s * 64 + l] = ci[q0 + s];
          for (int c = 0; c < w; ++c) P.gval[(base + (size_t)s * 64 + l) * w + c] = va[(q0 + s) * w + c];
        }
      }
    }
  };
  add_groups(gi);
  P.ngrp_interior = (int64_t)P.ggrp.size();
  add_groups(gg);
  if (getenv("MXG_PAT_STATS")) {
    long long actual = 0, hist[GEN_MAXW + 1] = {0};
    for (int32_t q : gi) { actual += rp[q + 1] - rp[q]; hist[rp[q + 1] - rp[q]]++; }
    for (int32_t q : gg) { actual += rp[q + 1] - rp[q]; hist[rp[q + 1] - rp[q]]++; }
    fprintf(stderr, "[mxg_tile] group rows: %lld entries in %lld slots; rows by length:", actual, (long long)P.gcol.size());
    for (int l = 0; l <= GEN_MAXW; ++l) if (hist[l]) fprintf(stderr, " %d:%lld", l, hist[l]);
    fprintf(stderr, "\n");
  }
}

template <typename T>
int upload(void** dst, const std::vector<T>& v, hipStream_t s) {
  MXG_HIP(hipMalloc(dst, sizeof(T) * std::max<size_t>(v.size(), 1)));
  if (!v.empty()) MXG_HIP(hipMemcpyAsync(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, s));
  return 0;
}

inline long long ceil_div(long long a, long long b) { return (a + b - 1) / b; }
inline long long up8(long long a) { return ceil_div(a, MXG_NUM_XCD) * MXG_NUM_XCD; }

// workgroups of one phase: generic part (groups + long rows, rounded up to 8) + short-pattern tiles (rounded up to 8);
// gridB: the separate launch of the long-pattern tiles
struct PhaseGeom { long long ngbG, ngbL, ngb, ntbA, ntbB, grid, gridB; };
PhaseGeom phase_geom(const mxg_csr* a, long long ngroups, long long nlong, bool tiles) {
  PhaseGeom g;
  g.ngbG = ceil_div(ngroups, TILE_WAVES);
  g.ngbL = ceil_div(nlong, TILE_BLOCK);
  g.ngb = up8(g.ngbG) + up8(g.ngbL);
  const int tpw = TILE_TPW;
  g.ntbA = tiles ? ceil_div(a->tile_ntA, (long long)TILE_WAVES * tpw) : 0;
  g.ntbB = tiles ? ceil_div(a->tile_ntB, (long long)TILE_WAVES) : 0;
  g.grid = g.ngb + up8(g.ntbA);
  g.gridB = up8(g.ntbB);
  return g;
}

template <bool CPLX, bool DOT, int EPI>
void launch_phase(mxg_ctx* c, const mxg_csr* a, const double* x, double* y, int k, TileArgs A, const PhaseGeom& g, bool halo) {
  const GenGroup* gg = (const GenGroup*)a->d_ggrp;
  if (g.grid > 0) {
    if (halo) hipLaunchKernelGGL((k_spmv_tile<CPLX, DOT, EPI, true>), dim3((unsigned)g.grid, (unsigned)k), dim3(TILE_BLOCK), 0, c->stream,
                                 (const TileDesc*)a->d_tiledescA, (const void*)a->d_tiledictA, gg, x, y, A);
    else hipLaunchKernelGGL((k_spmv_tile<CPLX, DOT, EPI, false>), dim3((unsigned)g.grid, (unsigned)k), dim3(TILE_BLOCK), 0, c->stream,
                            (const TileDesc*)a->d_tiledescA, (const void*)a->d_tiledictA, gg, x, y, A);
  }
  if (!CPLX && g.gridB > 0) {              // long-pattern tiles: their partial sums follow those of the main launch
    if (A.partial) A.partial += g.grid;
    hipLaunchKernelGGL((k_spmv_tile<false, DOT, EPI, false, true>), dim3((unsigned)g.gridB, (unsigned)k), dim3(TILE_BLOCK), 0, c->stream,
                       (const TileDesc*)a->d_tiledescB, (const void*)a->d_tiledictB, gg, x, y, A);
  }
}

}  // namespace

void mxg_tile_free(mxg_csr* a) {
  void** p[] = {&a->d_tiledescA, &a->d_tiledictA, &a->d_tiledescB, &a->d_tiledictB, (void**)&a->d_grow, &a->d_ggrp,
                (void**)&a->d_gcol, (void**)&a->d_gval, (void**)&a->d_lrows};
  for (void** q : p) { if (*q) hipFree(*q); *q = nullptr; }
  a->tile_on = 0;
}

// Build the tile form from host CSR arrays.  The form is kept (a->tile_on = 1) when the tiles cover at least
// MXG_TILE_MIN_COVERAGE (default 0.8) of the rows; otherwise the matrix is not a lattice operator and keeps k_spmv_cs.
int mxg_tile_build(mxg_csr* a, const int64_t* rp, const int32_t* ci, const double* va) {
  mxg_ctx* c = a->ctx;
  TilePlan P;
  tile_plan(rp, ci, va, a->nrows, a->ncols_owned, a->is_complex != 0, P);
  mxg_tile_free(a);
  const int64_t n = a->nrows;
  a->pat_tile_coverage = n > 0 ? (double)P.covered / (double)n : 0.0;
  a->pat_ngrows = P.nirregular;
  a->tile_ntA = (int64_t)P.tilesA.size(); a->tile_ntB = (int64_t)P.tilesB.size();
  a->pat_ntile = a->tile_ntA + a->tile_ntB;
  a->pat_ntilepat = (int64_t)(P.dictA.size() + P.dictB.size() + P.dictC.size());
  if (getenv("MXG_PAT_STATS"))
    fprintf(stderr, "[mxg_tile] %lld rows: %lld + %lld tiles of %lld patterns cover %.2f %%, %lld irregular rows in %lld groups (%lld ghost-free) + %lld long rows\n",
            (long long)n, (long long)a->tile_ntA, (long long)a->tile_ntB, (long long)a->pat_ntilepat, 100.0 * a->pat_tile_coverage,
            (long long)P.nirregular, (long long)P.ggrp.size(), (long long)P.ngrp_interior, (long long)P.lrows.size());
  const double min_cov = getenv("MXG_TILE_MIN_COVERAGE") ? atof(getenv("MXG_TILE_MIN_COVERAGE")) : 0.80;
  if (a->pat_tile_coverage < min_cov) { a->pat_ntile = 0; a->pat_ntilepat = 0; return 0; }
  MXG_TRY(upload(&a->d_tiledescA, P.tilesA, c->stream));
  if (P.cplx) MXG_TRY(upload(&a->d_tiledictA, P.dictC, c->stream));
  else MXG_TRY(upload(&a->d_tiledictA, P.dictA, c->stream));
  MXG_TRY(upload(&a->d_tiledescB, P.tilesB, c->stream));
  MXG_TRY(upload(&a->d_tiledictB, P.dictB, c->stream));
  MXG_TRY(upload((void**)&a->d_grow, P.grow, c->stream));
  MXG_TRY(upload(&a->d_ggrp, P.ggrp, c->stream));
  MXG_TRY(upload((void**)&a->d_gcol, P.gcol, c->stream));
  MXG_TRY(upload((void**)&a->d_gval, P.gval, c->stream));
  MXG_TRY(upload((void**)&a->d_lrows, P.lrows, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  a->gen_ngrp = (int64_t)P.ggrp.size(); a->gen_ngrp_interior = P.ngrp_interior;
  a->gen_nlong = (int64_t)P.lrows.size(); a->gen_nlong_interior = P.nl_interior;
  a->tile_bytes = (int64_t)(P.tilesA.size() + P.tilesB.size()) * 16 + (int64_t)P.dictA.size() * 256 + (int64_t)P.dictB.size() * 512 +
                  (int64_t)P.dictC.size() * 1024 + (int64_t)P.grow.size() * 4 + (int64_t)P.gcol.size() * 4 + (int64_t)P.gval.size() * 8;
  a->tile_on = 1;
  return 0;
}

// partial-sum slots a DOT launch writes (both phases of a row-partitioned matrix)
int64_t mxg_tile_partials(const mxg_csr* a) {
  const bool halo = a->nghost > 0 || a->nneigh > 0;
  const PhaseGeom g1 = phase_geom(a, halo ? a->gen_ngrp_interior : a->gen_ngrp, halo ? a->gen_nlong_interior : a->gen_nlong, true);
  if (!halo) return g1.grid + g1.gridB;
  return g1.grid + g1.gridB + phase_geom(a, a->gen_ngrp - a->gen_ngrp_interior, a->gen_nlong - a->gen_nlong_interior, false).grid;
}

// y = epilogue(A x) for k columns (x, y: first column and leading dimension in doubles).  halo: the matrix has ghost
// columns and the caller has started the exchange (halo_start): tiles and ghost-free groups are launched at once, the
// stream then waits for the halo event and the ghost-reading groups follow, reading xg (column j at xg + j * ldg).
// dot: per-workgroup partial sums of (A x) . xdot go to partial[0 .. mxg_tile_partials); with epilogue 1 / 4 (a smoother
// step) the sums are (new iterate) . rhs instead - the r.z of the multigrid-CG from the cycle's last step.  epi / dot: k = 1, real.
int mxg_tile_launch(mxg_ctx* c, mxg_csr* a, const double* x, long long ldx, double* y, long long ldy, int k, int epi,
                    int dot, const double* xdot, double* partial, const double* rhs, const double* diag, double omega,
                    double beta, int halo, const double* xg, long long ldg) {
  MXG_CHECK(a->tile_on, "mxg_tile_launch: the matrix has no tile form");
  MXG_CHECK(!(a->is_complex && (epi != 0 || dot)), "mxg_tile_launch: complex products have no fused epilogue");
  MXG_CHECK(k == 1 || (epi == 0 && !dot), "mxg_tile_launch: fused epilogues take one column");
  MXG_CHECK(!dot || epi == 0 || epi == 1 || epi == 4, "mxg_tile_launch: the fused dot goes with the plain product or a smoother step (epilogue 1 / 4)");
  TileArgs A{};
  A.ntA = a->tile_ntA; A.ntB = a->tile_ntB;
  A.grow = a->d_grow; A.gcol = a->d_gcol; A.gval = a->d_gval;
  A.lrows = a->d_lrows;
  A.rowptr = (const long long*)a->d_rowptr; A.colidx = a->d_colidx; A.vals = a->d_vals;
  A.ldx = ldx; A.ldy = ldy; A.order = TILE_ORDER;
  A.xdot = xdot; A.partial = partial; A.rhs = rhs; A.diag = diag; A.omega = omega; A.beta = beta;
  A.xg = xg; A.ldg = ldg; A.nowned = a->ncols_owned;
  auto run = [&](const PhaseGeom& g, bool h) {
    A.ngbG = g.ngbG; A.ngbL = g.ngbL; A.ngb = g.ngb; A.ntbA = g.ntbA; A.ntbB = g.ntbB;
    if (g.ntbA == 0) A.ntA = 0;
    if (g.ntbB == 0) A.ntB = 0;
    if (a->is_complex) launch_phase<true, false, 0>(c, a, x, y, k, A, g, h);
    else if (dot && epi == 4) launch_phase<false, true, 4>(c, a, x, y, k, A, g, h);
    else if (dot && epi == 1) launch_phase<false, true, 1>(c, a, x, y, k, A, g, h);
    else if (dot) launch_phase<false, true, 0>(c, a, x, y, k, A, g, h);
    else if (epi == 1) launch_phase<false, false, 1>(c, a, x, y, k, A, g, h);
    else if (epi == 2) launch_phase<false, false, 2>(c, a, x, y, k, A, g, h);
    else if (epi == 3) launch_phase<false, false, 3>(c, a, x, y, k, A, g, h);
    else if (epi == 4) launch_phase<false, false, 4>(c, a, x, y, k, A, g, h);
    else if (epi == 5) launch_phase<false, false, 5>(c, a, x, y, k, A, g, h);
    else launch_phase<false, false, 0>(c, a, x, y, k, A, g, h);
  };
  if (!halo) {
    A.ng0 = 0; A.ngn = a->gen_ngrp; A.nl0 = 0; A.nln = a->gen_nlong;
    run(phase_geom(a, A.ngn, A.nln, true), false);
  } else {
    A.ng0 = 0; A.ngn = a->gen_ngrp_interior; A.nl0 = 0; A.nln = a->gen_nlong_interior;
    const PhaseGeom g1 = phase_geom(a, A.ngn, A.nln, true);
    run(g1, false);
    MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
    A.ntA = 0; A.ntB = 0;
    A.ng0 = a->gen_ngrp_interior; A.ngn = a->gen_ngrp - a->gen_ngrp_interior;
    A.nl0 = a->gen_nlong_interior; A.nln = a->gen_nlong - a->gen_nlong_interior;
    if (partial) A.partial = partial + g1.grid + g1.gridB;
    run(phase_geom(a, A.ngn, A.nln, false), true);
  }
  if (hipGetLastError() != hipSuccess) { mxg_set_error("mxg_tile_launch: launch failed"); return 1; }
  return 0;
}

// Test hook (tests/test_tile_plan.py, no GPU needed): build the plan on the host and evaluate y = A x by walking the
// PLAN's data structures with the kernels' arithmetic (every product rounded, sums in slot order).  It validates the
// setup code against the oracle in the CPU-only container; it is not a product path - nothing in the library calls it.
extern "C" int mxg_debug_tile_plan_apply(int64_t nrows, int64_t ncols_owned, const int64_t* rp, const int32_t* ci, const double* va,
                                         int is_complex, const double* x, const double* xghost, double* y, int64_t* stats) {
  TilePlan P;
  tile_plan(rp, ci, va, nrows, ncols_owned, is_complex != 0, P);
  const int w = is_complex ? 2 : 1;
  std::vector<char> written((size_t)std::max<int64_t>(nrows, 1), 0);
  auto xat = [&](int64_t c, int comp) { return c >= ncols_owned ? xghost[(c - ncols_owned) * w + comp] : x[c * w + comp]; };
  auto put = [&](int64_t row, double re, double im) {
    if (written[(size_t)row]) return 1;
    written[(size_t)row] = 1;
    y[row * w] = re; if (w == 2) y[row * w + 1] = im;
    return 0;
  };
  int dup = 0;
  auto walk_tiles = [&](const std::vector<TileDesc>& tiles, auto& dict, int nl) {
    for (const TileDesc& d : tiles) {
      const auto& e = dict[(size_t)d.pid];
      if (e.len > nl) { dup += 1000; continue; }
      for (int t = 0; t < d.nrows; ++t) {
        const int64_t row = (int64_t)d.start + t;
        double s = 0.0;
        for (int i = 0; i < e.len; ++i) s = s + (e.uniform ? e.v[0][i] : e.v[t % 3][i]) * x[row + e.delta[i]];   // as the kernel picks its value
        dup += put(row, s, 0.0);
      }
    }
  };
  if (!is_complex) {
    walk_tiles(P.tilesA, P.dictA, TL_SHORT);
    walk_tiles(P.tilesB, P.dictB, TL_LONG);
    for (const TileDesc& d : P.tilesA) if (d.nrows & 1) dup += 1000;
    for (const TileDesc& d : P.tilesB) if (d.nrows & 1) dup += 1000;
  } else {
    for (const TileDesc& d : P.tilesA) {
      const TileEntryC& e = P.dictC[(size_t)d.pid];
      for (int t = 0; t < d.nrows; ++t) {
        const int64_t row = (int64_t)d.start + t;
        double sr = 0.0, si = 0.0;
        if (e.len[t % 3] > e.maxlen || e.maxlen > TL_LONG) dup += 1000;
        for (int i = 0; i < e.len[t % 3]; ++i) {
          const double ar = e.v[t % 3][i][0], ai = e.v[t % 3][i][1], xr = x[(row + e.delta[t % 3][i]) * 2], xi = x[(row + e.delta[t % 3][i]) * 2 + 1];
          const double pr = ar * xr - ai * xi, pi = ar * xi + ai * xr;
          sr = sr + pr; si = si + pi;
        }
        dup += put(row, sr, si);
      }
    }
  }
  for (size_t g = 0; g < P.ggrp.size(); ++g)
    for (int l = 0; l < 64; ++l) {
      const int64_t row = P.grow[g * 64 + l];
      if (row < 0) continue;
      double sr = 0.0, si = 0.0;
      for (int s = 0; s < P.ggrp[g].width; ++s) {
        const int64_t slot = P.ggrp[g].off + (int64_t)s * 64 + l;
        const int64_t cc = P.gcol[(size_t)slot];
        if (cc < 0) continue;
        if ((int64_t)g < P.ngrp_interior && cc >= ncols_owned) dup += 1000;   // a ghost-free group must not read ghosts
        if (w == 2) {
          const double ar = P.gval[(size_t)slot * 2], ai = P.gval[(size_t)slot * 2 + 1], xr = xat(cc, 0), xi = xat(cc, 1);
          const double pr = ar * xr - ai * xi, pi = ar * xi + ai * xr;
          sr = sr + pr; si = si + pi;
        } else {
          sr = sr + P.gval[(size_t)slot] * xat(cc, 0);
        }
      }
      dup += put(row, sr, si);
    }
  for (size_t i = 0; i < P.lrows.size(); ++i) {
    const int64_t row = P.lrows[i];
    double sr = 0.0, si = 0.0;
    for (int64_t q = rp[row]; q < rp[row + 1]; ++q) {
      if ((int64_t)i < P.nl_interior && ci[q] >= ncols_owned) dup += 1000;
      if (w == 2) {
        const double ar = va[2 * q], ai = va[2 * q + 1], xr = xat(ci[q], 0), xi = xat(ci[q], 1);
        const double pr = ar * xr - ai * xi, pi = ar * xi + ai * xr;
        sr = sr + pr; si = si + pi;
      } else {
        sr = sr + va[q] * xat(ci[q], 0);
      }
    }
    dup += put(row, sr, si);
  }
  int64_t missing = 0;
  for (int64_t r2 = 0; r2 < nrows; ++r2) missing += !written[(size_t)r2];
  if (stats) {
    stats[0] = (int64_t)P.tilesA.size(); stats[1] = (int64_t)P.tilesB.size();
    stats[2] = (int64_t)(P.dictA.size() + P.dictB.size() + P.dictC.size());
    stats[3] = (int64_t)P.ggrp.size(); stats[4] = P.ngrp_interior; stats[5] = (int64_t)P.lrows.size();
    stats[6] = P.covered; stats[7] = missing; stats[8] = dup;
  }
  return (missing == 0 && dup == 0) ? 0 : 1;
}
