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
#include <functional>
#include <thread>

#include "mxg_internal.h"

namespace {

// Launch geometry of the stream kernel: SBLK threads own a row block of at most
// NNZ_BLK = SBLK * EPT non-zeros (8 bytes of LDS each) and ROWS_BLK rows (bounded separately
// because of empty rows).  The real kernel loads aligned pairs starting at (base & ~1), so a
// block holds at most NNZ_BLK - 1 entries to stay inside the EPT * SBLK window.
template <int SBLK_, int EPT_>
struct StreamCfg {
  static constexpr int SBLK = SBLK_;
  static constexpr int EPT = EPT_;
  static constexpr int NNZ_BLK = SBLK_ * EPT_;
  static constexpr int ROWS_BLK = NNZ_BLK / 2;
  static constexpr int NNZ_CAP = NNZ_BLK - 1;
};
typedef StreamCfg<256, 8> CfgDefault;   // 2048 nnz, 16 KiB products + 4 KiB offsets per block
constexpr int SBLK = 256;               // block size of the auxiliary kernels

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef int int2v __attribute__((ext_vector_type(2)));

template <bool NT = true, typename T>
__device__ inline T nt_load(const T* p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// XCD-aware remap of the hardware workgroup id to a logical row block:
// hardware id w runs on XCD (w % 8); give XCD x the contiguous block range
// [x * per, (x+1) * per).  A pure permutation: correctness never depends on placement.
__device__ inline long long xcd_block(long long w, long long nblocks) {
  const long long per = (nblocks + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  return (w % MXG_NUM_XCD) * per + (w / MXG_NUM_XCD);
}

// One row block per workgroup.  CPLX: values and vectors are (re, im) pairs.
template <bool CPLX, typename CFG, bool NT, bool XCD>
__global__ __launch_bounds__(CFG::SBLK) void k_spmv_stream(
    const long long* __restrict__ rowptr, const int* __restrict__ colidx,
    const double* __restrict__ vals, const int* __restrict__ rowblk, long long nblocks,
    MvDesc X, MvDesc Y) {
  constexpr int W = CPLX ? 2 : 1;
  constexpr int SBLK = CFG::SBLK, EPT = CFG::EPT, NNZ_BLK = CFG::NNZ_BLK, ROWS_BLK = CFG::ROWS_BLK,
                NNZ_CAP = CFG::NNZ_CAP;
  __shared__ double prod[NNZ_BLK * W];
  __shared__ int off[ROWS_BLK + 1];

  const long long b = XCD ? xcd_block(blockIdx.x, nblocks) : (long long)blockIdx.x;
  if (b >= nblocks) return;
  const int r0 = rowblk[b], r1 = rowblk[b + 1];
  const int nr = r1 - r0;
  const long long base = rowptr[r0];
  const int nnzb = (int)(rowptr[r1] - base);

  for (int t = threadIdx.x; t <= nr; t += SBLK) off[t] = (int)(nt_load<NT>(rowptr + r0 + t) - base);

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
          dbl2 vv = nt_load<NT>(reinterpret_cast<const dbl2*>(vals) + idx);
          v[e][0] = vv.x; v[e][W - 1] = vv.y;
          c[e] = nt_load<NT>(colidx + idx);
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
          dbl2 vv = nt_load<NT>(reinterpret_cast<const dbl2*>(vals + idx));
          int2v cc = nt_load<NT>(reinterpret_cast<const int2v*>(colidx + idx));
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
          if (NT) __builtin_nontemporal_store(out2, reinterpret_cast<dbl2*>(y) + r0 + t);
          else reinterpret_cast<dbl2*>(y)[r0 + t] = out2;
        } else {
          double sum = 0.0;
          for (int q = s; q < e; ++q) sum += prod[q];
          if (NT) __builtin_nontemporal_store(sum, y + r0 + t);
          else y[r0 + t] = sum;
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


// 32-byte row-block descriptor of the row-gather kernels
struct BlkDesc {
  long long base;   // rowptr[r0]
  int r0, nr, nnzb; // first row, rows, non-zeros of the block
  int rb;           // position of the block's 16-bit row offsets in roff[] (r0 + block number in row order)
  int pad[2];
};

// ------------------------------------------------------------------------------------------
// Row-gather kernel (real scalars): coalesced matrix stream -> LDS (raw values + columns), then
// ONE THREAD PER ROW walks its entries in LDS and gathers x.  Consecutive lanes hold consecutive
// rows, whose k-th neighbours are consecutive x entries on the Yee lattice, so every gather
// instruction touches a few contiguous 128-B lines instead of a dozen scattered ones (the
// per-non-zero gather of the stream kernel costs ~25 % in L1 tag lookups even when it hits).
// The row sum is formed in registers in storage order: bit-identical to the serial row loop.
// ------------------------------------------------------------------------------------------
template <int SBLK_, int NNZ_BLK_>
struct RgCfg {
  static constexpr int SBLK = SBLK_, NNZ_BLK = NNZ_BLK_;
  static constexpr int NNZ_CAP = NNZ_BLK_ - 4;     // the LDS image starts at a 16-byte aligned entry: <= 3 entries of slack
  static constexpr int ROWS_CAP = SBLK_;          // one row per thread
  static constexpr int EPT = NNZ_BLK_ / SBLK_;
};
typedef RgCfg<256, 2048> RgCfgA;   // 16 KiB values + 8 KiB columns per block

// DOT: additionally emit partial[b] = sum over the block's rows of y[row] * xdot[row] (the p.Ap of
// CG fused into the product; one value per block, combined in block order by k_sum_partials)
// HALO: columns >= nowned are ghosts and live in a separate buffer (column j at xghost + j * ldghost): the
// owned part of x is read in place, no [owned | ghosts] copy is assembled.
// CH: entries of a row walked per trip (7 = the bulk stencil of the vector Laplacian): the thread's index is clamped
// into its row and the products past the end are masked, so a wavefront issues the gathers of a trip back to back
// instead of diverging on the row lengths.
constexpr int RG_CH = 7;
template <typename CFG, bool DOT = false, bool HALO = false>
__global__ __launch_bounds__(CFG::SBLK) void k_spmv_rowgather(
    const BlkDesc* __restrict__ desc, long long nblocks, const long long* __restrict__ rowptr,
    const int* __restrict__ colidx, const double* __restrict__ vals, MvDesc X, MvDesc Y,
    const double* __restrict__ xdot = nullptr, double* __restrict__ partial = nullptr,
    const unsigned short* __restrict__ roff = nullptr, const double* __restrict__ xghost = nullptr,
    long long ldghost = 0, int nowned = 0) {
  constexpr int SB = CFG::SBLK, EPT = CFG::EPT, CH = RG_CH;
  __shared__ double sval[CFG::NNZ_BLK];
  __shared__ int scol[CFG::NNZ_BLK];
  const long long b = xcd_block(blockIdx.x, nblocks);
  if (b >= nblocks) return;
  const BlkDesc d = desc[b];
  if (d.nnzb > CFG::NNZ_CAP) {
    // one row longer than a block (row blocks never mix such a row with others): the whole workgroup strides over
    // it; partial sums are combined in a fixed order
    for (int j = 0; j < X.k; ++j) {
      const double* x = X.col(j);
      const double* xg = HALO ? xghost + j * ldghost : nullptr;
      double sr = 0.0;
      for (long long q = d.base + threadIdx.x; q < d.base + d.nnzb; q += SB) {
        const int cc = colidx[q];
        sr += vals[q] * ((HALO && cc >= nowned) ? xg[cc - nowned] : x[cc]);
      }
      if (j > 0) __syncthreads();
      sval[threadIdx.x] = sr;
      __syncthreads();
      if (threadIdx.x == 0) {
        double tr = 0.0;
        for (int q = 0; q < SB; ++q) tr += sval[q];
        Y.col(j)[d.r0] = tr;
        if (DOT && j == 0) partial[b] = tr * xdot[d.r0];
      }
    }
    return;
  }
  // coalesced stream -> LDS (aligned pairs from an even start, see the stream kernel)
  const long long lo = d.base & ~1LL;
  const long long end = d.base + d.nnzb;
  const int shift = (int)(lo - d.base);
#pragma unroll
  for (int e = 0; e < EPT; e += 2) {
    const long long idx = lo + 2 * (threadIdx.x + (long long)(e / 2) * SB);
    if (idx < end) {
      dbl2 vv = nt_load(reinterpret_cast<const dbl2*>(vals + idx));
      int2v cc = nt_load(reinterpret_cast<const int2v*>(colidx + idx));
      const int p = shift + 2 * (threadIdx.x + (e / 2) * SB);
      if (p >= 0) { sval[p] = vv.x; scol[p] = cc.x; }
      if (p + 1 < d.nnzb) { sval[p + 1] = vv.y; scol[p + 1] = cc.y; }
    }
  }
  int o0 = 0, o1 = 0;
  if ((int)threadIdx.x < d.nr) {
    if (roff) {
      // 16-bit in-block row offsets (rowptr[r] - base, block b starts at r0 + b): 2 B/row instead of 8
      const unsigned short* ro = roff + d.rb + threadIdx.x;
      o0 = ro[0];
      o1 = ro[1];
    } else {
      o0 = (int)(nt_load(rowptr + d.r0 + threadIdx.x) - d.base);
      o1 = (int)(nt_load(rowptr + d.r0 + threadIdx.x + 1) - d.base);
    }
  }
  __syncthreads();
  for (int j = 0; j < X.k; ++j) {
    // the LDS-resident entries of the block are walked once per column of the block of vectors
    const double* __restrict__ xc = X.col(j);
    const double* __restrict__ xgc = HALO ? xghost + j * ldghost : nullptr;
    double sum = 0.0;
    for (int q0 = o0; q0 < o1; q0 += CH) {
      double v[CH], xv[CH];
      int cc[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int q = min(q0 + i, o1 - 1);
        cc[i] = scol[q];
        v[i] = sval[q];
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (HALO) xv[i] = cc[i] >= nowned ? xgc[cc[i] - nowned] : xc[cc[i]];
        else xv[i] = xc[cc[i]];
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const double s2 = sum + v[i] * xv[i];
        sum = q0 + i < o1 ? s2 : sum;
      }
    }
    if ((int)threadIdx.x < d.nr) __builtin_nontemporal_store(sum, Y.col(j) + d.r0 + threadIdx.x);
    if (DOT && j == 0) {
      double part = (int)threadIdx.x < d.nr ? sum * xdot[d.r0 + threadIdx.x] : 0.0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
      __syncthreads();                       // every row sum is done with sval[]
      if ((threadIdx.x & 63) == 0) sval[threadIdx.x >> 6] = part;
      __syncthreads();
      if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < SB / 64; ++w) tot += sval[w];
        partial[b] = tot;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------
// Persistent, double-buffered row-gather kernel for COMPLEX scalars ("rgp"): the arithmetic of k_spmv_rowgather on
// (re, im) pairs, but a workgroup walks a strided sequence of row blocks and the NEXT block's values / columns / row
// offsets are already on their way into the other half of LDS - by LDS-DMA (global_load_lds_dwordx4: no VGPR
// staging, no ds_write) - while the current block gathers x.  A complex block holds half as many entries (1020) and
// its stream is twice as long per row, so one block per launch slot paid its four dependent round trips (descriptor
// -> stream -> gather -> store) too often; here the descriptor is two iterations ahead (scalar loads), the stream one
// iteration ahead.  Workgroups of one XCD take interleaved blocks of that XCD's contiguous eighth, so they advance
// through x together.  (For real scalars the same structure ties with k_spmv_rowgather and was dropped.)
// LDS image of a block = the global arrays from a 16-byte aligned start: entry q of the block's CSR range sits at
// sval[q - base] (16-byte entries) and scol[q - (base & ~3)]; row offsets come from a copy laid out 528 B per block.
// ------------------------------------------------------------------------------------------
constexpr int RGP_NNZ_CAP = RgCfgA::NNZ_CAP;   // real blocks (k_spmv_rowgather, k_spmv_cs): 2044 entries
constexpr int RGP_ROFF_STRIDE = 264;           // ushorts per block in roff16[] (257 offsets, padded to 16 B)
constexpr int RGP_NNZ_CAP_CPLX = 1020;         // the 16 KiB value image holds 1024 complex entries
typedef __attribute__((address_space(3))) void* lptr_t;

// One LDS-DMA piece: every active lane moves 16 bytes from its own global address to LDS at base + 16 * lane
// (base wave-uniform, in M0).  Written as inline asm on purpose: with the builtin, hipcc (ROCm 7.2) puts an
// s_waitcnt vmcnt(0) in front of every later LDS read, which would serialise the next block's stream with this
// block's gathers.  The asm is invisible to the compiler's counters, so the kernel waits for it explicitly
// (s_waitcnt vmcnt(0) + barrier before the first read of a staged block).  NT: non-temporal (streamed once).
template <int NT>
__device__ inline void lds_dma16(const void* g, void* l) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)l);
  unsigned keep;
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
__device__ inline void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// the fields of a BlkDesc the kernel needs, as scalars (copying the 32-byte struct drags its padding through a
// vector load whose wait would drain the LDS-DMA queue)
struct RgpBlk { long long base; int r0, nr, nnzb; };
__device__ inline RgpBlk rgp_desc(const BlkDesc* __restrict__ desc, long long b) {
  RgpBlk d;
  d.base = desc[b].base; d.r0 = desc[b].r0; d.nr = desc[b].nr; d.nnzb = desc[b].nnzb;
  return d;
}

__device__ inline void rgp_issue(const RgpBlk& d, const int* __restrict__ colidx, const double* __restrict__ vals,
                                 const unsigned short* __restrict__ roff16, long long b, double* sv, int* sc,
                                 unsigned short* sr) {
  if (d.nnzb > RGP_NNZ_CAP_CPLX) return;        // one over-long row: read in place
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long end = d.base + d.nnzb;
  const long long lo4 = d.base & ~3LL;
#pragma unroll
  for (int p = 0; p < 4; ++p) {            // values: 16 pieces of 1 KiB, four per wave, one complex entry per lane
    const int piece = wave + 4 * p;
    const long long idx = d.base + 64 * piece + lane;
    if (idx < end) lds_dma16<1>(vals + 2 * idx, sv + 128 * piece);
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {            // columns: 8 pieces, two per wave
    const int piece = wave + 4 * p;
    const long long idx = lo4 + 256 * piece + 4 * lane;
    if (idx < end) lds_dma16<1>(colidx + idx, sc + 256 * piece);
  }
  if (wave == 0 && 8 * lane <= d.nr) lds_dma16<0>(roff16 + b * RGP_ROFF_STRIDE + 8 * lane, sr);
}

// x, y: first column and leading dimension in doubles (contiguous column ranges; the host splits other views into
// ranges).  Per entry (ar xr - ai xi, ar xi + ai xr), each product rounded, sums in storage order - the arithmetic of
// oracle_spmv_complex.
__global__ __launch_bounds__(256) void k_spmv_rgp(
    const BlkDesc* __restrict__ desc, long long nblocks, const int* __restrict__ colidx,
    const double* __restrict__ vals, const unsigned short* __restrict__ roff16,
    const double* __restrict__ x0, long long ldx, double* __restrict__ y0, long long ldy, int k) {
  constexpr int CAP = RGP_NNZ_CAP_CPLX, CH = RG_CH;
  __shared__ __attribute__((aligned(16))) double sval[2][2048];
  __shared__ __attribute__((aligned(16))) int scol[2][2048];
  __shared__ __attribute__((aligned(16))) unsigned short sroff[2][RGP_ROFF_STRIDE + 8];
  const long long W = gridDim.x / MXG_NUM_XCD;                 // workgroups per XCD (grid is a multiple of 8)
  const long long per = (nblocks + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  const long long xcd = blockIdx.x % MXG_NUM_XCD;
  long long b = xcd * per + blockIdx.x / MXG_NUM_XCD;
  const long long last = min((xcd + 1) * per, nblocks);
  if (b >= last) return;
  const int t = threadIdx.x;
  RgpBlk d = rgp_desc(desc, b), dn = d;
  rgp_issue(d, colidx, vals, roff16, b, sval[0], scol[0], sroff[0]);
  bool has_next = b + W < last;
  if (has_next) dn = rgp_desc(desc, b + W);
  for (int it = 0; b < last; b += W, ++it) {
    const int s = it & 1;
    lds_dma_wait_all();
    __syncthreads();       // block b has landed (every wave waited for its own pieces), nobody reads the other half any more
    RgpBlk dnn = dn;
    const bool has_nn = b + 2 * W < last;
    if (has_next) {
      rgp_issue(dn, colidx, vals, roff16, b + W, sval[s ^ 1], scol[s ^ 1], sroff[s ^ 1]);
      if (has_nn) dnn = rgp_desc(desc, b + 2 * W);
    }
    if (d.nnzb > CAP) {
      for (int j = 0; j < k; ++j) {
        const double* x = x0 + j * ldx;
        double sr = 0.0, si = 0.0;
        for (long long q = d.base + t; q < d.base + d.nnzb; q += 256) {
          const int cc = colidx[q];
          const double ar = vals[2 * q], ai = vals[2 * q + 1], xr = x[2 * (long long)cc], xi = x[2 * (long long)cc + 1];
          sr += ar * xr - ai * xi;
          si += ar * xi + ai * xr;
        }
        __syncthreads();
        sval[s][t] = sr;
        sval[s][256 + t] = si;
        __syncthreads();
        if (t == 0) {
          double tr = 0.0, ti = 0.0;
          for (int q = 0; q < 256; ++q) { tr += sval[s][q]; ti += sval[s][256 + q]; }
          y0[j * ldy + 2 * (long long)d.r0] = tr;
          y0[j * ldy + 2 * (long long)d.r0 + 1] = ti;
        }
      }
    } else {
      const int* __restrict__ sc = scol[s] + (int)(d.base & 3);
      int o0 = 0, o1 = 0;
      if (t < d.nr) { o0 = sroff[s][t]; o1 = sroff[s][t + 1]; }
      const dbl2* __restrict__ sv = reinterpret_cast<const dbl2*>(sval[s]);
      for (int j = 0; j < k; ++j) {
        const dbl2* __restrict__ x = reinterpret_cast<const dbl2*>(x0 + j * ldx);
        double sr = 0.0, si = 0.0;
        for (int q0 = o0; q0 < o1; q0 += CH) {
          dbl2 v[CH], xv[CH];
          int cc[CH];
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const int q = min(q0 + i, o1 - 1);
            cc[i] = sc[q];
            v[i] = sv[q];
          }
#pragma unroll
          for (int i = 0; i < CH; ++i) xv[i] = x[cc[i]];
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const double pr = v[i].x * xv[i].x - v[i].y * xv[i].y;
            const double pi = v[i].x * xv[i].y + v[i].y * xv[i].x;
            const double r2 = sr + pr, i2 = si + pi;
            sr = q0 + i < o1 ? r2 : sr;
            si = q0 + i < o1 ? i2 : si;
          }
        }
        if (t < d.nr) {
          dbl2 out2; out2.x = sr; out2.y = si;
          __builtin_nontemporal_store(out2, reinterpret_cast<dbl2*>(y0 + j * ldy) + d.r0 + t);
        }
      }
    }
    d = dn; dn = dnn; has_next = has_nn;
  }
}


// ------------------------------------------------------------------------------------------
// Compressed-stream variant of the row-gather kernel (opt-in: mxg_csr_compress).
// The operators of the Yee lattice are structured: inside a block of 256 consecutive rows the
// column indices fall into a handful of narrow bands (one per stencil leg) and, away from cut
// cells, only a few distinct values occur.  Setup therefore re-encodes every row block as
//   segbase[16] int32          band origins,
//   vtab[nval <= 256] double   the block's distinct values (exact bit patterns),
//   code[nnz] uint16           4-bit band id | 12-bit offset  ->  col = segbase[id] + offset,
//   vid[nnz] uint8             index into vtab,
// 3 bytes per entry instead of 12.  Blocks that do not fit (more than 16 bands or 256 distinct
// values: cut-cell blocks) keep the plain CSR arrays and are walked row by row.  Values, columns
// and the summation order are exactly those of the CSR arrays: results stay bit-identical.
// ------------------------------------------------------------------------------------------
struct CsDesc {
  long long base;   // rowptr[r0]
  long long soff;   // byte offset of the block's packed stream (16-byte aligned); -1: plain CSR block
  int r0, nrnv, nnzb, rb;   // nrnv = rows (<= 256) | distinct values << 16 | flags; rb = position of the block's row offsets in roff[]
  int rec16, pad;           // length of the block's record in 16-byte units
  __host__ __device__ inline int nr() const { return nrnv & 0xffff; }
  __host__ __device__ inline int nval() const { return (nrnv >> 16) & 0x1ff; }
  // the block's record holds only its row offsets (more bands / distinct values / bytes than the packed form takes):
  // columns and values come from the CSR arrays
  __host__ __device__ inline bool roff_only() const { return (nrnv >> 30) & 1; }
};
constexpr int CS_ROFF_ONLY = 1 << 30;   // generic rows read the CSR arrays in place (row offsets are those of the CSR arrays)
static_assert(sizeof(CsDesc) == 40, "descriptor layout");
constexpr int CS_SEGS = 16, CS_OFFBITS = 12, CS_MAXVAL = 256;
// every record starts with a 32-byte header {r0, nrnv (as in CsDesc), nnzb, 0, base, 0}
constexpr int CS_HDR = 32;
constexpr int CS_MAX_RECORD = 8192;     // bytes: a block whose packed form is longer keeps the CSR arrays
__host__ __device__ inline int cs_roff_bytes(int nr) { return ((nr + 1 + 7) >> 3) * 16; }
// bytes of a block's record; packed: [header][16 band origins][value table][codes][value ids][row offsets]
__host__ __device__ inline int cs_record_bytes(int nnzb, int nval, int nr, bool roff_only) {
  if (roff_only) return CS_HDR + CS_SEGS * 4 + cs_roff_bytes(nr);
  return CS_HDR + CS_SEGS * 4 + ((nval + 1) & ~1) * 8 + ((nnzb + 7) >> 3) * 16 + ((nnzb + 15) >> 4) * 16 + cs_roff_bytes(nr);
}

typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

// EPI: 0  y = A x
//      1  y = x + omega (rhs - A x) / diag   (one damped-Jacobi sweep; rows with diag == 0 keep x)
//      2  y = rhs - A x                      (residual)
//      3  y = y + omega A x                  (coarse-grid correction in place: A = prolongation)
//      4  y = x + beta (x - y) + omega (rhs - A x) / diag   (Chebyshev smoother step, previous iterate in y)
//      5  y = (1 + beta) x + omega (rhs - A x) / diag       (the same after a zero previous iterate)
// (multigrid smoother / residual fused into the product: no separate pass over the vectors)
template <int EPI>
__device__ inline double cs_epilogue(double ax, long long row, const double* __restrict__ x,
                                     const double* __restrict__ rhs, const double* __restrict__ diag, double omega,
                                     double beta, const double* y) {
  if (EPI == 1) {
    const double dg = diag[row], xr = x[row];
    return dg != 0.0 ? xr + omega * (rhs[row] - ax) / dg : xr;
  }
  if (EPI == 5) {   // Chebyshev step after a zero previous iterate: y = (1 + beta) x + omega (rhs - A x) / diag
    const double dg = diag[row], xr = x[row];
    return dg != 0.0 ? xr + beta * xr + omega * (rhs[row] - ax) / dg : xr;
  }
  if (EPI == 4) {   // Chebyshev step, the previous iterate sits in y: y = x + beta (x - y) + omega (rhs - A x) / diag
    const double dg = diag[row], xr = x[row];
    return dg != 0.0 ? xr + beta * (xr - y[row]) + omega * (rhs[row] - ax) / dg : xr;
  }
  if (EPI == 2) return rhs[row] - ax;
  if (EPI == 3) return y[row] + omega * ax;
  return ax;
}

// RPT rows per thread (CS_RPT = 2, 128 threads per 256-row block): twice the independent gathers in flight
// per wave at the same occupancy - the packed stream leaves the kernel latency-bound, not HBM-bound
// (A/B at 256^3: 1 row/thread 0.234 ms, 2 rows 0.203 ms, 4 rows 0.313 ms).
constexpr int CS_RPT = 2;
// One column per launch (a block of k vectors is k launches: the packed stream is cheap to re-read, and
// the single-column body keeps 16 waves per CU resident).
// HALO: columns >= nowned are ghosts, read from xg[] (the receive buffer of the exchange); the blocks of a
// row-partitioned matrix that reference none are ordered first and launched without it while the halo is in flight.
template <typename CFG, int CH, bool DOT, int EPI = 0, bool HALO = false>
__global__ __launch_bounds__(CFG::SBLK / CS_RPT) void k_spmv_cs(
    const CsDesc* __restrict__ desc, long long nblocks, const unsigned char* __restrict__ stream,
    const unsigned short* __restrict__ roff, const int* __restrict__ colidx, const double* __restrict__ vals,
    const double* __restrict__ x, double* __restrict__ y,
    const double* __restrict__ xdot, double* __restrict__ partial,
    const double* __restrict__ rhs, const double* __restrict__ diag, double omega,
    const double* __restrict__ xg, long long nowned, double beta) {
  auto xat = [&](int cc) -> double { return (HALO && cc >= nowned) ? xg[cc - nowned] : x[cc]; };
  constexpr int RPT = CS_RPT, SB = CFG::SBLK / RPT;
  __shared__ __attribute__((aligned(16))) unsigned short scode[CFG::NNZ_BLK];
  __shared__ __attribute__((aligned(16))) unsigned char svid[CFG::NNZ_BLK];
  __shared__ double svtab[CS_MAXVAL];
  __shared__ int sseg[CS_SEGS];
  __shared__ double sred[SB / 64];
  const long long b = xcd_block(blockIdx.x, nblocks);
  if (b >= nblocks) return;
  const CsDesc d = desc[b];
  const int t = threadIdx.x;
  if (d.nnzb > CFG::NNZ_CAP) {     // one over-long row: whole workgroup, fixed-order combine (as k_spmv_rowgather)
    double sr = 0.0;
#pragma unroll 1
    for (long long q = d.base + t; q < d.base + d.nnzb; q += SB) sr += vals[q] * xat(colidx[q]);
    svtab[t] = sr;
    __syncthreads();
    if (t == 0) {
      double tr = 0.0;
#pragma unroll 4
      for (int q = 0; q < SB; ++q) tr += svtab[q];   // (full unrolling of this cold loop would cost 256 VGPRs)
      y[d.r0] = cs_epilogue<EPI>(tr, d.r0, x, rhs, diag, omega, beta, y);
      if (DOT) partial[b] = tr * xdot[d.r0];
    }
    return;
  }
  const bool packed = d.soff >= 0 && !d.roff_only();
  if (packed) {
    const unsigned char* sp = stream + d.soff + CS_HDR;
    if (t < CS_SEGS) sseg[t] = nt_load(reinterpret_cast<const int*>(sp) + t);
    const double* vt = reinterpret_cast<const double*>(sp + CS_SEGS * 4);
#pragma unroll 1
    for (int p = t; p < d.nval(); p += SB) svtab[p] = nt_load(vt + p);
    const int nvpad = (d.nval() + 1) & ~1;
    const uint4v* cp = reinterpret_cast<const uint4v*>(sp + CS_SEGS * 4 + nvpad * 8);
    const int ncode16 = (d.nnzb + 7) >> 3;            // 16-byte pieces of the code array (8 codes each)
#pragma unroll 1
    for (int p = t; p < ncode16; p += SB) reinterpret_cast<uint4v*>(scode)[p] = nt_load(cp + p);
    const uint4v* ip = cp + ncode16;
    const int nid16 = (d.nnzb + 15) >> 4;
    if (t < nid16) reinterpret_cast<uint4v*>(svid)[t] = nt_load(ip + t);   // <= 128 pieces
  }
  int o0[RPT], o1[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int row = t + r * SB;
    o0[r] = 0; o1[r] = 0;
    if (row < d.nr()) {
      const unsigned short* ro = roff + d.rb + row;
      o0[r] = ro[0];
      o1[r] = ro[1];
    }
  }
  __syncthreads();
  double sum[RPT];
  if (packed) {
    // first CH entries of every row: all gathers of the thread's rows are issued before any product
    double v[RPT][CH], xv[RPT][CH];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      sum[r] = 0.0;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const bool on = o0[r] + i < o1[r];
        const unsigned code = on ? scode[o0[r] + i] : 0u;
        const int cc = on ? sseg[code >> CS_OFFBITS] + (int)(code & ((1u << CS_OFFBITS) - 1u)) : 0;
        v[r][i] = on ? svtab[svid[o0[r] + i]] : 0.0;
        xv[r][i] = on ? xat(cc) : 0.0;
      }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
#pragma unroll
      for (int i = 0; i < CH; ++i)
        if (o0[r] + i < o1[r]) sum[r] += v[r][i] * xv[r][i];
#pragma unroll 1
      for (int q = o0[r] + CH; q < o1[r]; ++q) {       // longer rows: the tail, in storage order
        const unsigned code = scode[q];
        sum[r] += svtab[svid[q]] * xat(sseg[code >> CS_OFFBITS] + (int)(code & ((1u << CS_OFFBITS) - 1u)));
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      sum[r] = 0.0;
#pragma unroll 1
      for (long long q = d.base + o0[r]; q < d.base + o1[r]; ++q) sum[r] += vals[q] * xat(colidx[q]);
    }
  }
  double part = 0.0;
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int row = t + r * SB;
    if (row < d.nr()) {
      __builtin_nontemporal_store(cs_epilogue<EPI>(sum[r], d.r0 + row, x, rhs, diag, omega, beta, y), y + d.r0 + row);
      if (DOT) part += sum[r] * xdot[d.r0 + row];
    }
  }
  if (DOT) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if ((t & 63) == 0) sred[t >> 6] = part;
    __syncthreads();
    if (t == 0) {
      double tot = 0.0;
      for (int w = 0; w < SB / 64; ++w) tot += sred[w];
      partial[b] = tot;
    }
  }
}

// out[0] = sum of partial[0..n) in index order (single workgroup, fixed tree: reproducible)
__global__ void k_sum_partials(const double* __restrict__ partial, long long n, double* __restrict__ out) {
  __shared__ double lds[1024];
  double s = 0.0;
  long long i = threadIdx.x;
  for (; i + 7 * 1024 < n; i += 8 * 1024) {        // eight loads in flight, added in index order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < n; i += 1024) s += partial[i];
  lds[threadIdx.x] = s;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) lds[threadIdx.x] += lds[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = lds[0];
}

// greedy row blocks: <= nnz_cap non-zeros and <= rows_cap rows; an over-long row stands alone
std::vector<int32_t> build_row_blocks(const int64_t* rowptr, int64_t nrows, int nnz_cap, int rows_cap) {
  std::vector<int32_t> blk;
  blk.push_back(0);
  int64_t r = 0;
  while (r < nrows) {
    int64_t e = r;
    const int64_t start = rowptr[r];
    while (e < nrows && e - r < rows_cap && rowptr[e + 1] - start <= nnz_cap) ++e;
    if (e == r) e = r + 1;  // single long row
    blk.push_back((int32_t)e);
    r = e;
  }
  return blk;
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

template <bool CPLX, typename CFG, bool NT, bool XCD>
void launch_stream(mxg_ctx* c, const mxg_csr* a, const MvDesc& X, const MvDesc& Y) {
  const long long per = (a->nblocks + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  const unsigned grid = XCD ? (unsigned)(per * MXG_NUM_XCD) : (unsigned)a->nblocks;
  hipLaunchKernelGGL((k_spmv_stream<CPLX, CFG, NT, XCD>), dim3(grid), dim3(CFG::SBLK), 0, c->stream,
                     (const long long*)a->d_rowptr, a->d_colidx, a->d_vals, a->d_rowblk,
                     (long long)a->nblocks, X, Y);
}

}  // namespace

static int build_pipe_desc(mxg_csr* a, int geo, const int64_t* host_rowptr);
// a->kernel: which CSR kernel a product runs (mxg_csr_set_kernel) and whether a compressed form is active.
//   default          real: k_spmv_rowgather with 16-bit in-block row offsets; complex: k_spmv_rgp (persistent, LDS-DMA)
//   KF_ROWPTR64      real: the row-gather kernel reads the int64 row pointers of the CSR arrays (8 B/row instead of 2)
//   KF_STREAM        complex: one block per workgroup (k_spmv_stream)
//   KF_COMPRESSED    mxg_csr_compress has run: products use the tile form or the compressed stream
constexpr int KF_ROWPTR64 = 1, KF_STREAM = 2, KF_COMPRESSED = 4;

extern "C" int mxg_csr_create(mxg_ctx* c, int64_t nrows, int64_t ncols_owned, int64_t nghost,
                              const int64_t* rowptr, const int32_t* colidx, const double* vals,
                              int is_complex, const mxg_halo_plan* plan, mxg_csr** out) {
  MXG_CHECK(c && rowptr && out, "mxg_csr_create: NULL argument");
  MxgSetupTrace trace("mxg_csr_create (nnz)", rowptr[nrows]);
  MXG_CHECK(nrows >= 0 && ncols_owned >= 0 && nghost >= 0, "mxg_csr_create: negative size");
  MXG_CHECK(nrows < (int64_t)1 << 31 && ncols_owned + nghost < (int64_t)1 << 31,
            "mxg_csr_create: local sizes must fit int32 column / row-block indices");
  MXG_CHECK(rowptr[0] == 0, "mxg_csr_create: rowptr[0] must be 0");
  MXG_CHECK(nghost == 0 || plan != nullptr, "mxg_csr_create: ghosts without a halo plan");
  const int64_t nnz = rowptr[nrows];
  const int64_t ncols = ncols_owned + nghost;
  {
    // validation on host threads (a serial pass over the 1.5e8 entries of a 256^3 operator cost 0.1 s per matrix);
    // the first offending row / entry in index order is the one reported
    unsigned nthreads = std::thread::hardware_concurrency();
    if (const char* e = getenv("OMP_NUM_THREADS")) { const int v = atoi(e); if (v > 0) nthreads = (unsigned)v; }
    if (nthreads > 32) nthreads = 32;
    if (nthreads < 1 || nnz < (int64_t)1 << 20) nthreads = 1;
    std::vector<int64_t> bad_row(nthreads, -1), bad_ent(nthreads, -1);
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < nthreads; ++i)
      pool.emplace_back([&, i] {
        for (int64_t r = nrows * i / nthreads; r < nrows * (i + 1) / nthreads; ++r)
          if (rowptr[r + 1] < rowptr[r]) { bad_row[i] = r; break; }
        for (int64_t q = nnz * i / nthreads; q < nnz * (i + 1) / nthreads; ++q)
          if (colidx[q] < 0 || colidx[q] >= ncols) { bad_ent[i] = q; break; }
      });
    for (auto& th : pool) th.join();
    for (unsigned i = 0; i < nthreads; ++i)
      MXG_CHECK(bad_row[i] < 0, "mxg_csr_create: rowptr not monotone at row %lld", (long long)bad_row[i]);
    for (unsigned i = 0; i < nthreads; ++i)
      MXG_CHECK(bad_ent[i] < 0, "mxg_csr_create: column index %d out of range [0,%lld) at entry %lld", colidx[bad_ent[i]],
                (long long)ncols, (long long)bad_ent[i]);
  }
  MXG_HIP(hipSetDevice(c->device));
  mxg_csr* a = new mxg_csr();
  a->ctx = c;
  a->nrows = nrows; a->ncols_owned = ncols_owned; a->nghost = nghost; a->nnz = nnz;
  a->is_complex = is_complex ? 1 : 0;
  const int w = is_complex ? 2 : 1;

  std::vector<int32_t> blk = build_row_blocks(rowptr, nrows, CfgDefault::NNZ_CAP, CfgDefault::ROWS_BLK);
  a->nblocks = (int64_t)blk.size() - 1;

  // +8 entries of slack: the kernels load aligned pairs / 16-byte pieces and may touch a few elements past nnz
  MXG_HIP(hipMalloc(&a->d_rowptr, sizeof(int64_t) * (nrows + 1)));
  MXG_HIP(hipMalloc(&a->d_colidx, sizeof(int32_t) * (nnz + 8)));
  MXG_HIP(hipMalloc(&a->d_vals, sizeof(double) * (nnz + 8) * w));
  MXG_HIP(hipMalloc(&a->d_rowblk, sizeof(int32_t) * blk.size()));
  MXG_HIP(hipMemsetAsync(a->d_colidx, 0, sizeof(int32_t) * (nnz + 8), c->stream));
  MXG_HIP(hipMemsetAsync(a->d_vals, 0, sizeof(double) * (nnz + 8) * w, c->stream));
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
  if (!a->is_complex && nnz > 0) {
    // production kernel: row-gather, 2048-nnz blocks, gather chunk 7 (measured best on the
    // 7-point bulk rows of the vector Laplacian; 70.8 % of HBM peak at 256^3)
    MXG_TRY(build_pipe_desc(a, 10, rowptr));
  } else if (a->is_complex && nnz > 0) {
    // complex scalars (Bloch phases, lossy dielectrics): the persistent row-gather kernel over 16-byte values
    MXG_TRY(build_pipe_desc(a, 12, rowptr));
  } else if (!a->is_complex) {
    // a rank of a decomposed grid that owns nothing: same kernel family as everybody else (zero row blocks), so that
    // every rank takes the same solver path and issues the same sequence of collectives
    a->pipe_geo = 10;
    a->npipe = 0;
  }
  a->kernel = 0;
  *out = a;
  return 0;
}

// A matrix born on the device (mxg_spalg.hip): take ownership of finished CSR arrays (rowptr nrows + 1, colidx / vals
// with 8 entries of slack, columns ascending inside a row) and build what mxg_csr_create builds around them.
int mxg_csr_adopt_device(mxg_ctx* c, int64_t nrows, int64_t ncols, int64_t nnz, int64_t* d_rowptr, int32_t* d_colidx,
                         double* d_vals, int is_complex, mxg_csr** out) {
  MXG_CHECK(nrows < (int64_t)1 << 31 && ncols < (int64_t)1 << 31, "mxg_csr_adopt_device: sizes must fit int32 indices");
  mxg_csr* a = new mxg_csr();
  a->ctx = c;
  a->nrows = nrows; a->ncols_owned = ncols; a->nghost = 0; a->nnz = nnz;
  a->is_complex = is_complex ? 1 : 0;
  a->d_rowptr = d_rowptr; a->d_colidx = d_colidx; a->d_vals = d_vals;
  a->kernel = 0;
  a->lazy = 1;
  *out = a;
  if (getenv("MXG_LAZY_ADOPT") && atoi(getenv("MXG_LAZY_ADOPT")) == 0) return mxg_csr_materialize(a);
  return 0;
}

int mxg_csr_materialize(mxg_csr* a) {
  if (!a || !a->lazy) return 0;
  MxgSetupTrace trace("mxg_csr_materialize (nnz)", a->nnz);
  a->lazy = 0;
  mxg_ctx* c = a->ctx;
  const int64_t nrows = a->nrows, nnz = a->nnz;
  std::vector<int64_t> rp((size_t)nrows + 1);
  MXG_HIP(hipMemcpyAsync(rp.data(), a->d_rowptr, sizeof(int64_t) * (nrows + 1), hipMemcpyDeviceToHost, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  std::vector<int32_t> blk = build_row_blocks(rp.data(), nrows, CfgDefault::NNZ_CAP, CfgDefault::ROWS_BLK);
  a->nblocks = (int64_t)blk.size() - 1;
  MXG_HIP(hipMalloc(&a->d_rowblk, sizeof(int32_t) * blk.size()));
  MXG_HIP(hipMemcpyAsync(a->d_rowblk, blk.data(), sizeof(int32_t) * blk.size(), hipMemcpyHostToDevice, c->stream));
  const int w = a->is_complex ? 2 : 1;
  MXG_HIP(hipMalloc(&a->d_diag, sizeof(double) * std::max<int64_t>(nrows, 1) * w));
  if (nrows > 0) {
    hipLaunchKernelGGL(k_diag, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream,
                       (const long long*)a->d_rowptr, a->d_colidx, a->d_vals, (long long)nrows, w, a->d_diag);
    MXG_HIP(hipGetLastError());
  }
  MXG_HIP(hipStreamSynchronize(c->stream));
  if (nnz > 0) MXG_TRY(build_pipe_desc(a, a->is_complex ? 12 : 10, rp.data()));   // as mxg_csr_create
  else if (!a->is_complex) { a->pipe_geo = 10; a->npipe = 0; }
  return 0;
}

extern "C" int mxg_csr_destroy(mxg_csr* a) {
  if (!a) return 0;
  if (!mxg_ctx_alive(a->ctx)) { delete a; return 0; }   // context already gone
  hipSetDevice(a->ctx->device);
  hipStreamSynchronize(a->ctx->stream);
  if (a->ctx->comm_stream) hipStreamSynchronize(a->ctx->comm_stream);
  hipFree(a->d_rowptr); hipFree(a->d_colidx); hipFree(a->d_vals); hipFree(a->d_rowblk);
  hipFree(a->d_diag);
  if (a->d_blkdesc) hipFree(a->d_blkdesc);
  if (a->d_dotpartial) hipFree(a->d_dotpartial);
  if (a->d_roff) hipFree(a->d_roff);
  if (a->d_roff16) hipFree(a->d_roff16);
  mxg_tile_free(a);
  if (a->d_injcol) hipFree(a->d_injcol);
  if (a->d_ellcol) hipFree(a->d_ellcol);
  if (a->d_ellval) hipFree(a->d_ellval);
  if (a->d_injval) hipFree(a->d_injval);
  if (a->d_csdesc) hipFree(a->d_csdesc);
  if (a->d_csstream) hipFree(a->d_csstream);
  if (a->d_send_idx) hipFree(a->d_send_idx);
  if (a->d_sendbuf) hipFree(a->d_sendbuf);
  if (a->d_xext) hipFree(a->d_xext);
  if (a->d_ghost) hipFree(a->d_ghost);
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

// Row blocks of the row-gather kernels: 32-byte descriptors, 16-bit in-block row offsets (geo 10: real, <= 2044
// entries per block; geo 12: complex, <= 1020), ghost-free blocks first on a row-partitioned matrix.
static int build_pipe_desc(mxg_csr* a, int geo, const int64_t* host_rowptr) {
  mxg_ctx* c = a->ctx;
  std::vector<int64_t> rp_copy;
  if (!host_rowptr) {
    rp_copy.resize(a->nrows + 1);
    MXG_HIP(hipMemcpyAsync(rp_copy.data(), a->d_rowptr, sizeof(int64_t) * (a->nrows + 1),
                           hipMemcpyDeviceToHost, c->stream));
    MXG_HIP(hipStreamSynchronize(c->stream));
    host_rowptr = rp_copy.data();
  }
  const int cap = geo == 12 ? RGP_NNZ_CAP_CPLX : RGP_NNZ_CAP;
  const int rows_cap = RgCfgA::ROWS_CAP;
  std::vector<int32_t> blk = build_row_blocks(host_rowptr, a->nrows, cap, rows_cap);
  std::vector<BlkDesc> desc(blk.size() - 1);
  for (size_t b = 0; b + 1 < blk.size(); ++b) {
    BlkDesc d{};
    d.r0 = blk[b]; d.nr = blk[b + 1] - blk[b];
    d.base = host_rowptr[blk[b]];
    d.nnzb = (int)(host_rowptr[blk[b + 1]] - d.base);
    d.rb = d.r0 + (int)b;
    desc[b] = d;
  }
  if (a->d_blkdesc) hipFree(a->d_blkdesc);
  if (a->d_dotpartial) { hipFree(a->d_dotpartial); a->d_dotpartial = nullptr; a->dotpartial_cap = 0; }
  if (a->d_roff) { hipFree(a->d_roff); a->d_roff = nullptr; }
  {
    // row-gather kernel: in-block row offsets as uint16 (nr + 1 per block, block b at r0 + b);
    // blocks holding one over-long row take the whole-workgroup path and never read them
    std::vector<unsigned short> roff((size_t)a->nrows + desc.size() + 1, 0);
    for (size_t b = 0; b < desc.size(); ++b) {
      const BlkDesc& d = desc[b];
      if (d.nnzb > cap) continue;
      for (int r = 0; r <= d.nr; ++r) roff[(size_t)d.rb + r] = (unsigned short)(host_rowptr[d.r0 + r] - d.base);
    }
    MXG_HIP(hipMalloc(&a->d_roff, sizeof(unsigned short) * roff.size()));
    MXG_HIP(hipMemcpyAsync(a->d_roff, roff.data(), sizeof(unsigned short) * roff.size(), hipMemcpyHostToDevice,
                           c->stream));
    MXG_HIP(hipStreamSynchronize(c->stream));
  }
  a->n_interior = (int64_t)desc.size();
  if (geo == 10 && a->nghost > 0) {
    // distributed matrix: blocks that touch no ghost column first - they run while the halo is in flight
    std::vector<int32_t> ci((size_t)a->nnz);
    if (a->nnz > 0)
      MXG_HIP(hipMemcpy(ci.data(), a->d_colidx, sizeof(int32_t) * a->nnz, hipMemcpyDeviceToHost));
    std::vector<BlkDesc> inner, outer;
    for (const BlkDesc& d : desc) {
      bool ghost = false;
      for (long long q = d.base; q < d.base + d.nnzb && !ghost; ++q) ghost = ci[(size_t)q] >= a->ncols_owned;
      (ghost ? outer : inner).push_back(d);
    }
    a->n_interior = (int64_t)inner.size();
    desc = inner;
    desc.insert(desc.end(), outer.begin(), outer.end());
  }
  MXG_HIP(hipMalloc(&a->d_blkdesc, sizeof(BlkDesc) * std::max<size_t>(desc.size(), 1)));
  if (!desc.empty())
    MXG_HIP(hipMemcpyAsync(a->d_blkdesc, desc.data(), sizeof(BlkDesc) * desc.size(),
                           hipMemcpyHostToDevice, c->stream));
  if (a->d_roff16) { hipFree(a->d_roff16); a->d_roff16 = nullptr; }
  if (geo == 12) {
    // persistent kernel: the same offsets, RGP_ROFF_STRIDE ushorts per block in descriptor order, so that a block's
    // offsets are one 16-byte aligned LDS-DMA piece
    std::vector<unsigned short> r16(desc.size() * (size_t)RGP_ROFF_STRIDE + 8, 0);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < (long long)desc.size(); ++b) {
      const BlkDesc& d = desc[(size_t)b];
      if (d.nnzb > cap) continue;
      for (int r = 0; r <= d.nr; ++r) r16[(size_t)b * RGP_ROFF_STRIDE + r] = (unsigned short)(host_rowptr[d.r0 + r] - d.base);
    }
    MXG_HIP(hipMalloc(&a->d_roff16, sizeof(unsigned short) * r16.size()));
    MXG_HIP(hipMemcpyAsync(a->d_roff16, r16.data(), sizeof(unsigned short) * r16.size(), hipMemcpyHostToDevice, c->stream));
  }
  MXG_HIP(hipStreamSynchronize(c->stream));
  a->npipe = (int64_t)desc.size();
  a->pipe_geo = geo;
  return 0;
}


// Product kernel of compressed matrices: wavefront-uniform tile patterns where they cover the matrix (lattice
// operators; k_spmv_tile, mxg_tile.hip), otherwise - and with MXG_PAT=0 at compress time - the compressed stream (k_spmv_cs).
extern "C" int mxg_csr_product_form(const mxg_csr* a, int64_t* tiles, int64_t* patterns, int64_t* irregular_rows, double* coverage) {
  if (!a) return -1;
  const bool tile = (a->kernel & KF_COMPRESSED) != 0 && a->tile_on;
  if (tiles) *tiles = tile ? a->pat_ntile : 0;
  if (patterns) *patterns = tile ? a->pat_ntilepat : 0;
  if (irregular_rows) *irregular_rows = tile ? a->pat_ngrows : 0;
  if (coverage) *coverage = a->pat_tile_coverage;
  if (tile) return 3;
  return (a->kernel & KF_COMPRESSED) != 0 && a->ncs > 0 ? 1 : 0;
}

static bool tile_wanted() { const char* e = getenv("MXG_PAT"); return !e || atoi(e) != 0; }
static inline bool tile_active(const mxg_csr* a) { return (a->kernel & KF_COMPRESSED) != 0 && a->tile_on; }
static inline bool cs_active(const mxg_csr* a) { return (a->kernel & KF_COMPRESSED) != 0 && !a->tile_on && a->ncs > 0 && !a->is_complex; }

// Records for the compressed-stream kernel over the row blocks of the row-gather geometry (RgCfgA):
//   [header 32 B][16 band origins][value table][codes][value ids][row offsets]
static int build_cs(mxg_csr* a, const int64_t* rp, const int32_t* ci, const double* va) {
  mxg_ctx* c = a->ctx;
  std::vector<int32_t> blk = build_row_blocks(rp, a->nrows, RgCfgA::NNZ_CAP, RgCfgA::ROWS_CAP);
  const size_t nb = blk.size() - 1;
  std::vector<CsDesc> desc(nb);
  std::vector<std::vector<unsigned char> > packs(nb);
  auto work = [&](size_t b0, size_t b1) {
    std::vector<int32_t> u;
    std::vector<uint64_t> w;
    for (size_t b = b0; b < b1; ++b) {
      CsDesc d{};
      d.r0 = blk[b]; d.nrnv = blk[b + 1] - blk[b];
      d.base = rp[blk[b]];
      d.nnzb = (int)(rp[blk[b + 1]] - d.base);
      d.soff = 0;           // final offset assigned below
      d.rb = d.r0 + (int)b;
      const int nr = d.nrnv;
      std::vector<unsigned char>& pk = packs[b];
      int flags = 0, nval = 0;
      auto finish = [&](int nnz_generic) {                   // header + descriptor once the record is complete
        d.nrnv = nr | (nval << 16) | flags;
        int32_t h[8] = {d.r0, d.nrnv, nnz_generic, 0, 0, 0, 0, 0};
        std::memcpy(h + 4, &d.base, sizeof d.base);
        std::memcpy(pk.data(), h, sizeof h);
        d.rec16 = (int)(pk.size() / 16);
        desc[b] = d;
      };
      if (d.nnzb > RgCfgA::NNZ_CAP) {                       // one over-long row: header only, the row is read in place
        pk.assign(CS_HDR, 0);
        finish(d.nnzb);
        continue;
      }
      const int ng = d.nnzb;
      const int32_t* gcol = ci + d.base;
      const double* gval = va + d.base;
      // rows read from the CSR arrays in place (more bands / distinct values / bytes than the packed form takes):
      // the record carries the row offsets of the CSR arrays themselves
      auto csr_rows = [&]() {
        pk.assign((size_t)CS_HDR + CS_SEGS * 4 + (size_t)cs_roff_bytes(nr), 0);
        unsigned short* ro = reinterpret_cast<unsigned short*>(pk.data() + CS_HDR + CS_SEGS * 4);
        for (int r = 0; r <= nr; ++r) ro[r] = (unsigned short)(rp[d.r0 + r] - d.base);
        flags |= CS_ROFF_ONLY;
        nval = 0;
        finish(d.nnzb);
      };
      if (ng == 0) { csr_rows(); continue; }                 // (an empty block: offsets only)
      // column bands: greedy cover of the sorted distinct columns with windows of 2^12
      u.assign(gcol, gcol + ng);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      int32_t seg[CS_SEGS];
      int nseg = 0;
      bool ok = true;
      for (size_t i = 0; i < u.size();) {
        if (nseg == CS_SEGS) { ok = false; break; }
        seg[nseg++] = u[i];
        const int64_t lim = (int64_t)u[i] + (1 << CS_OFFBITS);
        while (i < u.size() && u[i] < lim) ++i;
      }
      if (!ok) { csr_rows(); continue; }
      // distinct values by bit pattern
      w.resize((size_t)ng);
      std::memcpy(w.data(), gval, sizeof(double) * (size_t)ng);
      std::sort(w.begin(), w.end());
      w.erase(std::unique(w.begin(), w.end()), w.end());
      if (w.size() > (size_t)CS_MAXVAL) { csr_rows(); continue; }
      nval = (int)w.size();
      const int nvpad = (nval + 1) & ~1;
      const size_t ncode16 = ((size_t)ng + 7) >> 3, nid16 = ((size_t)ng + 15) >> 4;
      const size_t nbytes = (size_t)cs_record_bytes(ng, nval, nr, false);
      if (nbytes > (size_t)CS_MAX_RECORD) { csr_rows(); continue; }
      pk.assign(nbytes, 0);
      unsigned char* g = pk.data() + CS_HDR;
      int32_t segpad[CS_SEGS];
      for (int i = 0; i < CS_SEGS; ++i) segpad[i] = i < nseg ? seg[i] : 0;
      std::memcpy(g, segpad, sizeof segpad);
      std::memcpy(g + CS_SEGS * 4, w.data(), sizeof(uint64_t) * (size_t)nval);
      unsigned short* code = reinterpret_cast<unsigned short*>(g + CS_SEGS * 4 + (size_t)nvpad * 8);
      unsigned char* vid = g + CS_SEGS * 4 + (size_t)nvpad * 8 + ncode16 * 16;
      unsigned short* ro = reinterpret_cast<unsigned short*>(vid + nid16 * 16);
      for (int q = 0; q < ng; ++q) {
        const int32_t col = gcol[q];
        const int sidx = (int)(std::upper_bound(seg, seg + nseg, col) - seg) - 1;
        code[q] = (unsigned short)((sidx << CS_OFFBITS) | (col - seg[sidx]));
        uint64_t bits;
        std::memcpy(&bits, &gval[q], sizeof bits);
        vid[q] = (unsigned char)(std::lower_bound(w.begin(), w.end(), bits) - w.begin());
      }
      for (int r = 0; r <= nr; ++r) ro[r] = (unsigned short)(rp[d.r0 + r] - d.base);
      finish(ng);
    }
  };
  unsigned nthreads = std::thread::hardware_concurrency();
  if (const char* e = getenv("OMP_NUM_THREADS")) { const int v = atoi(e); if (v > 0) nthreads = (unsigned)v; }
  if (nthreads > 32) nthreads = 32;
  if (nthreads < 1 || nb < 64) nthreads = 1;
  std::vector<std::thread> pool;
  for (unsigned i = 0; i < nthreads; ++i)
    pool.emplace_back(work, nb * i / nthreads, nb * (i + 1) / nthreads);
  for (auto& th : pool) th.join();
  size_t total = 0, nfall = 0;
  for (size_t b = 0; b < nb; ++b) {
    if (desc[b].nnzb > RgCfgA::NNZ_CAP || (desc[b].roff_only() && desc[b].nnzb > 0)) ++nfall;
    desc[b].soff = (long long)total;
    total += packs[b].size();
  }
  std::vector<unsigned char> stream(total + 64, 0);
  for (size_t b = 0; b < nb; ++b)
    if (desc[b].soff >= 0) std::memcpy(stream.data() + desc[b].soff, packs[b].data(), packs[b].size());
  a->ncs_interior = (int64_t)nb;
  if (a->nghost > 0) {
    // row-partitioned matrix: blocks that touch no ghost column first - they run while the halo is in flight
    std::vector<CsDesc> inner, outer;
    for (const CsDesc& d : desc) {
      bool ghost = false;
      for (long long q = d.base; q < d.base + d.nnzb && !ghost; ++q) ghost = ci[q] >= a->ncols_owned;
      (ghost ? outer : inner).push_back(d);
    }
    a->ncs_interior = (int64_t)inner.size();
    desc = inner;
    desc.insert(desc.end(), outer.begin(), outer.end());
  }
  if (a->d_csdesc) hipFree(a->d_csdesc);
  if (a->d_csstream) hipFree(a->d_csstream);
  MXG_HIP(hipMalloc(&a->d_csdesc, sizeof(CsDesc) * std::max<size_t>(nb, 1)));
  MXG_HIP(hipMalloc(&a->d_csstream, stream.size()));
  if (nb) MXG_HIP(hipMemcpyAsync(a->d_csdesc, desc.data(), sizeof(CsDesc) * nb, hipMemcpyHostToDevice, c->stream));
  MXG_HIP(hipMemcpyAsync(a->d_csstream, stream.data(), stream.size(), hipMemcpyHostToDevice, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  a->ncs = (int64_t)nb;
  a->cs_fallback = (int64_t)nfall;
  a->cs_bytes = (int64_t)total;
  return 0;
}

extern "C" int mxg_csr_compress_from(mxg_csr* a, const int64_t* rowptr, const int32_t* colidx, const double* vals,
                                     int64_t* nblocks, int64_t* nfallback, int64_t* stream_bytes) {
  MXG_CHECK(a, "mxg_csr_compress: NULL argument");
  MxgSetupTrace trace(rowptr ? "mxg_csr_compress_from (nnz)" : "mxg_csr_compress (nnz)", a->nnz);
  MXG_TRY(mxg_csr_materialize(a));
  mxg_ctx* c = a->ctx;
  MXG_HIP(hipSetDevice(c->device));
  const bool want_tile = tile_wanted();
  // (re)build when there is no form yet, or the form at hand is not the one MXG_PAT asks for now
  bool have;
  if (a->tile_on) have = want_tile;
  else if (a->is_complex) have = a->tile_tried || !want_tile;       // complex matrices have no compressed stream
  else have = a->ncs > 0 && (!want_tile || a->tile_tried);
  if (!have && a->nrows > 0) {
    std::vector<int64_t> rp;
    std::vector<int32_t> ci;
    std::vector<double> va;
    const int w = a->is_complex ? 2 : 1;
    if (!rowptr || !colidx || !vals) {       // no host copy offered: read the arrays back
      rp.resize((size_t)a->nrows + 1); ci.resize((size_t)a->nnz + 1); va.resize(((size_t)a->nnz + 1) * w);
      MXG_HIP(hipMemcpyAsync(rp.data(), a->d_rowptr, sizeof(int64_t) * (a->nrows + 1), hipMemcpyDeviceToHost, c->stream));
      if (a->nnz > 0) {
        MXG_HIP(hipMemcpyAsync(ci.data(), a->d_colidx, sizeof(int32_t) * a->nnz, hipMemcpyDeviceToHost, c->stream));
        MXG_HIP(hipMemcpyAsync(va.data(), a->d_vals, sizeof(double) * a->nnz * w, hipMemcpyDeviceToHost, c->stream));
      }
      MXG_HIP(hipStreamSynchronize(c->stream));
      rowptr = rp.data(); colidx = ci.data(); vals = va.data();
    }
    // the tile form first: a lattice operator needs no compressed stream at all - its encoding pass (0.6 s for a
    // 256^3 operator) and 0.5 GB of device memory are saved
    mxg_tile_free(a);
    a->tile_tried = 0;
    if (want_tile) { MXG_TRY(mxg_tile_build(a, rowptr, colidx, vals)); a->tile_tried = 1; }
    if (!a->tile_on && !a->is_complex && a->ncs == 0) MXG_TRY(build_cs(a, rowptr, colidx, vals));
  }
  a->kernel = (a->kernel & ~KF_COMPRESSED) | ((a->tile_on || (a->ncs > 0 && !a->is_complex)) ? KF_COMPRESSED : 0);
  if (a->tile_on) {                   // tile form: work units, no fall-back blocks, bytes of the form
    if (nblocks) *nblocks = a->pat_ntile + a->gen_ngrp + a->gen_nlong;     // tiles + groups + long rows
    if (nfallback) *nfallback = 0;
    if (stream_bytes) *stream_bytes = a->tile_bytes;
    return 0;
  }
  if (nblocks) *nblocks = a->ncs;
  if (nfallback) *nfallback = a->cs_fallback;
  if (stream_bytes) *stream_bytes = a->cs_bytes;
  return 0;
}

extern "C" const char* mxg_csr_kernel_name(const mxg_csr* a) {
  if (!a) return "";
  if (tile_active(a)) return a->is_complex ? "k_spmv_tile<complex>" : "k_spmv_tile";
  if (cs_active(a)) return "k_spmv_cs";
  if (a->is_complex) return (a->kernel & KF_STREAM) ? "k_spmv_stream<complex>" : "k_spmv_rgp";
  return (a->kernel & KF_ROWPTR64) ? "k_spmv_rowgather (int64 row pointers)" : "k_spmv_rowgather (16-bit in-block row offsets)";
}

extern "C" int mxg_csr_compress(mxg_csr* a, int64_t* nblocks, int64_t* nfallback, int64_t* stream_bytes) {
  return mxg_csr_compress_from(a, nullptr, nullptr, nullptr, nblocks, nfallback, stream_bytes);
}

// which: 0 default CSR kernels, 1 = KF_ROWPTR64, 2 = KF_STREAM (see above).  Always leaves the compressed form
// inactive (the CSR arrays are what is multiplied); mxg_csr_compress switches it on again.
extern "C" int mxg_csr_set_kernel(mxg_csr* a, int which) {
  MXG_CHECK(a, "mxg_csr_set_kernel: NULL argument");
  MXG_CHECK(which >= 0 && which <= 2, "mxg_csr_set_kernel: unknown kernel code %d (0 default, 1 int64 row pointers, 2 complex stream kernel)", which);
  MXG_TRY(mxg_csr_materialize(a));
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
  if (k > a->halo_kcap || !a->d_xext) {
    MXG_HIP(hipStreamSynchronize(c->stream));
    MXG_HIP(hipStreamSynchronize(c->comm_stream));
    if (a->d_sendbuf) hipFree(a->d_sendbuf);
    if (a->d_xext) hipFree(a->d_xext);
    if (a->d_ghost) { hipFree(a->d_ghost); a->d_ghost = nullptr; }
    MXG_HIP(hipMalloc(&a->d_sendbuf, sizeof(double) * std::max<int64_t>(a->nsend, 1) * w * k));
    MXG_HIP(hipMalloc(&a->d_xext, sizeof(double) * ldx * k));
    a->halo_kcap = k;   // (d_ghost is re-created on demand by halo_start)
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
  MXG_CHECK(mxg_comm_on(c), "halo exchange needs a communicator (nranks > 1)");
  std::vector<MxgXfer> xf;
  for (int j = 0; j < k; ++j)
    for (int p = 0; p < a->nneigh; ++p)
      xf.push_back({a->neigh[p], a->d_sendbuf + (a->nsend * j + a->send_off[p]) * w, (a->send_off[p + 1] - a->send_off[p]) * w,
                    a->d_xext + ldx * j + (a->ncols_owned + a->recv_off[p]) * w, (a->recv_off[p + 1] - a->recv_off[p]) * w});
  MXG_TRY(mxg_comm_exchange(c, xf.data(), (int)xf.size(), c->stream));
  xext->base = a->d_xext; xext->ld = ldx; xext->n = next; xext->k = k; xext->cplx = a->is_complex;
  xext->col0 = 0; xext->cols = nullptr;
  return 0;
}


// Overlapped ghost import for the row-gather kernel: pack on the compute stream, send/receive on the
// communication stream straight into the ghost buffer (owned x is never copied); the caller launches the
// interior blocks, then waits for ev_halo before the blocks that read ghosts.
static int halo_start(mxg_ctx* c, mxg_csr* a, const mxg_mv* x) {
  const int k = x->k;
  const int w = a->is_complex ? 2 : 1;
  const int64_t ldg = (a->nghost * w + 31) & ~(int64_t)31;      // doubles per column of the ghost buffer
  if (k > a->halo_kcap || !a->d_ghost) {
    MXG_HIP(hipStreamSynchronize(c->stream));
    MXG_HIP(hipStreamSynchronize(c->comm_stream));
    if (a->d_sendbuf) hipFree(a->d_sendbuf);
    if (a->d_xext) { hipFree(a->d_xext); a->d_xext = nullptr; }
    if (a->d_ghost) hipFree(a->d_ghost);
    const int kk = std::max(k, a->halo_kcap);
    MXG_HIP(hipMalloc(&a->d_sendbuf, sizeof(double) * std::max<int64_t>(a->nsend, 1) * w * kk));
    MXG_HIP(hipMalloc(&a->d_ghost, sizeof(double) * std::max<int64_t>(ldg, 32) * kk));
    a->halo_kcap = kk;
    a->ldghost = ldg;
  }
  for (int j = 0; j < k && a->nsend > 0; ++j) {
    const int grid = mxg_grid_for(a->nsend, 256, 1, 4 * c->num_cu);
    hipLaunchKernelGGL(k_pack, dim3(grid), dim3(256), 0, c->stream, x->colptr(j),
                       (const long long*)a->d_send_idx, (long long)a->nsend, w, a->d_sendbuf + a->nsend * w * j);
  }
  MXG_HIP(hipGetLastError());
  MXG_CHECK(mxg_comm_on(c), "halo exchange needs a communicator (nranks > 1)");
  MXG_HIP(hipEventRecord(c->ev_pack, c->stream));
  MXG_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
  std::vector<MxgXfer> xf;
  for (int j = 0; j < k; ++j)
    for (int p = 0; p < a->nneigh; ++p)
      xf.push_back({a->neigh[p], a->d_sendbuf + (a->nsend * j + a->send_off[p]) * w, (a->send_off[p + 1] - a->send_off[p]) * w,
                    a->d_ghost + ldg * j + a->recv_off[p] * w, (a->recv_off[p + 1] - a->recv_off[p]) * w});
  MXG_TRY(mxg_comm_exchange(c, xf.data(), (int)xf.size(), c->comm_stream));
  MXG_HIP(hipEventRecord(c->ev_halo, c->comm_stream));
  return 0;
}

// interior blocks, wait for the halo, boundary blocks (row-gather kernel, real scalars)
template <bool DOT>
static int launch_rowgather_halo(mxg_ctx* c, mxg_csr* a, const MvDesc& X, const MvDesc& Y, const double* xdot,
                                 double* partial) {
  const unsigned short* roff = (a->kernel & KF_ROWPTR64) ? (const unsigned short*)nullptr : (const unsigned short*)a->d_roff;
  const BlkDesc* desc = (const BlkDesc*)a->d_blkdesc;
  const long long ni = a->n_interior, nb = a->npipe - a->n_interior;
  if (ni > 0) {
    const unsigned grid = (unsigned)(((ni + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD);
    hipLaunchKernelGGL((k_spmv_rowgather<RgCfgA, DOT, false>), dim3(grid), dim3(RgCfgA::SBLK), 0, c->stream, desc, ni,
                       (const long long*)a->d_rowptr, a->d_colidx, a->d_vals, X, Y, xdot, partial, roff,
                       (const double*)nullptr, 0LL, 0);
  }
  MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
  if (nb > 0) {
    const unsigned grid = (unsigned)(((nb + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD);
    hipLaunchKernelGGL((k_spmv_rowgather<RgCfgA, DOT, true>), dim3(grid), dim3(RgCfgA::SBLK), 0, c->stream, desc + ni, nb,
                       (const long long*)a->d_rowptr, a->d_colidx, a->d_vals, X, Y, xdot, partial ? partial + ni : nullptr, roff,
                       (const double*)a->d_ghost, (long long)a->ldghost, (int)a->ncols_owned);
  }
  MXG_HIP(hipGetLastError());
  return 0;
}

// compressed-stream kernel on a row-partitioned matrix (real, one column): interior blocks, wait for the halo,
// boundary blocks with ghost reads.  halo_start(c, a, x) must have been called.
template <bool DOT, int EPI>
static int launch_cs_halo(mxg_ctx* c, mxg_csr* a, const double* x, double* y, const double* xdot, double* partial,
                          const double* rhs, const double* diag, double omega, double beta = 0.0) {
  const CsDesc* desc = (const CsDesc*)a->d_csdesc;
  const long long ni = a->ncs_interior, nb = a->ncs - a->ncs_interior;
  if (ni > 0) {
    const unsigned grid = (unsigned)(((ni + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD);
    hipLaunchKernelGGL((k_spmv_cs<RgCfgA, 7, DOT, EPI, false>), dim3(grid), dim3(RgCfgA::SBLK / CS_RPT), 0, c->stream, desc, ni,
                       (const unsigned char*)a->d_csstream, (const unsigned short*)a->d_roff, a->d_colidx, a->d_vals, x, y,
                       xdot, partial, rhs, diag, omega, (const double*)nullptr, 0LL, beta);
  }
  MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
  if (nb > 0) {
    const unsigned grid = (unsigned)(((nb + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD);
    hipLaunchKernelGGL((k_spmv_cs<RgCfgA, 7, DOT, EPI, true>), dim3(grid), dim3(RgCfgA::SBLK / CS_RPT), 0, c->stream, desc + ni, nb,
                       (const unsigned char*)a->d_csstream, (const unsigned short*)a->d_roff, a->d_colidx, a->d_vals, x, y,
                       xdot, DOT ? partial + ni : partial, rhs, diag, omega, (const double*)a->d_ghost, (long long)a->ncols_owned, beta);
  }
  MXG_HIP(hipGetLastError());
  return 0;
}

static inline bool is_halo(const mxg_csr* a) { return a->nghost > 0 || a->nneigh > 0; }
// the fused / overlapped compressed paths of a row-partitioned matrix take one column that does not alias y
static inline bool cs_halo_ok(const mxg_csr* a, const mxg_mv* x, const mxg_mv* y) {
  return cs_active(a) && a->pipe_geo == 10 && x->k == 1 && y->k == 1 && x->base != y->base && is_halo(a);
}
// contiguous column ranges of both views: [j0, j0 + kk) of x and y are kk consecutive columns of their allocations
static inline int contiguous_run(const mxg_mv* x, const mxg_mv* y, int j0) {
  int kk = 1;
  while (j0 + kk < x->k && x->hcols[j0 + kk] == x->hcols[j0] + kk && y->hcols[j0 + kk] == y->hcols[j0] + kk) ++kk;
  return kk;
}

extern "C" int mxg_csr_apply(mxg_ctx* c, const mxg_csr* a_, const mxg_mv* x, mxg_mv* y) {
  MXG_CHECK(c && a_ && x && y, "mxg_csr_apply: NULL argument");
  mxg_csr* a = const_cast<mxg_csr*>(a_);
  MXG_TRY(mxg_csr_materialize(a));
  MXG_CHECK(x->k == y->k, "mxg_csr_apply: x has %d columns, y has %d", x->k, y->k);
  MXG_CHECK(x->nlocal == a->ncols_owned, "mxg_csr_apply: x has %lld rows, matrix has %lld owned columns",
            (long long)x->nlocal, (long long)a->ncols_owned);
  MXG_CHECK(y->nlocal == a->nrows, "mxg_csr_apply: y has %lld rows, matrix has %lld rows",
            (long long)y->nlocal, (long long)a->nrows);
  MXG_CHECK(x->is_complex == a->is_complex && y->is_complex == a->is_complex,
            "mxg_csr_apply: scalar type mismatch");
  MvDesc X = x->desc();
  const MvDesc Y = y->desc();
  const bool halo = is_halo(a);
  const int w = a->is_complex ? 2 : 1;
  // ---- row-partitioned matrix, x and y distinct: ghost import overlapped with the rows that read no ghost
  if (halo && x->base != y->base) {
    if (tile_active(a)) {
      MXG_TRY(halo_start(c, a, x));
      if (a->nrows == 0) { MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0)); return 0; }
      (void)hipGetLastError();
      for (int j0 = 0; j0 < x->k;) {          // one launch pair per contiguous range of columns
        const int kk = contiguous_run(x, y, j0);
        // (every launch waits on the same halo event: the exchange carried all columns)
        MXG_TRY(mxg_tile_launch(c, a, x->colptr(j0), x->ld, y->colptr(j0), y->ld, kk, 0, 0, nullptr, nullptr, nullptr, nullptr,
                                0.0, 0.0, 1, a->d_ghost + a->ldghost * j0, a->ldghost));
        j0 += kk;
      }
      return 0;
    }
    if (cs_halo_ok(a, x, y)) {
      MXG_TRY(halo_start(c, a, x));
      if (a->nrows == 0) { MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0)); return 0; }
      (void)hipGetLastError();
      return launch_cs_halo<false, 0>(c, a, x->colptr(0), y->colptr(0), nullptr, nullptr, nullptr, nullptr, 0.0);
    }
    if (!a->is_complex && a->pipe_geo == 10) {
      // default CSR kernel: owned x read in place
      MXG_TRY(halo_start(c, a, x));
      if (a->nrows == 0) { MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0)); return 0; }
      (void)hipGetLastError();
      return launch_rowgather_halo<false>(c, a, X, Y, nullptr, nullptr);
    }
  }
  if (halo) {
    MXG_TRY(halo_exchange(c, a, x, &X));        // [owned | ghosts] copy of x: complex matrices, aliased x and y
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
  (void)hipGetLastError();   // do not attribute a stale sticky error to this launch
  const bool x_is_view = X.base == x->base;     // otherwise X is a packed copy: column j of the view is column j of the copy
  if (a->nnz == 0) {
    for (int j = 0; j < y->k; ++j) MXG_HIP(hipMemsetAsync(y->colptr(j), 0, sizeof(double) * a->nrows * w, c->stream));
  } else if (tile_active(a) && !halo) {
    for (int j0 = 0; j0 < x->k;) {
      const int kk = contiguous_run(x, y, j0);
      const double* xp = x_is_view ? x->colptr(j0) : X.base + X.ld * j0;
      MXG_TRY(mxg_tile_launch(c, a, xp, X.ld, y->colptr(j0), y->ld, kk, 0, 0, nullptr, nullptr, nullptr, nullptr, 0.0, 0.0, 0,
                              nullptr, 0));
      j0 += kk;
    }
  } else if (cs_active(a) && !halo && a->pipe_geo == 10) {
    const long long per = (a->ncs + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
    const unsigned grid = (unsigned)(per * MXG_NUM_XCD);
    for (int j = 0; j < X.k; ++j)      // one column per launch: the packed stream is cheap to re-read
      hipLaunchKernelGGL((k_spmv_cs<RgCfgA, 7, false>), dim3(grid), dim3(RgCfgA::SBLK / CS_RPT), 0, c->stream,
                         (const CsDesc*)a->d_csdesc, (long long)a->ncs, (const unsigned char*)a->d_csstream,
                         (const unsigned short*)a->d_roff, a->d_colidx, a->d_vals, (const double*)X.col(j), Y.col(j),
                         (const double*)nullptr, (double*)nullptr, (const double*)nullptr, (const double*)nullptr, 0.0,
                         (const double*)nullptr, 0LL, 0.0);
  } else if (!a->is_complex) {
    const long long per = (a->npipe + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
    const unsigned grid = (unsigned)(per * MXG_NUM_XCD);
    hipLaunchKernelGGL((k_spmv_rowgather<RgCfgA, false, false>), dim3(grid), dim3(RgCfgA::SBLK), 0, c->stream,
                       (const BlkDesc*)a->d_blkdesc, (long long)a->npipe, (const long long*)a->d_rowptr,
                       a->d_colidx, a->d_vals, X, Y, (const double*)nullptr, (double*)nullptr,
                       (a->kernel & KF_ROWPTR64) ? (const unsigned short*)nullptr : (const unsigned short*)a->d_roff);
  } else if (!(a->kernel & KF_STREAM) && a->pipe_geo == 12 && a->npipe > 0 && a->d_roff16) {
    // complex: persistent double-buffered row-gather kernel, 3 workgroups (2 x 24.5 KiB of LDS each) per CU
    const unsigned grid = (unsigned)std::max(MXG_NUM_XCD, (c->num_cu * 3 / MXG_NUM_XCD) * MXG_NUM_XCD);
    for (int j0 = 0; j0 < X.k;) {
      const int kk = contiguous_run(x, y, j0);
      const double* xp = x_is_view ? x->colptr(j0) : X.base + X.ld * j0;
      hipLaunchKernelGGL(k_spmv_rgp, dim3(grid), dim3(256), 0, c->stream, (const BlkDesc*)a->d_blkdesc,
                         (long long)a->npipe, (const int*)a->d_colidx, (const double*)a->d_vals,
                         (const unsigned short*)a->d_roff16, xp, (long long)X.ld, y->colptr(j0), (long long)y->ld, kk);
      j0 += kk;
    }
  } else {
    launch_stream<true, CfgDefault, true, true>(c, a, X, Y);
  }
  MXG_HIP(hipGetLastError());
  return 0;
}

// Product with a fused epilogue on a compressed matrix (single column, no halo): see cs_epilogue.
// Returns 1 without touching y when the matrix is not compressed (the caller takes the unfused path).
// ---- injection matrices (at most one entry per row): y += omega * A x without a matrix stream
namespace {
// flags[0] = rows with more than one entry, flags[1] = entries whose value differs from v0
__global__ void k_inj_build(const long long* __restrict__ rowptr, const int* __restrict__ colidx, const double* __restrict__ vals,
                            long long nrows, double v0, int* __restrict__ injcol, double* __restrict__ injval,
                            unsigned long long* __restrict__ flags) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const long long q = rowptr[r], len = rowptr[r + 1] - q;
  if (len > 1) atomicAdd(&flags[0], 1ull);
  if (len >= 1) {
    const double v = vals[q];
    injcol[r] = colidx[q];
    injval[r] = v;
    if (__double_as_longlong(v) != __double_as_longlong(v0)) atomicAdd(&flags[1], 1ull);
  } else {
    injcol[r] = -1;
    injval[r] = 0.0;
  }
}
// y[r] = y[r] + omega * (v * x[c]): the arithmetic of the coarse-correction epilogue (EPI 3) on a one-entry row sum
template <bool UNIFORM>
__global__ __launch_bounds__(256) void k_inj_add(const int* __restrict__ injcol, const double* __restrict__ injval, double vu,
                                                  long long nrows, const double* __restrict__ x, double* __restrict__ y,
                                                  double omega) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= nrows) return;
  const int cc = injcol[r];
  const double yy = y[r];
  double ax = 0.0;
  if (cc >= 0) ax = 0.0 + (UNIFORM ? vu : injval[r]) * x[cc];
  __builtin_nontemporal_store(yy + omega * ax, y + r);   // keep the coarse vector's lines in L2, not the streamed y
}
int inj_prepare(mxg_ctx* c, mxg_csr* a) {
  if (a->inj_state != 0) return 0;
  a->inj_state = -1;
  if (a->is_complex || a->nghost != 0 || a->nneigh != 0 || a->nrows == 0 || a->nnz > a->nrows || !a->d_rowptr) return 0;
  unsigned long long* d_flags = nullptr;
  MXG_HIP(hipMalloc(&a->d_injcol, sizeof(int32_t) * a->nrows));
  MXG_HIP(hipMalloc(&a->d_injval, sizeof(double) * a->nrows));
  MXG_HIP(hipMalloc(&d_flags, sizeof(unsigned long long) * 2));
  MXG_HIP(hipMemsetAsync(d_flags, 0, sizeof(unsigned long long) * 2, c->stream));
  double v0 = 0.0;
  if (a->nnz > 0) MXG_HIP(hipMemcpyAsync(&v0, a->d_vals, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  hipLaunchKernelGGL(k_inj_build, dim3((unsigned)((a->nrows + 255) / 256)), dim3(256), 0, c->stream, (const long long*)a->d_rowptr,
                     (const int*)a->d_colidx, (const double*)a->d_vals, (long long)a->nrows, v0, a->d_injcol, a->d_injval, d_flags);
  unsigned long long flags[2] = {1, 1};
  MXG_HIP(hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  hipFree(d_flags);
  if (flags[0] != 0) {                   // some row has two entries: not an injection
    hipFree(a->d_injcol); hipFree(a->d_injval); a->d_injcol = nullptr; a->d_injval = nullptr;
    return 0;
  }
  if (flags[1] == 0) { hipFree(a->d_injval); a->d_injval = nullptr; a->inj_uniform = v0; }
  a->inj_state = 1;
  return 0;
}
}  // namespace

// ---- ELL form of short-row gather matrices (multigrid restrictions): columns [width][nrows], coalesced, no row pointers
namespace {
// flags[0] = longest row, flags[1] = entries whose value differs from v0
__global__ void k_ell_scan(const long long* __restrict__ rowptr, const double* __restrict__ vals, long long nrows, double v0,
                           unsigned long long* __restrict__ flags) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const long long q0 = rowptr[r], q1 = rowptr[r + 1];
  atomicMax(&flags[0], (unsigned long long)(q1 - q0));
  unsigned long long diff = 0;
  for (long long q = q0; q < q1; ++q) diff += __double_as_longlong(vals[q]) != __double_as_longlong(v0);
  if (diff) atomicAdd(&flags[1], diff);
}
__global__ void k_ell_fill(const long long* __restrict__ rowptr, const int* __restrict__ colidx, const double* __restrict__ vals,
                           long long nrows, int width, int* __restrict__ ecol, double* __restrict__ eval) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const long long q0 = rowptr[r], len = rowptr[r + 1] - q0;
  for (int j = 0; j < width; ++j) {
    ecol[(size_t)j * nrows + r] = j < len ? colidx[q0 + j] : -1;
    if (eval) eval[(size_t)j * nrows + r] = j < len ? vals[q0 + j] : 0.0;
  }
}
// entries in storage order, products rounded before they are added: the row loop of the CSR kernels
template <bool UNIFORM>
__global__ __launch_bounds__(256) void k_ell_apply(const int* __restrict__ ecol, const double* __restrict__ eval, double vu,
                                                    long long nrows, int width, const double* __restrict__ x, double* __restrict__ y) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= nrows) return;
  double sum = 0.0;
  for (int j0 = 0; j0 < width; j0 += 8) {
    int cc[8];
    double vv[8], xx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int jj = j0 + j < width ? j0 + j : width - 1;
      cc[j] = ecol[(size_t)jj * nrows + r];
      vv[j] = UNIFORM ? vu : eval[(size_t)jj * nrows + r];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) xx[j] = x[cc[j] >= 0 ? cc[j] : 0];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double s2 = sum + vv[j] * xx[j];
      sum = (j0 + j < width && cc[j] >= 0) ? s2 : sum;
    }
  }
  __builtin_nontemporal_store(sum, y + r);
}
int ell_prepare(mxg_ctx* c, mxg_csr* a) {
  if (a->ell_state != 0) return 0;
  a->ell_state = -1;
  if (a->is_complex || a->nghost != 0 || a->nneigh != 0 || a->nrows == 0 || a->nnz == 0 || a->ncols_owned < 1 || !a->d_rowptr) return 0;
  unsigned long long* d_flags = nullptr;
  MXG_HIP(hipMalloc(&d_flags, sizeof(unsigned long long) * 2));
  MXG_HIP(hipMemsetAsync(d_flags, 0, sizeof(unsigned long long) * 2, c->stream));
  double v0 = 0.0;
  MXG_HIP(hipMemcpyAsync(&v0, a->d_vals, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  const unsigned grid = (unsigned)((a->nrows + 255) / 256);
  hipLaunchKernelGGL(k_ell_scan, dim3(grid), dim3(256), 0, c->stream, (const long long*)a->d_rowptr, (const double*)a->d_vals,
                     (long long)a->nrows, v0, d_flags);
  unsigned long long flags[2] = {1000, 1};
  MXG_HIP(hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  hipFree(d_flags);
  // worth it for short rows only, and only when padding does not blow the matrix up
  if (flags[0] == 0 || flags[0] > 16 || (double)flags[0] * (double)a->nrows > 1.5 * (double)a->nnz) return 0;
  a->ell_width = (int)flags[0];
  const bool uniform = flags[1] == 0;
  MXG_HIP(hipMalloc(&a->d_ellcol, sizeof(int32_t) * (size_t)a->ell_width * a->nrows));
  if (!uniform) MXG_HIP(hipMalloc(&a->d_ellval, sizeof(double) * (size_t)a->ell_width * a->nrows));
  hipLaunchKernelGGL(k_ell_fill, dim3(grid), dim3(256), 0, c->stream, (const long long*)a->d_rowptr, (const int*)a->d_colidx,
                     (const double*)a->d_vals, (long long)a->nrows, a->ell_width, a->d_ellcol, a->d_ellval);
  MXG_HIP(hipStreamSynchronize(c->stream));
  a->ell_uniform = v0;
  a->ell_state = 1;
  return 0;
}
}  // namespace

int mxg_dev_csr_apply_ell(mxg_ctx* c, mxg_csr* a, const mxg_mv* x, mxg_mv* y) {
  static const bool ell_off = getenv("MXG_ELL") && atoi(getenv("MXG_ELL")) == 0;     // read once: this runs per level per cycle
  if (x->k != 1 || y->k != 1 || x->base == y->base || ell_off) return 1;
  if (ell_prepare(c, a) != 0) return 2;
  if (a->ell_state != 1) return 1;
  const unsigned grid = (unsigned)((a->nrows + 255) / 256);
  if (a->d_ellval)
    hipLaunchKernelGGL((k_ell_apply<false>), dim3(grid), dim3(256), 0, c->stream, (const int*)a->d_ellcol, (const double*)a->d_ellval, 0.0,
                       (long long)a->nrows, a->ell_width, (const double*)x->colptr(0), y->colptr(0));
  else
    hipLaunchKernelGGL((k_ell_apply<true>), dim3(grid), dim3(256), 0, c->stream, (const int*)a->d_ellcol, (const double*)nullptr,
                       a->ell_uniform, (long long)a->nrows, a->ell_width, (const double*)x->colptr(0), y->colptr(0));
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

int mxg_dev_csr_apply_epi(mxg_ctx* c, mxg_csr* a, const mxg_mv* x, mxg_mv* y, int epi, const double* rhs,
                          const double* diag, double omega, double beta) {
  MXG_TRY(mxg_csr_materialize(const_cast<mxg_csr*>(a)));
  static const bool inject_off = getenv("MXG_INJECT") && atoi(getenv("MXG_INJECT")) == 0;
  if (epi == 3 && x->k == 1 && y->k == 1 && x->base != y->base && !inject_off) {
    if (inj_prepare(c, a) != 0) return 2;
    if (a->inj_state == 1) {
      const unsigned grid = (unsigned)((a->nrows + 255) / 256);
      if (a->d_injval)
        hipLaunchKernelGGL((k_inj_add<false>), dim3(grid), dim3(256), 0, c->stream, (const int*)a->d_injcol, (const double*)a->d_injval,
                           0.0, (long long)a->nrows, (const double*)x->colptr(0), y->colptr(0), omega);
      else
        hipLaunchKernelGGL((k_inj_add<true>), dim3(grid), dim3(256), 0, c->stream, (const int*)a->d_injcol, (const double*)nullptr,
                           a->inj_uniform, (long long)a->nrows, (const double*)x->colptr(0), y->colptr(0), omega);
      return hipGetLastError() == hipSuccess ? 0 : 2;
    }
  }
  const bool one_col = x->k == 1 && y->k == 1 && x->base != y->base;
  if (tile_active(a) && !a->is_complex && one_col) {
    const bool halo = is_halo(a);
    if (halo) {
      if (epi == 3) return 1;                    // the correction's operand is a prolongation: never row-partitioned with ghosts here
      if (halo_start(c, a, x) != 0) return 2;
      if (a->nrows == 0) { hipStreamWaitEvent(c->stream, c->ev_halo, 0); return 0; }
    }
    (void)hipGetLastError();
    if (mxg_tile_launch(c, a, x->colptr(0), x->ld, y->colptr(0), y->ld, 1, epi, 0, nullptr, nullptr, rhs, diag, omega, beta,
                        halo ? 1 : 0, a->d_ghost, a->ldghost)) return 2;
    return 0;
  }
  if (cs_halo_ok(a, x, y) && epi != 3) {
    if (halo_start(c, a, x) != 0) return 2;
    if (a->nrows == 0) { hipStreamWaitEvent(c->stream, c->ev_halo, 0); return 0; }
    (void)hipGetLastError();
    int rc;
    if (epi == 1) rc = launch_cs_halo<false, 1>(c, a, x->colptr(0), y->colptr(0), nullptr, nullptr, rhs, diag, omega);
    else if (epi == 4) rc = launch_cs_halo<false, 4>(c, a, x->colptr(0), y->colptr(0), nullptr, nullptr, rhs, diag, omega, beta);
    else if (epi == 5) rc = launch_cs_halo<false, 5>(c, a, x->colptr(0), y->colptr(0), nullptr, nullptr, rhs, diag, omega, beta);
    else if (epi == 2) rc = launch_cs_halo<false, 2>(c, a, x->colptr(0), y->colptr(0), nullptr, nullptr, rhs, diag, omega);
    else rc = launch_cs_halo<false, 0>(c, a, x->colptr(0), y->colptr(0), nullptr, nullptr, rhs, diag, omega);
    return rc == 0 ? 0 : 2;
  }
  if (!(cs_active(a) && a->pipe_geo == 10 && !is_halo(a) && one_col)) return 1;
  const MvDesc X = x->desc(), Y = y->desc();
  const long long per = (a->ncs + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  const unsigned grid = (unsigned)(per * MXG_NUM_XCD);
  (void)hipGetLastError();
#define MXG_CS_EPI(E)                                                                                          \
  hipLaunchKernelGGL((k_spmv_cs<RgCfgA, 7, false, E>), dim3(grid), dim3(RgCfgA::SBLK / CS_RPT), 0, c->stream,     \
                     (const CsDesc*)a->d_csdesc, (long long)a->ncs, (const unsigned char*)a->d_csstream,        \
                     (const unsigned short*)a->d_roff, a->d_colidx, a->d_vals, (const double*)X.col(0), Y.col(0), \
                     (const double*)nullptr, (double*)nullptr, rhs, diag, omega, (const double*)nullptr, 0LL, beta)
  if (epi == 1) MXG_CS_EPI(1); else if (epi == 2) MXG_CS_EPI(2); else if (epi == 3) MXG_CS_EPI(3); else if (epi == 4) MXG_CS_EPI(4);
  else if (epi == 5) MXG_CS_EPI(5); else MXG_CS_EPI(0);
#undef MXG_CS_EPI
  if (hipGetLastError() != hipSuccess) { mxg_set_error("mxg_dev_csr_apply_epi: launch failed"); return 2; }
  return 0;
}

// A smoother step (epilogue 1 or 4) whose launch also leaves d_out[0] = y_new . rhs (owned rows): the multigrid-CG takes
// its r.z from the last step of the cycle instead of a pass of its own over r and z.  Tile form only: returns 1
// (nothing launched) for any other form, 0 when done, 2 on an error.
int mxg_dev_csr_apply_epi_dot(mxg_ctx* c, mxg_csr* a, const mxg_mv* x, mxg_mv* y, int epi, const double* rhs,
                              const double* diag, double omega, double beta, double* d_out) {
  if (!(tile_active(a) && !a->is_complex && x->k == 1 && y->k == 1 && x->base != y->base && (epi == 1 || epi == 4))) return 1;
  const bool halo = is_halo(a);
  const int64_t np = mxg_tile_partials(a);
  if (np <= 0) return 1;
  if (!a->d_dotpartial || a->dotpartial_cap < np) {
    if (a->d_dotpartial) hipFree(a->d_dotpartial);
    if (hipMalloc(&a->d_dotpartial, sizeof(double) * np) != hipSuccess) { mxg_set_error("mxg_dev_csr_apply_epi_dot: out of memory"); return 2; }
    a->dotpartial_cap = np;
  }
  if (halo) {
    if (halo_start(c, a, x) != 0) return 2;
    if (a->nrows == 0) {
      hipStreamWaitEvent(c->stream, c->ev_halo, 0);
      return hipMemsetAsync(d_out, 0, sizeof(double), c->stream) == hipSuccess ? 0 : 2;
    }
  }
  (void)hipGetLastError();
  if (mxg_tile_launch(c, a, x->colptr(0), x->ld, y->colptr(0), y->ld, 1, epi, 1, nullptr, a->d_dotpartial, rhs, diag, omega, beta,
                      halo ? 1 : 0, a->d_ghost, a->ldghost)) return 2;
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, c->stream, (const double*)a->d_dotpartial, (long long)np, d_out);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// y = A x and d_out[0] = x . y in one pass over the matrix (real, one column, row-gather geometry).
// x . y uses the owned rows only; the caller all-reduces d_out across ranks if needed.
int mxg_dev_csr_apply_dot(mxg_ctx* c, mxg_csr* a, const mxg_mv* x, mxg_mv* y, double* d_out) {
  MXG_TRY(mxg_csr_materialize(const_cast<mxg_csr*>(a)));
  MXG_CHECK(!a->is_complex && x->k == 1 && y->k == 1 && a->nrows == a->ncols_owned,
            "mxg_dev_csr_apply_dot: real square single-column case only");
  MXG_CHECK(a->pipe_geo == 10, "mxg_dev_csr_apply_dot: row-gather descriptors missing");
  const bool halo = is_halo(a);
  if (a->npipe == 0) {              // no rows on this rank: take part in the ghost import (nothing to send), x . y = 0
    if (a->nneigh > 0) { MXG_TRY(halo_start(c, a, x)); MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0)); }
    MXG_HIP(hipMemsetAsync(d_out, 0, sizeof(double), c->stream));
    return 0;
  }
  MvDesc X = x->desc();
  const MvDesc Y = y->desc();
  const bool tile = tile_active(a);
  const int64_t np = tile ? mxg_tile_partials(a) : a->npipe;
  if (!a->d_dotpartial || a->dotpartial_cap < np) {
    if (a->d_dotpartial) hipFree(a->d_dotpartial);
    MXG_HIP(hipMalloc(&a->d_dotpartial, sizeof(double) * np));
    a->dotpartial_cap = np;
  }
  if (halo) MXG_TRY(halo_start(c, a, x));
  (void)hipGetLastError();
  if (tile) {
    MXG_TRY(mxg_tile_launch(c, a, x->colptr(0), x->ld, y->colptr(0), y->ld, 1, 0, 1, (const double*)x->colptr(0), a->d_dotpartial,
                            nullptr, nullptr, 0.0, 0.0, halo ? 1 : 0, a->d_ghost, a->ldghost));
  } else if (halo) {
    if (cs_halo_ok(a, x, y) && a->ncs == a->npipe)
      MXG_TRY((launch_cs_halo<true, 0>(c, a, x->colptr(0), y->colptr(0), (const double*)x->colptr(0), a->d_dotpartial,
                                       nullptr, nullptr, 0.0)));
    else
      MXG_TRY((launch_rowgather_halo<true>(c, a, X, Y, (const double*)x->colptr(0), a->d_dotpartial)));
  } else {
    const long long per = (a->npipe + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
    const unsigned grid = (unsigned)(per * MXG_NUM_XCD);
    if (cs_active(a) && a->ncs == a->npipe)
      hipLaunchKernelGGL((k_spmv_cs<RgCfgA, 7, true>), dim3(grid), dim3(RgCfgA::SBLK / CS_RPT), 0, c->stream,
                         (const CsDesc*)a->d_csdesc, (long long)a->ncs, (const unsigned char*)a->d_csstream,
                         (const unsigned short*)a->d_roff, a->d_colidx, a->d_vals, (const double*)X.col(0), Y.col(0),
                         (const double*)x->colptr(0), a->d_dotpartial, (const double*)nullptr, (const double*)nullptr, 0.0,
                         (const double*)nullptr, 0LL, 0.0);
    else
      hipLaunchKernelGGL((k_spmv_rowgather<RgCfgA, true>), dim3(grid), dim3(RgCfgA::SBLK), 0, c->stream,
                         (const BlkDesc*)a->d_blkdesc, (long long)a->npipe, (const long long*)a->d_rowptr,
                         a->d_colidx, a->d_vals, X, Y, (const double*)x->colptr(0), a->d_dotpartial,
                         (a->kernel & KF_ROWPTR64) ? (const unsigned short*)nullptr : (const unsigned short*)a->d_roff);
  }
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, c->stream, (const double*)a->d_dotpartial, (long long)np, d_out);
  MXG_HIP(hipGetLastError());
  return 0;
}
