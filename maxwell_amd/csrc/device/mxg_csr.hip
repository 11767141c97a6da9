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
typedef StreamCfg<512, 8> CfgWide;      // 4096 nnz, 512 threads
typedef StreamCfg<256, 4> CfgSmall;     // 1024 nnz: more, lighter blocks
typedef StreamCfg<256, 16> CfgTall;     // 4096 nnz, 256 threads, 16 entries per thread
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
template <bool CPLX, typename CFG, bool NT, bool XCD, int DIAG = 0>
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
            // DIAG builds are timing-only probes (wrong results): 1 = no x gather at all,
            // 2 = gather from a 2 KiB window (always L1 resident)
            if (DIAG == 1) prod[p] = v[e][0] * (double)c[e];
            else if (DIAG == 2) prod[p] = v[e][0] * x[c[e] & 255];
            else prod[p] = v[e][0] * x[c[e]];
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


// ------------------------------------------------------------------------------------------
// Software-pipelined persistent variant of the stream kernel (real scalars, one vector).
// The one-block-per-workgroup kernel above has four dependent memory round trips per block
// (block bounds -> row pointers / matrix stream -> x gather -> store), so even at full
// occupancy a CU keeps too few bytes in flight to saturate HBM3E.  Here a workgroup walks a
// strided sequence of row blocks and always has the NEXT block's matrix stream (values,
// column indices, row pointers: registers only, no LDS) in flight while it gathers x,
// reduces and stores the CURRENT block.  Block bounds come from one 32-byte descriptor.
// Arithmetic and summation order are unchanged: results are bit-identical to the kernel above.
// ------------------------------------------------------------------------------------------
struct BlkDesc {
  long long base;   // rowptr[r0]
  int r0, nr, nnzb; // first row, rows, non-zeros of the block
  int rb;           // position of the block's 16-bit row offsets in roff[] (r0 + block number in row order)
  int pad[2];
};

template <typename CFG>
struct PipeRegs {
  double v[CFG::EPT];
  int c[CFG::EPT];
  int o0[2], o1[2];   // row offsets (relative to base) of the <= 2 rows this thread sums
};

template <typename CFG>
__device__ inline void pipe_load(PipeRegs<CFG>& R, const BlkDesc& d, const long long* __restrict__ rowptr,
                                 const int* __restrict__ colidx, const double* __restrict__ vals) {
  constexpr int SB = CFG::SBLK, EPT = CFG::EPT;
  const long long lo = d.base & ~1LL;
  const long long end = d.base + d.nnzb;
#pragma unroll
  for (int e = 0; e < EPT; e += 2) {
    const long long idx = lo + 2 * (threadIdx.x + (long long)(e / 2) * SB);
    if (idx < end) {   // pair (idx, idx+1) touches the block (idx + 1 >= base always)
      dbl2 vv = nt_load(reinterpret_cast<const dbl2*>(vals + idx));
      int2v cc = nt_load(reinterpret_cast<const int2v*>(colidx + idx));
      R.v[e] = vv.x; R.v[e + 1] = vv.y;
      R.c[e] = cc.x; R.c[e + 1] = cc.y;
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int t = threadIdx.x + r * SB;
    if (t < d.nr) {
      R.o0[r] = (int)(nt_load(rowptr + d.r0 + t) - d.base);
      R.o1[r] = (int)(nt_load(rowptr + d.r0 + t + 1) - d.base);
    }
  }
}

// issue the x gathers of a loaded block (its column indices are already in registers)
template <typename CFG>
__device__ inline void pipe_gather(const PipeRegs<CFG>& R, const BlkDesc& d, double (&xv)[CFG::EPT],
                                   const double* __restrict__ x) {
  constexpr int SB = CFG::SBLK, EPT = CFG::EPT;
  const int shift = (int)((d.base & ~1LL) - d.base);   // 0 or -1
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int p = shift + 2 * (threadIdx.x + (e / 2) * SB) + (e & 1);
    xv[e] = (p >= 0 && p < d.nnzb && d.nnzb <= CFG::NNZ_CAP) ? x[R.c[e]] : 0.0;
  }
}

template <typename CFG>
__device__ inline void pipe_finish(const PipeRegs<CFG>& R, const BlkDesc& d, const double (&xv)[CFG::EPT],
                                   double* prod, const int* __restrict__ colidx,
                                   const double* __restrict__ vals, const double* __restrict__ x,
                                   double* __restrict__ y) {
  constexpr int SB = CFG::SBLK, EPT = CFG::EPT;
  if (d.nnzb <= CFG::NNZ_CAP) {
    const int shift = (int)((d.base & ~1LL) - d.base);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int p = shift + 2 * (threadIdx.x + (e / 2) * SB) + (e & 1);
      if (p >= 0 && p < d.nnzb) prod[p] = R.v[e] * xv[e];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int t = threadIdx.x + r * SB;
      if (t < d.nr) {
        double sum = 0.0;
        for (int q = R.o0[r]; q < R.o1[r]; ++q) sum += prod[q];
        __builtin_nontemporal_store(sum, y + d.r0 + t);
      }
    }
    __syncthreads();
  } else {
    // single over-long row: whole workgroup strides over it, fixed-order combine
    double sr = 0.0;
    for (long long q = d.base + threadIdx.x; q < d.base + d.nnzb; q += SB) sr += vals[q] * x[colidx[q]];
    prod[threadIdx.x] = sr;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tr = 0.0;
      for (int q = 0; q < SB; ++q) tr += prod[q];
      y[d.r0] = tr;
    }
    __syncthreads();
  }
}

template <typename CFG>
__global__ __launch_bounds__(CFG::SBLK) void k_spmv_pipe(
    const BlkDesc* __restrict__ desc, long long nblocks, const long long* __restrict__ rowptr,
    const int* __restrict__ colidx, const double* __restrict__ vals, const double* __restrict__ x,
    double* __restrict__ y) {
  __shared__ double prod[CFG::NNZ_BLK];
  // XCD x walks its own contiguous eighth of the blocks; its workgroups interleave inside it
  const long long per = (nblocks + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  const long long xcd = blockIdx.x % MXG_NUM_XCD, slot = blockIdx.x / MXG_NUM_XCD;
  const long long stride = gridDim.x / MXG_NUM_XCD;
  long long b = xcd * per + slot;
  long long last = (xcd + 1) * per;
  if (last > nblocks) last = nblocks;
  if (b >= last) return;
  PipeRegs<CFG> A, B;
  double xv[CFG::EPT];
  BlkDesc dA = desc[b], dB;
  pipe_load<CFG>(A, dA, rowptr, colidx, vals);
  while (true) {
    // the gathers of the current block are issued BEFORE the next block's stream: vmcnt retires
    // in order, so the products below only wait for the gathers, not for the prefetch
    pipe_gather<CFG>(A, dA, xv, x);
    const bool moreB = b + stride < last;
    if (moreB) { dB = desc[b + stride]; pipe_load<CFG>(B, dB, rowptr, colidx, vals); }
    pipe_finish<CFG>(A, dA, xv, prod, colidx, vals, x, y);
    if (!moreB) break;
    b += stride;
    pipe_gather<CFG>(B, dB, xv, x);
    const bool moreA = b + stride < last;
    if (moreA) { dA = desc[b + stride]; pipe_load<CFG>(A, dA, rowptr, colidx, vals); }
    pipe_finish<CFG>(B, dB, xv, prod, colidx, vals, x, y);
    if (!moreA) break;
    b += stride;
  }
}

typedef StreamCfg<256, 4> PipeCfgA;   // 1024 nnz per block, 8 KiB LDS
typedef StreamCfg<256, 8> PipeCfgB;   // 2048 nnz per block, 16 KiB LDS


// ------------------------------------------------------------------------------------------
// x-window kernel (real scalars, one vector): the production SpMV.
// Profiling the stream kernel showed the per-lane x gather through the vector L1 (one tag
// lookup per touched 128-B line, ~12 lines per wave instruction) costs ~25 % of the run time
// even when every line hits.  Here the x entries a row block needs are staged in LDS first:
// at setup the block's column indices are clustered into <= 8 contiguous windows of x and
// every non-zero gets a 16-bit LDS offset (window offset + col - window start).  The kernel
// loads the windows with coalesced 16-byte loads, gathers from LDS, and never touches the
// 32-bit column indices: the matrix stream shrinks from 12 to 10 bytes per non-zero.
// Blocks whose columns do not cluster (more than 8 windows / more x entries than the LDS
// tile) keep the global gather.  Products and the per-row summation order are unchanged, so
// the result is still bit-identical to the serial CSR row loop.
// ------------------------------------------------------------------------------------------
constexpr int XW_MAXWIN = 8;
struct BlkDescX {
  long long base;            // rowptr[r0]
  int r0, nr, nnzb, nwin;    // nwin < 0: unstaged block (global gather through colidx)
  int win_lo[XW_MAXWIN];     // first x entry of each window (even)
  int win_len[XW_MAXWIN];    // entries (even)
  int pad[2];
};
static_assert(sizeof(BlkDescX) == 96, "descriptor layout");

template <int SBLK_, int EPT_, int XCAP_>
struct XwCfg {
  static constexpr int SBLK = SBLK_, EPT = EPT_, XCAP = XCAP_;
  static constexpr int NNZ_BLK = SBLK_ * EPT_;
  static constexpr int NNZ_CAP = NNZ_BLK - 1;
  static constexpr int ROWS_CAP = 2 * SBLK_;
};
typedef XwCfg<256, 8, 2048> XwCfgA;   // 16 KiB products + 16 KiB x tile
typedef XwCfg<256, 4, 1024> XwCfgB;   //  8 KiB products +  8 KiB x tile
typedef unsigned short us2v __attribute__((ext_vector_type(2)));

template <typename CFG>
__global__ __launch_bounds__(CFG::SBLK) void k_spmv_xwin(
    const BlkDescX* __restrict__ desc, long long nblocks, const long long* __restrict__ rowptr,
    const int* __restrict__ colidx, const unsigned short* __restrict__ lcol,
    const double* __restrict__ vals, MvDesc X, MvDesc Y) {
  constexpr int SB = CFG::SBLK, EPT = CFG::EPT;
  __shared__ double prod[CFG::NNZ_BLK];
  __shared__ double xs[CFG::XCAP];
  const long long b = xcd_block(blockIdx.x, nblocks);
  if (b >= nblocks) return;
  const BlkDescX& d = desc[b];
  const long long base = d.base;
  const int r0 = d.r0, nr = d.nr, nnzb = d.nnzb, nwin = d.nwin;
  if (nnzb > CFG::NNZ_CAP) {
    // single over-long row: whole workgroup strides over it, fixed-order combine
    for (int j = 0; j < X.k; ++j) {
      const double* x = X.col(j);
      double sr = 0.0;
      for (long long q = base + threadIdx.x; q < base + nnzb; q += SB) sr += vals[q] * x[colidx[q]];
      if (j > 0) __syncthreads();
      prod[threadIdx.x] = sr;
      __syncthreads();
      if (threadIdx.x == 0) {
        double tr = 0.0;
        for (int q = 0; q < SB; ++q) tr += prod[q];
        Y.col(j)[r0] = tr;
      }
    }
    return;
  }
  const long long lo = base & ~1LL;
  const long long end = base + nnzb;
  const int shift = (int)(lo - base);   // 0 or -1
  // 1. matrix stream of the block -> registers (values + 16-bit local columns, or 32-bit columns)
  double v[EPT];
  int c[EPT];
#pragma unroll
  for (int e = 0; e < EPT; e += 2) {
    const long long idx = lo + 2 * (threadIdx.x + (long long)(e / 2) * SB);
    if (idx < end) {
      dbl2 vv = nt_load(reinterpret_cast<const dbl2*>(vals + idx));
      v[e] = vv.x; v[e + 1] = vv.y;
      if (nwin >= 0) {
        us2v cc = nt_load(reinterpret_cast<const us2v*>(lcol + idx));
        c[e] = cc.x; c[e + 1] = cc.y;
      } else {
        int2v cc = nt_load(reinterpret_cast<const int2v*>(colidx + idx));
        c[e] = cc.x; c[e + 1] = cc.y;
      }
    }
  }
  // row offsets of the (at most two) rows this thread sums
  int o0[2], o1[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int t = threadIdx.x + r * SB;
    if (t < nr) {
      o0[r] = (int)(nt_load(rowptr + r0 + t) - base);
      o1[r] = (int)(nt_load(rowptr + r0 + t + 1) - base);
    }
  }
  for (int j = 0; j < X.k; ++j) {
  const double* __restrict__ x = X.col(j);
  double* __restrict__ y = Y.col(j);
  if (j > 0) __syncthreads();   // previous column is done with prod[] / xs[]
  // 2. x windows -> LDS (coalesced 16-byte loads; x stays cacheable, it is re-used by neighbours)
  if (nwin > 0) {
    int off = 0;
    for (int w = 0; w < nwin; ++w) {
      const int wlo = d.win_lo[w], wlen = d.win_len[w];
      for (int i = 2 * threadIdx.x; i < wlen; i += 2 * SB)
        *reinterpret_cast<double2*>(xs + off + i) = *reinterpret_cast<const double2*>(x + wlo + i);
      off += wlen;
    }
    __syncthreads();
  }
  // 3. products
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int p = shift + 2 * (threadIdx.x + (e / 2) * SB) + (e & 1);
    if (p >= 0 && p < nnzb) prod[p] = v[e] * (nwin >= 0 ? xs[c[e]] : x[c[e]]);
  }
  __syncthreads();
  // 4. per-row sums in storage order
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int t = threadIdx.x + r * SB;
    if (t < nr) {
      double sum = 0.0;
      for (int q = o0[r]; q < o1[r]; ++q) sum += prod[q];
      __builtin_nontemporal_store(sum, y + r0 + t);
    }
  }
  }  // columns
}


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
typedef RgCfg<256, 3072> RgCfgB;   // rows up to 12 nnz stay at 256 rows per block

// DOT: additionally emit partial[b] = sum over the block's rows of y[row] * xdot[row] (the p.Ap of
// CG fused into the product; one value per block, combined in block order by k_sum_partials)
// HALO: columns >= nowned are ghosts and live in a separate buffer (column j at xghost + j * ldghost): the
// owned part of x is read in place, no [owned | ghosts] copy is assembled.
template <typename CFG, int CH, bool DOT = false, bool HALO = false, int KT = 1>
__global__ __launch_bounds__(CFG::SBLK) void k_spmv_rowgather(
    const BlkDesc* __restrict__ desc, long long nblocks, const long long* __restrict__ rowptr,
    const int* __restrict__ colidx, const double* __restrict__ vals, MvDesc X, MvDesc Y,
    const double* __restrict__ xdot = nullptr, double* __restrict__ partial = nullptr,
    const unsigned short* __restrict__ roff = nullptr, const double* __restrict__ xghost = nullptr,
    long long ldghost = 0, int nowned = 0) {
  constexpr int SB = CFG::SBLK, EPT = CFG::EPT;
  __shared__ double sval[CFG::NNZ_BLK];
  __shared__ int scol[CFG::NNZ_BLK];
  const long long b = xcd_block(blockIdx.x, nblocks);
  if (b >= nblocks) return;
  const BlkDesc d = desc[b];
  if (d.nnzb > CFG::NNZ_CAP) {
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
  for (int j0 = 0; j0 < X.k; j0 += KT) {
    // a tile of KT columns (KT > 1: contiguous column views only, see the launch site): the CH x KT gathers of a
    // chunk are issued back to back; indices clamped into the row, entries past its end computed and discarded
    const double* __restrict__ xc[KT];
    const double* __restrict__ xgc[KT];
#pragma unroll
    for (int jj = 0; jj < KT; ++jj) {
      const int j = min(j0 + jj, X.k - 1);
      xc[jj] = X.col(j);
      xgc[jj] = HALO ? xghost + j * ldghost : nullptr;
    }
    double sumt[KT];
#pragma unroll
    for (int jj = 0; jj < KT; ++jj) sumt[jj] = 0.0;
    for (int q0 = o0; q0 < o1; q0 += CH) {
      double v[CH], xv[KT][CH];
      int cc[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int q = min(q0 + i, o1 - 1);
        cc[i] = scol[q];
        v[i] = sval[q];
      }
#pragma unroll
      for (int jj = 0; jj < KT; ++jj)
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          if (HALO) xv[jj][i] = cc[i] >= nowned ? xgc[jj][cc[i] - nowned] : xc[jj][cc[i]];
          else xv[jj][i] = xc[jj][cc[i]];
        }
#pragma unroll
      for (int jj = 0; jj < KT; ++jj)
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const double s2 = sumt[jj] + v[i] * xv[jj][i];
          sumt[jj] = q0 + i < o1 ? s2 : sumt[jj];
        }
    }
    if ((int)threadIdx.x < d.nr)
#pragma unroll
      for (int jj = 0; jj < KT; ++jj)
        if (j0 + jj < X.k) __builtin_nontemporal_store(sumt[jj], Y.col(j0 + jj) + d.r0 + threadIdx.x);
    const int j = j0;
    const double sum = sumt[0];
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
// Persistent, double-buffered row-gather kernel ("rgp"): the same arithmetic as k_spmv_rowgather, but a
// workgroup walks a strided sequence of row blocks and the NEXT block's values / columns / row offsets are
// already on their way into the other half of LDS - by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no
// ds_write) - while the current block gathers x.  One block per launch slot paid four dependent round trips
// (descriptor -> stream -> gather -> store) with its 24 KiB of LDS idle for three of them; here the descriptor
// is two iterations ahead (scalar loads), the stream one iteration ahead, and a CU always has 3 x 24 KiB of
// matrix stream in flight.  Workgroups of one XCD take interleaved blocks of that XCD's contiguous eighth, so
// they advance through x together and the 4 MiB L2 sees the same window as before.
// LDS image of a block = the global arrays from a 16-byte aligned start: entry q of the block's CSR range sits
// at sval[q - (base & ~1)] and scol[q - (base & ~3)]; row offsets come from a copy laid out 528 B per block.
// ------------------------------------------------------------------------------------------
constexpr int RGP_NNZ_CAP = RgCfgA::NNZ_CAP;   // 2044: + <= 3 entries of alignment slack stays inside 2048 slots
constexpr int RGP_ROFF_STRIDE = 264;       // ushorts per block in roff16[] (257 offsets, padded to 16 B)
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

// CPLX: an entry's value is 16 bytes (re, im): the 16 KiB value image holds 1024 entries, the block partition of a
// complex matrix caps a block at RGP_NNZ_CAP_CPLX
constexpr int RGP_NNZ_CAP_CPLX = 1020;
template <int AUX, bool CPLX>
__device__ inline void rgp_issue(const RgpBlk& d, const int* __restrict__ colidx, const double* __restrict__ vals,
                                 const unsigned short* __restrict__ roff16, long long b, double* sv, int* sc,
                                 unsigned short* sr) {
  if (d.nnzb > (CPLX ? RGP_NNZ_CAP_CPLX : RGP_NNZ_CAP)) return;        // one over-long row: read in place
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long end = d.base + d.nnzb;
  const long long lo = CPLX ? d.base : (d.base & ~1LL), lo4 = d.base & ~3LL;
#pragma unroll
  for (int p = 0; p < 4; ++p) {            // values: 16 pieces of 1 KiB, four per wave
    const int piece = wave + 4 * p;
    if (CPLX) {
      const long long idx = lo + 64 * piece + lane;                   // one complex entry per lane
      if (idx < end) lds_dma16<AUX>(vals + 2 * idx, sv + 128 * piece);
    } else {
      const long long idx = lo + 128 * piece + 2 * lane;
      if (idx < end) lds_dma16<AUX>(vals + idx, sv + 128 * piece);
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {            // columns: 8 pieces, two per wave
    const int piece = wave + 4 * p;
    const long long idx = lo4 + 256 * piece + 4 * lane;
    if (idx < end) lds_dma16<AUX>(colidx + idx, sc + 256 * piece);
  }
  if (wave == 0 && 8 * lane <= d.nr) lds_dma16<0>(roff16 + b * RGP_ROFF_STRIDE + 8 * lane, sr);
}

// x, y: first column and leading dimension in doubles (contiguous column ranges; the host splits other views into
// ranges).  KT: columns whose gathers are in flight together (a block of k vectors is walked in tiles of KT columns
// over the LDS-resident matrix entries).  CPLX: per entry (ar xr - ai xi, ar xi + ai xr), each product rounded, sums
// in storage order - the arithmetic of oracle_spmv_complex.
template <int CH, int AUX, bool CPLX = false, int KT = 1>
__global__ __launch_bounds__(256) void k_spmv_rgp(
    const BlkDesc* __restrict__ desc, long long nblocks, const int* __restrict__ colidx,
    const double* __restrict__ vals, const unsigned short* __restrict__ roff16,
    const double* __restrict__ x0, long long ldx, double* __restrict__ y0, long long ldy, int k) {
  constexpr int CAP = CPLX ? RGP_NNZ_CAP_CPLX : RGP_NNZ_CAP;
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
  rgp_issue<AUX, CPLX>(d, colidx, vals, roff16, b, sval[0], scol[0], sroff[0]);
  bool has_next = b + W < last;
  if (has_next) dn = rgp_desc(desc, b + W);
  for (int it = 0; b < last; b += W, ++it) {
    const int s = it & 1;
    lds_dma_wait_all();
    __syncthreads();       // block b has landed (every wave waited for its own pieces), nobody reads the other half any more
    RgpBlk dnn = dn;
    const bool has_nn = b + 2 * W < last;
    if (has_next) {
      rgp_issue<AUX, CPLX>(dn, colidx, vals, roff16, b + W, sval[s ^ 1], scol[s ^ 1], sroff[s ^ 1]);
      if (has_nn) dnn = rgp_desc(desc, b + 2 * W);
    }
    if (d.nnzb > CAP) {
      for (int j = 0; j < k; ++j) {
        const double* x = x0 + j * ldx;
        double sr = 0.0, si = 0.0;
        for (long long q = d.base + t; q < d.base + d.nnzb; q += 256) {
          const int cc = colidx[q];
          if (CPLX) {
            const double ar = vals[2 * q], ai = vals[2 * q + 1], xr = x[2 * (long long)cc], xi = x[2 * (long long)cc + 1];
            sr += ar * xr - ai * xi;
            si += ar * xi + ai * xr;
          } else {
            sr += vals[q] * x[cc];
          }
        }
        __syncthreads();
        sval[s][t] = sr;
        if (CPLX) sval[s][256 + t] = si;
        __syncthreads();
        if (t == 0) {
          double tr = 0.0, ti = 0.0;
          for (int q = 0; q < 256; ++q) { tr += sval[s][q]; if (CPLX) ti += sval[s][256 + q]; }
          if (CPLX) { y0[j * ldy + 2 * (long long)d.r0] = tr; y0[j * ldy + 2 * (long long)d.r0 + 1] = ti; }
          else y0[j * ldy + d.r0] = tr;
        }
      }
    } else {
      const int* __restrict__ sc = scol[s] + (int)(d.base & 3);
      int o0 = 0, o1 = 0;
      if (t < d.nr) { o0 = sroff[s][t]; o1 = sroff[s][t + 1]; }
      if (CPLX) {
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
      } else {
        const double* __restrict__ sv = sval[s] + (int)(d.base & 1);
        for (int j0 = 0; j0 < k; j0 += KT) {
          // a tile of KT columns: the CH x KT gathers of a chunk are issued back to back (indices clamped into the
          // row; entries past the row's end are computed and discarded)
          double sum[KT];
#pragma unroll
          for (int jj = 0; jj < KT; ++jj) sum[jj] = 0.0;
          for (int q0 = o0; q0 < o1; q0 += CH) {
            double v[CH], xv[KT][CH];
            int cc[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
              const int q = min(q0 + i, o1 - 1);
              cc[i] = sc[q];
              v[i] = sv[q];
            }
#pragma unroll
            for (int jj = 0; jj < KT; ++jj) {
              const double* __restrict__ x = x0 + min(j0 + jj, k - 1) * ldx;
#pragma unroll
              for (int i = 0; i < CH; ++i) xv[jj][i] = x[cc[i]];
            }
#pragma unroll
            for (int jj = 0; jj < KT; ++jj)
#pragma unroll
              for (int i = 0; i < CH; ++i) {
                const double s2 = sum[jj] + v[i] * xv[jj][i];
                sum[jj] = q0 + i < o1 ? s2 : sum[jj];
              }
          }
          if (t < d.nr)
#pragma unroll
            for (int jj = 0; jj < KT; ++jj)
              if (j0 + jj < k) __builtin_nontemporal_store(sum[jj], y0 + (j0 + jj) * ldy + d.r0 + t);
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
  int rec16, pad;           // length of the block's record in 16-byte units (the persistent kernel's loader)
  __host__ __device__ inline int nr() const { return nrnv & 0xffff; }
  __host__ __device__ inline int nval() const { return (nrnv >> 16) & 0x1ff; }
  // the block's record holds only its row offsets (more bands / distinct values / bytes than the packed form takes):
  // columns and values come from the CSR arrays
  __host__ __device__ inline bool roff_only() const { return (nrnv >> 30) & 1; }
  __host__ __device__ inline bool patterned() const { return (nrnv >> 29) & 1; }
};
constexpr int CS_ROFF_ONLY = 1 << 30;   // generic rows read the CSR arrays in place (row offsets are those of the CSR arrays)
constexpr int CS_PATTERNED = 1 << 29;   // the record has a pattern table + per-row pattern ids (persistent kernel only)
constexpr int CS_HAS_GENERIC = 1 << 28; // patterned record with a generic part for the rows no pattern covers
// Pattern table of a patterned record: most rows of a Yee operator's block repeat a handful of (column - row, value)
// sequences (the bulk stencil, per z-line); such a row costs one byte instead of 3 bytes per entry.
constexpr int CS_NPAT = 4, CS_PLEN = 8;
constexpr int CS_PAT_BYTES = CS_NPAT * CS_PLEN * 4 + CS_NPAT * CS_PLEN * 8 + 16;   // deltas, values, lengths
constexpr int CS_ROWPAT_BYTES = 256;
static_assert(sizeof(CsDesc) == 40, "descriptor layout");
constexpr int CS_SEGS = 16, CS_OFFBITS = 12, CS_MAXVAL = 256;
// every record starts with a 32-byte header {r0, nrnv (as in CsDesc), nnzb, 0, base, 0}: the persistent kernel's
// consumer waves take a block's shape from LDS and never wait for a global descriptor
constexpr int CS_HDR = 32;

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

// ------------------------------------------------------------------------------------------
// Persistent compressed-stream kernel ("csp"): a dedicated LOADER wave keeps the packed records of the next D - 1
// row blocks in flight by LDS-DMA into a ring of D slots while four CONSUMER waves (one row per thread) decode
// and gather the current one.  k_spmv_cs has one block per workgroup and three dependent round trips per block
// (descriptor -> record -> gathers); at 16 waves per CU that left ~0.8 GB moving at 3.8 TB/s.  Here
//   * a block's record (band origins, value table, codes, value ids AND its row offsets) is one contiguous,
//     16-byte aligned run of <= 8 KiB = eight 1 KiB DMA pieces, always eight (lanes past the end re-read the
//     record's first bytes), so the loader's vmcnt arithmetic is static: after issuing block i + D - 1 it waits
//     s_waitcnt vmcnt(8 (D - 2)), i.e. for block i + 1, and never drains the queue;
//   * vmcnt is per wave: the consumers' gathers and stores do not touch the loader's counter, and the DMA is inline
//     asm, so the compiler puts no wait in front of the consumers' LDS reads;
//   * one raw s_barrier per block hands slot i to the consumers and slot i - 1 back to the loader;
//   * 24 KiB of LDS per workgroup at D = 3: six workgroups (24 consumer waves, 12 records in flight) per CU.
// Same decode, same products, same summation order as k_spmv_cs: bit-identical results.
// ------------------------------------------------------------------------------------------
constexpr int CSP_SLOT = 8192;                       // bytes per ring slot = 8 pieces
constexpr int CSP_THREADS = 320;                     // 4 consumer waves + 1 loader wave

__host__ __device__ inline int cs_roff_bytes(int nr) { return ((nr + 1 + 7) >> 3) * 16; }
// bytes of a block's record; packed: [header][16 band origins][value table][codes][value ids][row offsets]
__host__ __device__ inline int cs_record_bytes(int nnzb, int nval, int nr, bool roff_only) {
  if (roff_only) return CS_HDR + CS_SEGS * 4 + cs_roff_bytes(nr);
  return CS_HDR + CS_SEGS * 4 + ((nval + 1) & ~1) * 8 + ((nnzb + 7) >> 3) * 16 + ((nnzb + 15) >> 4) * 16 + cs_roff_bytes(nr);
}

// workgroup barrier for LDS hand-offs without the vmcnt(0) a __syncthreads() carries: the wave's own LDS operations
// are complete, the "memory" clobber keeps the compiler from moving LDS accesses across it
__device__ inline void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N> __device__ inline void csp_wait_pieces();
template <> __device__ inline void csp_wait_pieces<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ inline void csp_wait_pieces<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
template <> __device__ inline void csp_wait_pieces<16>() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
template <> __device__ inline void csp_wait_pieces<24>() { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
template <> __device__ inline void csp_wait_pieces<32>() { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); }
template <> __device__ inline void csp_wait_pieces<48>() { asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); }

// One row of one staged record, in two phases so that a thread can have the gathers of several rows (of several
// records) in flight before it consumes any.  The consumers are instruction-bound (rocprofv3: SIMDs ~75 % busy at
// 0.20 ms, 127 VALU instructions per row with per-entry selects and 64-bit address arithmetic), so the pattern path
// is kept to one add, one gather and two products per entry: every pattern of a record has the same length (a
// wave-uniform loop bound from the header), deltas are stored in BYTES and x is addressed as base + 32-bit offset.
template <bool HALO>
struct CspRow {
  double xv[CS_PLEN];       // gathered x of the pattern's entries (the values stay in LDS until they are needed)
  double sum;
  int plen, r0, nr, nrnv, nnzg, pat, o0, o1;
  bool live, generic, longrow, patrow, packed;
  const unsigned char* rec;
  const unsigned char* gen;

  // phase 1a: header and pattern id (LDS reads only, so that the compiler can batch them over the thread's rows)
  __device__ inline void head(const unsigned char* record, int t, bool present) {
    rec = record;
    const uint4v h = *reinterpret_cast<const uint4v*>(rec);
    r0 = __builtin_amdgcn_readfirstlane((int)h.x);
    nrnv = __builtin_amdgcn_readfirstlane((int)h.y);
    nnzg = __builtin_amdgcn_readfirstlane((int)h.z);
    plen = __builtin_amdgcn_readfirstlane((int)h.w);
    nr = nrnv & 0xffff;
    longrow = present && nnzg > RgCfgA::NNZ_CAP;
    live = present && !longrow && t < nr;
    sum = 0.0; pat = 0; generic = live; patrow = false;
    gen = rec + CS_HDR;
    if (nrnv & CS_PATTERNED) {
      gen += CS_PAT_BYTES + CS_ROWPAT_BYTES;
      // a row that repeats one of the block's patterns: its entries are (row + delta_i, value_i) from the pattern
      // table (the same LDS address for every lane with that pattern: broadcast reads)
      const int p = live ? rec[CS_HDR + CS_PAT_BYTES + t] : 0xFF;
      generic = live && p == 0xFF && (nrnv & CS_HAS_GENERIC) != 0;
      patrow = p != 0xFF;
      pat = patrow ? p : 0;
    }
    packed = generic && !(nrnv & CS_ROFF_ONLY);
    o0 = 0; o1 = 0;
    if (packed) {
      const int nvpad = (((nrnv >> 16) & 0x1ff) + 1) & ~1;
      const unsigned short* sroff = reinterpret_cast<const unsigned short*>(
          gen + CS_SEGS * 4 + nvpad * 8 + ((nnzg + 7) >> 3) * 16 + ((nnzg + 15) >> 4) * 16);
      o0 = sroff[t]; o1 = sroff[t + 1];
    }
  }

  // phase 1b: the gathers of the row's first CS_PLEN entries.
  // The loads are inline asm: hipcc cannot prove that a load issued under one exec mask is consumed under the same
  // mask later and guards the re-use of its destination registers with s_waitcnt vmcnt(0) - in front of every entry,
  // which serialises the row's gathers.  Issued this way they are invisible to its counters; the kernel waits once
  // before any of them is consumed.  ONE asm statement per entry, whatever kind of row: two statements in the arms
  // of a branch would be merged by register copies placed right behind them - copies of registers whose loads are
  // still in flight.
  __device__ inline void gather(int t, const double* __restrict__ x, const double* __restrict__ xg, int nowned) {
    unsigned off[CS_PLEN];
#pragma unroll
    for (int i = 0; i < CS_PLEN; ++i) off[i] = 0;
    int nload = 0;
    if (patrow) {
      const uint4v* dp = reinterpret_cast<const uint4v*>(rec + CS_HDR + pat * (CS_PLEN * 4));
      const uint4v da = dp[0], db = dp[1];               // the pattern's eight byte deltas: two 16-byte LDS reads
      const unsigned dl8[CS_PLEN] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
      const unsigned off0 = (unsigned)(r0 + t) << 3;
#pragma unroll
      for (int i = 0; i < CS_PLEN; ++i) off[i] = off0 + dl8[i];
      nload = plen;
    } else if (packed) {
      // a row no pattern covers: indices clamped into the row (an empty row gathers x[0] and discards it)
      const int nvpad = (((nrnv >> 16) & 0x1ff) + 1) & ~1;
      const int* sseg = reinterpret_cast<const int*>(gen);
      const unsigned short* scode = reinterpret_cast<const unsigned short*>(gen + CS_SEGS * 4 + nvpad * 8);
      const int qlast = o1 > o0 ? o1 - 1 : o0;
      unsigned code[CS_PLEN];
#pragma unroll
      for (int i = 0; i < CS_PLEN; ++i) code[i] = scode[min(o0 + i, qlast)];
#pragma unroll
      for (int i = 0; i < CS_PLEN; ++i) {
        const int cc = sseg[code[i] >> CS_OFFBITS] + (int)(code[i] & ((1u << CS_OFFBITS) - 1u));
        off[i] = (unsigned)(o0 + i < o1 ? cc : 0) << 3;
      }
      nload = CS_PLEN;
    }
#pragma unroll
    for (int i = 0; i < CS_PLEN; ++i)
      if (i < nload) {
        if (HALO) {
          const unsigned nb8 = (unsigned)nowned << 3;
          const char* addr = off[i] >= nb8 ? reinterpret_cast<const char*>(xg) + (off[i] - nb8)
                                           : reinterpret_cast<const char*>(x) + off[i];
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(xv[i]) : "v"(addr));
        } else {
          asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(xv[i]) : "v"(off[i]), "s"(x));
        }
      }
  }

  // phase 2: rows no pattern covers (their own gathers, consumed at once), then the pattern's products in storage order
  template <int CH, typename XAT>
  __device__ inline void finish(int t, const int* __restrict__ colidx, const double* __restrict__ vals, XAT xat) {
    if (generic) {
      const int nvpad = (((nrnv >> 16) & 0x1ff) + 1) & ~1;
      if (packed) {
        const int* sseg = reinterpret_cast<const int*>(gen);
        const double* svtab = reinterpret_cast<const double*>(gen + CS_SEGS * 4);
        const unsigned short* scode = reinterpret_cast<const unsigned short*>(gen + CS_SEGS * 4 + nvpad * 8);
        const unsigned char* svid = gen + CS_SEGS * 4 + nvpad * 8 + ((nnzg + 7) >> 3) * 16;
        const int qlast = o1 > o0 ? o1 - 1 : o0;
        // first chunk: gathered in phase 1
        int vid[CS_PLEN];
#pragma unroll
        for (int i = 0; i < CS_PLEN; ++i) vid[i] = svid[min(o0 + i, qlast)];
#pragma unroll
        for (int i = 0; i < CS_PLEN; ++i) {
          asm volatile("" : "+v"(xv[i]));
          const double s2 = sum + svtab[vid[i]] * xv[i];
          sum = o0 + i < o1 ? s2 : sum;
        }
        // longer rows (wall rows have up to 15 entries): further chunks of CH, each a batch of gathers
        for (int q0 = o0 + CS_PLEN; q0 < o1; q0 += CH) {
          double gv[CH], gx[CH];
          unsigned code[CH];
          int vd[CH];
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const int q = min(q0 + i, qlast);
            code[i] = scode[q];
            vd[i] = svid[q];
          }
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            gx[i] = xat(sseg[code[i] >> CS_OFFBITS] + (int)(code[i] & ((1u << CS_OFFBITS) - 1u)));
            gv[i] = svtab[vd[i]];
          }
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const double s2 = sum + gv[i] * gx[i];
            sum = q0 + i < o1 ? s2 : sum;
          }
        }
      } else {
        const long long base = reinterpret_cast<const long long*>(rec)[2];
        const unsigned short* sroff = reinterpret_cast<const unsigned short*>(gen + CS_SEGS * 4);
        const int c0 = sroff[t], c1 = sroff[t + 1];
#pragma unroll 1
        for (long long q = base + c0; q < base + c1; ++q) sum += vals[q] * xat(colidx[q]);
      }
    } else if (patrow) {
      // the pattern's eight values in four 16-byte LDS reads (broadcast), then the products in storage order
      const dbl2* pv = reinterpret_cast<const dbl2*>(rec + CS_HDR + CS_NPAT * CS_PLEN * 4 + pat * (CS_PLEN * 8));
      const dbl2 v01 = pv[0], v23 = pv[1], v45 = pv[2], v67 = pv[3];
      const double v[CS_PLEN] = {v01.x, v01.y, v23.x, v23.y, v45.x, v45.y, v67.x, v67.y};
#pragma unroll
      for (int i = 0; i < CS_PLEN; ++i)
        if (i < plen) {
          asm volatile("" : "+v"(xv[i]));                   // the value the asm load delivered (after the kernel's wait)
          sum += v[i] * xv[i];
        }
    }
  }
};

// diagnostics (MXG_CSP_STAMPS=1): per-phase s_memtime sums of workgroup 0's loader wave and first consumer wave
__device__ unsigned long long g_csp_stamps[16];
__device__ inline unsigned long long csp_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

// NB records per iteration (all their pattern gathers in flight together), ring of D iterations' worth of slots
template <int CH, bool DOT, int EPI, int D, bool HALO = false, int NB = 2, bool STAMP = false>
__global__ __launch_bounds__(CSP_THREADS) void k_spmv_csp(
    const CsDesc* __restrict__ desc, long long nblocks, const unsigned char* __restrict__ stream,
    const int* __restrict__ colidx, const double* __restrict__ vals,
    const double* __restrict__ x, double* __restrict__ y,
    const double* __restrict__ xdot, double* __restrict__ partial,
    const double* __restrict__ rhs, const double* __restrict__ diag, double omega, double beta,
    const double* __restrict__ xg, int nowned) {
  auto xat = [&](int cc) -> double { return (HALO && cc >= nowned) ? xg[cc - nowned] : x[cc]; };
  __shared__ __attribute__((aligned(16))) unsigned char ring[D * NB][CSP_SLOT];
  __shared__ double sred[2][NB][4];
  // an XCD's contiguous eighth of the blocks, in groups of NB consecutive blocks dealt to its workgroups in turn
  const long long W = gridDim.x / MXG_NUM_XCD;
  const long long per = (nblocks + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  const long long xcd = blockIdx.x % MXG_NUM_XCD;
  const long long first = xcd * per + (blockIdx.x / MXG_NUM_XCD) * NB;
  const long long last = min((xcd + 1) * per, nblocks);
  const long long step = W * NB;
  if (first >= last) return;
  const int t = threadIdx.x;
  constexpr int PIECES = NB * (CSP_SLOT / 1024);          // DMA pieces per iteration
  if (t >= 256) {
    // ------------------------------------------------------------------ loader wave
    const int lane = t & 63;
    // group g of this workgroup = blocks g .. g + NB - 1 (those < last); a missing block -> dummy pieces of the
    // stream's first bytes, which keep the vmcnt arithmetic static
    long long so[NB], so_next[NB];
    int by[NB], by_next[NB];
    auto where = [&](long long g, long long* soff, int* bytes) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        soff[j] = 0; bytes[j] = 0;
        if (g + j < last) { soff[j] = desc[g + j].soff; bytes[j] = desc[g + j].rec16 * 16; }
      }
    };
    auto issue = [&](const long long* soff, const int* bytes, int slot0) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const unsigned char* rec = stream + soff[j];
#pragma unroll
        for (int p = 0; p < CSP_SLOT / 1024; ++p) {        // eight pieces per record, all 64 lanes always active
          const int off = 1024 * p + 16 * lane;
          lds_dma16<1>(rec + (off < bytes[j] ? off : 0), ring[slot0 + j] + 1024 * p);
        }
      }
    };
#pragma unroll
    for (int j = 0; j < D - 1; ++j) { where(first + j * step, so, by); issue(so, by, j * NB); }
    where(first + (D - 1) * step, so_next, by_next);
    int it = 0;
    unsigned long long tw = 0, tb = 0, ti = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (long long g = first; g < last; g += step, ++it) {
      if (STAMP) t0 = csp_now();
      csp_wait_pieces<PIECES * (D - 2)>();                  // group `it` has landed, D - 2 younger ones stay in flight
      if (STAMP) t1 = csp_now();
      asm volatile("s_barrier" ::: "memory");               // consumers: group it % D is yours, (it - 1) % D is mine
      if (STAMP) t2 = csp_now();
#pragma unroll
      for (int j = 0; j < NB; ++j) { so[j] = so_next[j]; by[j] = by_next[j]; }
      where(g + D * step, so_next, by_next);              // descriptors of the group after next: an iteration ahead
      issue(so, by, ((it + D - 1) % D) * NB);
      if (STAMP) { t3 = csp_now(); tw += t1 - t0; tb += t2 - t1; ti += t3 - t2; }
    }
    csp_wait_pieces<0>();
    asm volatile("s_barrier" ::: "memory");
    if (STAMP && blockIdx.x == 0 && lane == 0) { g_csp_stamps[0] = tw; g_csp_stamps[1] = tb; g_csp_stamps[2] = ti; g_csp_stamps[3] = (unsigned long long)it; }
    return;
  }
  // -------------------------------------------------------------------- consumer waves: one row per record per thread
  int it = 0;
  long long gprev = -1;
  int rprev[NB];
  bool lprev[NB], fprev[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) { rprev[j] = 0; lprev[j] = false; fprev[j] = false; }
  auto flush = [&](int itp) {                              // finish the previous group's cross-wave sums (thread 0)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (!fprev[j]) continue;
      const double* sp = sred[itp & 1][j];
      const double tot = ((sp[0] + sp[1]) + sp[2]) + sp[3];
      if (lprev[j]) { y[rprev[j]] = cs_epilogue<EPI>(tot, rprev[j], x, rhs, diag, omega, beta, y); if (DOT) partial[gprev + j] = tot * xdot[rprev[j]]; }
      else if (DOT) partial[gprev + j] = tot;
    }
  };
  unsigned long long cb = 0, cg = 0, cw = 0, cf = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (long long g = first; g < last; g += step, ++it) {
    if (STAMP) c0 = csp_now();
    wg_barrier_lds();                                     // the records of group g are in ring[(it % D) * NB ..]
    if (STAMP) c1 = csp_now();
    if (t == 0 && gprev >= 0) flush(it - 1);
    CspRow<HALO> row[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) row[j].head(ring[(it % D) * NB + j], t, g + j < last);
#pragma unroll
    for (int j = 0; j < NB; ++j) row[j].gather(t, x, xg, nowned);
    if (STAMP) c2 = csp_now();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every pattern gather of this thread's NB rows has landed
    if (STAMP) { c3 = csp_now(); cb += c1 - c0; cg += c2 - c1; cw += c3 - c2; }
    double part[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      part[j] = 0.0;
      if (row[j].longrow) {
        // one over-long row: the four waves share it, wave sums are combined in wave order after the next barrier
        const long long base = reinterpret_cast<const long long*>(row[j].rec)[2];
#pragma unroll 1
        for (long long q = base + t; q < base + row[j].nnzg; q += 256) part[j] += vals[q] * xat(colidx[q]);
      } else if (row[j].live) {
        row[j].template finish<CH>(t, colidx, vals, xat);
        const long long r = (long long)row[j].r0 + t;
        __builtin_nontemporal_store(cs_epilogue<EPI>(row[j].sum, r, x, rhs, diag, omega, beta, y), y + r);
        if (DOT) part[j] = row[j].sum * xdot[r];
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool present = g + j < last;
      if (present && (DOT || row[j].longrow)) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part[j] += __shfl_down(part[j], off, 64);
        if ((t & 63) == 0) sred[it & 1][j][t >> 6] = part[j];
      }
      fprev[j] = present && (DOT || row[j].longrow); rprev[j] = row[j].r0; lprev[j] = row[j].longrow;
    }
    gprev = g;
    if (STAMP) cf += csp_now() - c3;
  }
  wg_barrier_lds();
  if (t == 0 && gprev >= 0) flush(it - 1);
  if (STAMP && blockIdx.x == 0 && t == 0) { g_csp_stamps[4] = cb; g_csp_stamps[5] = cg; g_csp_stamps[6] = cw; g_csp_stamps[7] = cf; }
}


// out[0] = sum of partial[0..n) in index order (single workgroup, fixed tree: reproducible)
__global__ void k_sum_partials(const double* __restrict__ partial, long long n, double* __restrict__ out) {
  __shared__ double lds[1024];
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += 1024) s += partial[i];
  lds[threadIdx.x] = s;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) lds[threadIdx.x] += lds[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = lds[0];
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

template <bool CPLX, typename CFG, bool NT, bool XCD, int DIAG = 0>
void launch_stream(mxg_ctx* c, const mxg_csr* a, const MvDesc& X, const MvDesc& Y) {
  const long long per = (a->nblocks + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  const unsigned grid = XCD ? (unsigned)(per * MXG_NUM_XCD) : (unsigned)a->nblocks;
  hipLaunchKernelGGL((k_spmv_stream<CPLX, CFG, NT, XCD, DIAG>), dim3(grid), dim3(CFG::SBLK), 0, c->stream,
                     (const long long*)a->d_rowptr, a->d_colidx, a->d_vals, a->d_rowblk,
                     (long long)a->nblocks, X, Y);
}

}  // namespace

template <typename CFG>
static int build_xwin(mxg_csr* a, const int64_t* rowptr, const int32_t* colidx);
static int build_pipe_desc(mxg_csr* a, int geo, const int64_t* host_rowptr);
// row-gather, one 2048-nnz block per workgroup, gather chunk 7, 16-bit in-block row offsets.  The persistent LDS-DMA
// variant (| 0x2000000, k_spmv_rgp) measures the same at k = 1 (0.400 vs 0.394 ms on one box, 0.3585 vs 0.3761 before the
// branch-free row walk reached this kernel too) and loses at k >= 4 (12 instead of 24 waves per CU); complex matrices
// use it (their one-block kernel is the older stream kernel).
constexpr int MXG_KERNEL_DEFAULT = 0x750000;

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
    a->kernel = MXG_KERNEL_DEFAULT;
  } else if (a->is_complex && nnz > 0) {
    // complex scalars (Bloch phases, lossy dielectrics): the same persistent row-gather kernel over 16-byte values
    MXG_TRY(build_pipe_desc(a, 12, rowptr));
    a->kernel = MXG_KERNEL_DEFAULT | 0x2000000;
  } else if (!a->is_complex) {
    // a rank of a decomposed grid that owns nothing: same kernel family as everybody else (zero row blocks), so that
    // every rank takes the same solver path and issues the same sequence of collectives
    a->kernel = MXG_KERNEL_DEFAULT;
    a->pipe_geo = 10;
    a->npipe = 0;
  }
  *out = a;
  return 0;
}

// A matrix born on the device (mxg_spalg.hip): take ownership of finished CSR arrays (rowptr nrows + 1, colidx / vals
// with 8 entries of slack, columns ascending inside a row) and build what mxg_csr_create builds around them.
int mxg_csr_adopt_device(mxg_ctx* c, int64_t nrows, int64_t ncols, int64_t nnz, int64_t* d_rowptr, int32_t* d_colidx,
                         double* d_vals, mxg_csr** out) {
  MXG_CHECK(nrows < (int64_t)1 << 31 && ncols < (int64_t)1 << 31, "mxg_csr_adopt_device: sizes must fit int32 indices");
  mxg_csr* a = new mxg_csr();
  a->ctx = c;
  a->nrows = nrows; a->ncols_owned = ncols; a->nghost = 0; a->nnz = nnz;
  a->is_complex = 0;
  a->d_rowptr = d_rowptr; a->d_colidx = d_colidx; a->d_vals = d_vals;
  a->kernel = MXG_KERNEL_DEFAULT;
  a->lazy = 1;
  *out = a;
  if (getenv("MXG_LAZY_ADOPT") && atoi(getenv("MXG_LAZY_ADOPT")) == 0) return mxg_csr_materialize(a);
  return 0;
}

int mxg_csr_materialize(mxg_csr* a) {
  if (!a || !a->lazy) return 0;
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
  MXG_HIP(hipMalloc(&a->d_diag, sizeof(double) * std::max<int64_t>(nrows, 1)));
  if (nrows > 0) {
    hipLaunchKernelGGL(k_diag, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream,
                       (const long long*)a->d_rowptr, a->d_colidx, a->d_vals, (long long)nrows, 1, a->d_diag);
    MXG_HIP(hipGetLastError());
  }
  MXG_HIP(hipStreamSynchronize(c->stream));
  if (nnz > 0) MXG_TRY(build_pipe_desc(a, 10, rp.data()));
  else { a->pipe_geo = 10; a->npipe = 0; }
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
  if (a->d_rowpid) hipFree(a->d_rowpid);
  if (a->d_patdict) hipFree(a->d_patdict);
  if (a->d_grows) hipFree(a->d_grows);
  if (a->d_pairid) hipFree(a->d_pairid);
  if (a->d_injcol) hipFree(a->d_injcol);
  if (a->d_ellcol) hipFree(a->d_ellcol);
  if (a->d_ellval) hipFree(a->d_ellval);
  if (a->d_injval) hipFree(a->d_injval);
  if (a->d_tiledesc) hipFree(a->d_tiledesc);
  if (a->d_tiledict) hipFree(a->d_tiledict);
  if (a->d_pairdict) hipFree(a->d_pairdict);
  if (a->d_csdesc) hipFree(a->d_csdesc);
  if (a->d_csstream) hipFree(a->d_csstream);
  if (a->d_xwdesc) hipFree(a->d_xwdesc);
  if (a->d_lcol) hipFree(a->d_lcol);
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

// row blocks depend on the block geometry: rebuild them from the device copy of rowptr
static int rebuild_row_blocks(mxg_csr* a, int nnz_cap, int rows_cap) {
  mxg_ctx* c = a->ctx;
  std::vector<int64_t> rp(a->nrows + 1);
  MXG_HIP(hipMemcpyAsync(rp.data(), a->d_rowptr, sizeof(int64_t) * (a->nrows + 1),
                         hipMemcpyDeviceToHost, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  std::vector<int32_t> blk = build_row_blocks(rp.data(), a->nrows, nnz_cap, rows_cap);
  hipFree(a->d_rowblk);
  MXG_HIP(hipMalloc(&a->d_rowblk, sizeof(int32_t) * blk.size()));
  MXG_HIP(hipMemcpyAsync(a->d_rowblk, blk.data(), sizeof(int32_t) * blk.size(), hipMemcpyHostToDevice,
                         c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  a->nblocks = (int64_t)blk.size() - 1;
  return 0;
}

// ---------------------------------------------------------------------------------------
// setup of the x-window kernel: cluster every block's columns into windows, emit 16-bit
// local column ids.  Runs on the host at matrix creation, spread over hardware threads.
// ---------------------------------------------------------------------------------------
template <typename CFG>
static void build_xwin_range(const int64_t* rowptr, const int32_t* colidx, const std::vector<int32_t>& blk,
                             size_t b0, size_t b1, BlkDescX* desc, unsigned short* lcol, int64_t* unstaged) {
  constexpr int GAP = 64;   // start a new window when consecutive needed columns are further apart
  std::vector<int32_t> u;
  for (size_t b = b0; b < b1; ++b) {
    BlkDescX d{};
    d.r0 = blk[b]; d.nr = blk[b + 1] - blk[b];
    d.base = rowptr[blk[b]];
    d.nnzb = (int)(rowptr[blk[b + 1]] - d.base);
    d.nwin = -1;
    if (d.nnzb > 0 && d.nnzb <= CFG::NNZ_CAP) {
      u.assign(colidx + d.base, colidx + d.base + d.nnzb);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      int nwin = 0, total = 0;
      int lo[XW_MAXWIN], len[XW_MAXWIN], woff[XW_MAXWIN];
      bool ok = true;
      size_t i = 0;
      while (i < u.size()) {
        size_t j = i;
        while (j + 1 < u.size() && u[j + 1] - u[j] <= GAP) ++j;
        if (nwin == XW_MAXWIN) { ok = false; break; }
        const int wlo = u[i] & ~1;
        const int wlen = ((u[j] - wlo + 1) + 1) & ~1;
        lo[nwin] = wlo; len[nwin] = wlen; woff[nwin] = total;
        total += wlen;
        ++nwin;
        if (total > CFG::XCAP) { ok = false; break; }
        i = j + 1;
      }
      if (ok) {
        d.nwin = nwin;
        for (int w = 0; w < nwin; ++w) { d.win_lo[w] = lo[w]; d.win_len[w] = len[w]; }
        for (int64_t q = d.base; q < d.base + d.nnzb; ++q) {
          const int32_t cc = colidx[q];
          int w = 0;
          while (w + 1 < nwin && cc >= lo[w + 1]) ++w;
          lcol[q] = (unsigned short)(woff[w] + (cc - lo[w]));
        }
      } else {
        *unstaged += 1;
      }
    } else if (d.nnzb == 0) {
      d.nwin = 0;
    }
    desc[b] = d;
  }
}

template <typename CFG>
static int build_xwin(mxg_csr* a, const int64_t* rowptr, const int32_t* colidx) {
  mxg_ctx* c = a->ctx;
  std::vector<int32_t> blk = build_row_blocks(rowptr, a->nrows, CFG::NNZ_CAP, CFG::ROWS_CAP);
  const size_t nb = blk.size() - 1;
  std::vector<BlkDescX> desc(std::max<size_t>(nb, 1));
  std::vector<unsigned short> lcol((size_t)a->nnz + 2, 0);
  unsigned nthreads = std::thread::hardware_concurrency();
  if (nthreads > 16) nthreads = 16;
  if (nthreads < 1) nthreads = 1;
  if (nb < 64) nthreads = 1;
  std::vector<int64_t> unstaged(nthreads, 0);
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nthreads; ++t) {
    const size_t b0 = nb * t / nthreads, b1 = nb * (t + 1) / nthreads;
    pool.emplace_back(build_xwin_range<CFG>, rowptr, colidx, std::cref(blk), b0, b1, desc.data(),
                      lcol.data(), &unstaged[t]);
  }
  for (auto& th : pool) th.join();
  a->xw_unstaged = 0;
  for (auto v : unstaged) a->xw_unstaged += v;
  if (a->d_xwdesc) hipFree(a->d_xwdesc);
  if (a->d_lcol) hipFree(a->d_lcol);
  MXG_HIP(hipMalloc(&a->d_xwdesc, sizeof(BlkDescX) * desc.size()));
  MXG_HIP(hipMalloc(&a->d_lcol, sizeof(unsigned short) * lcol.size()));
  MXG_HIP(hipMemcpyAsync(a->d_xwdesc, desc.data(), sizeof(BlkDescX) * desc.size(), hipMemcpyHostToDevice,
                         c->stream));
  MXG_HIP(hipMemcpyAsync(a->d_lcol, lcol.data(), sizeof(unsigned short) * lcol.size(),
                         hipMemcpyHostToDevice, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  a->nxw = (int64_t)nb;
  return 0;
}

// descriptors of the pipelined kernel for block geometry `geo` (0: 256x4, 1: 256x8) and of the
// row-gather kernel (10: 2048 nnz, 11: 3072 nnz, both 256 rows)
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
  int cap = geo == 0 ? PipeCfgA::NNZ_CAP : PipeCfgB::NNZ_CAP;
  int rows_cap = 2 * (geo == 0 ? PipeCfgA::SBLK : PipeCfgB::SBLK);
  if (geo == 10) { cap = RGP_NNZ_CAP; rows_cap = RgCfgA::ROWS_CAP; }   // 2044: the persistent kernel's 16-byte aligned LDS image
  if (geo == 11) { cap = RgCfgB::NNZ_CAP; rows_cap = RgCfgB::ROWS_CAP; }
  if (geo == 12) { cap = RGP_NNZ_CAP_CPLX; rows_cap = RgCfgA::ROWS_CAP; }   // complex matrix, persistent row-gather kernel
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
  if (geo >= 10) {
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
  if (geo == 10 || geo == 12) {
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


// MXG_CSP=1: the persistent record kernel with a loader wave (k_spmv_csp) instead of one block per workgroup (k_spmv_cs).
// Opt-in: at 256^3 it runs 0.24 ms against 0.205 ms - its consumer waves are latency / instruction bound per block,
// not waiting for the record stream (profiles/r02b/csp_*).
static bool csp_wanted() { const char* e = getenv("MXG_CSP"); return e && atoi(e) != 0; }
// MXG_PAT=1: the pattern-dictionary product (mxg_pat.hip).  Opt-in: at 256^3 the scalar Laplacian runs 0.051 ms
// against 0.068 ms, the vector Laplacian 0.225 against 0.202 (its individual wall rows), the eigen-solve 2.63 s against
// 2.44 s - every variant lands near 0.2 ms: the 7 gathers per row, not the matrix bytes, set the time.
// Product kernel of compressed matrices: 3 (default) wavefront-uniform tile patterns where they cover the matrix
// (lattice operators; k_spmv_tile), otherwise and with 0 the compressed stream (k_spmv_cs); 1 row patterns, 2 row pairs
// (mxg_pat.hip; experiments kept for the record, profiles/r02e)
extern "C" int mxg_csr_product_form(const mxg_csr* a, int64_t* tiles, int64_t* patterns, int64_t* irregular_rows, double* coverage) {
  if (!a) return -1;
  const bool tile = (a->kernel & 0x10000000) != 0 && a->d_tiledesc != nullptr;
  if (tiles) *tiles = tile ? a->pat_ntile : 0;
  if (patterns) *patterns = tile ? a->pat_ntilepat : 0;
  if (irregular_rows) *irregular_rows = tile ? a->pat_ngrows : 0;
  if (coverage) *coverage = a->pat_tile_coverage;
  if (tile) return 3;
  return (a->kernel & 0x1000000) != 0 && a->ncs > 0 ? 1 : 0;
}

static int pat_mode() { const char* e = getenv("MXG_PAT"); return e ? atoi(e) : 3; }
static bool pat_wanted() { return pat_mode() != 0; }
static inline bool pat_on(const mxg_csr* a) { return (a->kernel & 0x10000000) != 0 && (a->d_rowpid != nullptr || a->d_pairid != nullptr || a->d_tiledesc != nullptr); }

// Records for the compressed-stream kernels over the row blocks of the row-gather geometry (RgCfgA).
//   [header 32 B][pattern table + per-row pattern ids: patterned records only]
//   [16 band origins][value table][codes][value ids][row offsets]   (the generic part)
// patterns = false builds the records k_spmv_cs (one block per workgroup) understands.
static int build_cs(mxg_csr* a, const int64_t* rp, const int32_t* ci, const double* va, bool patterns) {
  mxg_ctx* c = a->ctx;
  std::vector<int32_t> blk = build_row_blocks(rp, a->nrows, RgCfgA::NNZ_CAP, RgCfgA::ROWS_CAP);
  const size_t nb = blk.size() - 1;
  std::vector<CsDesc> desc(nb);
  std::vector<std::vector<unsigned char> > packs(nb);
  std::vector<int64_t> npatrows(nb, 0);
  auto work = [&](size_t b0, size_t b1) {
    std::vector<int32_t> u, gcol;
    std::vector<uint64_t> w;
    std::vector<double> gval;
    std::vector<unsigned short> groff;
    std::vector<std::pair<std::string, int> > sigs;           // (signature bytes, count) of the block's short rows
    std::vector<int> rowsig;
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
      int hdr_plen = 0;
      auto finish = [&](int nnz_generic) {                   // header + descriptor once the record is complete
        d.nrnv = nr | (nval << 16) | flags;
        int32_t h[8] = {d.r0, d.nrnv, nnz_generic, hdr_plen, 0, 0, 0, 0};
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
      // ---- patterns: the most frequent (column - row, value) sequences among the rows of <= CS_PLEN entries
      std::vector<unsigned char> rowpat;
      int npat = 0;
      int32_t pdelta[CS_NPAT][CS_PLEN] = {};
      double pval[CS_NPAT][CS_PLEN] = {};
      int32_t plen[CS_NPAT] = {};
      if (patterns && d.nnzb > 0 && a->ncols_owned + a->nghost < ((int64_t)1 << 28)) {
        sigs.clear(); rowsig.assign((size_t)nr, -1);
        std::string sg;
        for (int t = 0; t < nr; ++t) {
          const int64_t q0 = rp[d.r0 + t], q1 = rp[d.r0 + t + 1];
          const int len = (int)(q1 - q0);
          if (len == 0 || len > CS_PLEN) continue;
          sg.assign((size_t)len * 12, '\0');
          bool fits = true;
          for (int i = 0; i < len; ++i) {
            const int64_t dl = (int64_t)ci[q0 + i] - (d.r0 + t);
            if (dl >= (1 << 27) || dl <= -(1 << 27)) { fits = false; break; }
            const int32_t dl8 = (int32_t)(dl * 8);             // byte delta: the kernel adds it to 8 * row
            std::memcpy(&sg[(size_t)i * 12], &dl8, 4);
            std::memcpy(&sg[(size_t)i * 12 + 4], &va[q0 + i], 8);
          }
          if (!fits) continue;
          int found = -1;
          for (size_t k = 0; k < sigs.size(); ++k) if (sigs[k].first == sg) { found = (int)k; break; }
          if (found < 0) { if (sigs.size() >= 64) continue; sigs.emplace_back(sg, 0); found = (int)sigs.size() - 1; }
          ++sigs[(size_t)found].second;
          rowsig[(size_t)t] = found;
        }
        // every pattern of a record has the same length (the kernel's loop bound is wave-uniform): the length most
        // rows have
        int64_t bylen[CS_PLEN + 1] = {};
        for (const auto& sgk : sigs) bylen[sgk.first.size() / 12] += sgk.second;
        int best_len = 1;
        for (int l = 2; l <= CS_PLEN; ++l) if (bylen[l] > bylen[best_len]) best_len = l;
        std::vector<int> order;
        for (size_t k = 0; k < sigs.size(); ++k) if ((int)(sigs[k].first.size() / 12) == best_len) order.push_back((int)k);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sigs[(size_t)x].second > sigs[(size_t)y].second; });
        int covered = 0;
        std::vector<int> slot(sigs.size(), -1);
        for (size_t k = 0; k < order.size() && npat < CS_NPAT; ++k) {
          const auto& sgk = sigs[(size_t)order[k]];
          if (sgk.second < 4) break;                          // a pattern must pay for its 96 bytes
          slot[(size_t)order[k]] = npat;
          plen[npat] = (int)(sgk.first.size() / 12);
          for (int i = 0; i < plen[npat]; ++i) {
            std::memcpy(&pdelta[npat][i], &sgk.first[(size_t)i * 12], 4);
            std::memcpy(&pval[npat][i], &sgk.first[(size_t)i * 12 + 4], 8);
          }
          covered += sgk.second;
          ++npat;
        }
        if (npat > 0 && 2 * covered >= nr) {
          rowpat.assign(CS_ROWPAT_BYTES, 0xFF);
          for (int t = 0; t < nr; ++t)
            if (rowsig[(size_t)t] >= 0 && slot[(size_t)rowsig[(size_t)t]] >= 0) rowpat[(size_t)t] = (unsigned char)slot[(size_t)rowsig[(size_t)t]];
          npatrows[b] = covered;
          hdr_plen = best_len;
        } else {
          npat = 0;
        }
      }
      const bool patterned = npat > 0;
      // ---- the generic part: entries of the rows no pattern covers, with row offsets into it
      gcol.clear(); gval.clear(); groff.assign((size_t)nr + 1, 0);
      for (int t = 0; t < nr; ++t) {
        groff[(size_t)t] = (unsigned short)gcol.size();
        if (patterned && rowpat[(size_t)t] != 0xFF) continue;
        for (int64_t q = rp[d.r0 + t]; q < rp[d.r0 + t + 1]; ++q) { gcol.push_back(ci[q]); gval.push_back(va[q]); }
      }
      groff[(size_t)nr] = (unsigned short)gcol.size();
      const int ng = (int)gcol.size();
      const size_t front = (size_t)CS_HDR + (patterned ? (size_t)CS_PAT_BYTES + CS_ROWPAT_BYTES : 0);
      auto write_front = [&]() {
        if (!patterned) return;
        unsigned char* q = pk.data() + CS_HDR;
        std::memcpy(q, pdelta, sizeof pdelta);
        std::memcpy(q + sizeof pdelta, pval, sizeof pval);
        std::memcpy(q + sizeof pdelta + sizeof pval, plen, sizeof plen);
        std::memcpy(q + CS_PAT_BYTES, rowpat.data(), CS_ROWPAT_BYTES);
        flags |= CS_PATTERNED;
      };
      if (patterned && ng == 0) {                            // every row follows a pattern: no generic part at all
        pk.assign(front, 0);
        write_front();
        finish(0);
        continue;
      }
      // rows read from the CSR arrays in place (more bands / distinct values / bytes than the packed form takes):
      // the record carries the row offsets of the CSR arrays themselves
      auto csr_rows = [&]() {
        pk.assign(front + CS_SEGS * 4 + (size_t)cs_roff_bytes(nr), 0);
        write_front();
        unsigned short* ro = reinterpret_cast<unsigned short*>(pk.data() + front + CS_SEGS * 4);
        for (int r = 0; r <= nr; ++r) ro[r] = (unsigned short)(rp[d.r0 + r] - d.base);
        flags |= CS_ROFF_ONLY | (patterned ? CS_HAS_GENERIC : 0);
        nval = 0;
        finish(d.nnzb);
      };
      if (ng == 0) { csr_rows(); continue; }                 // (an empty block: offsets only)
      // column bands: greedy cover of the sorted distinct columns with windows of 2^12
      u.assign(gcol.begin(), gcol.end());
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
      std::memcpy(w.data(), gval.data(), sizeof(double) * (size_t)ng);
      std::sort(w.begin(), w.end());
      w.erase(std::unique(w.begin(), w.end()), w.end());
      if (w.size() > (size_t)CS_MAXVAL) { csr_rows(); continue; }
      nval = (int)w.size();
      const int nvpad = (nval + 1) & ~1;
      const size_t ncode16 = ((size_t)ng + 7) >> 3, nid16 = ((size_t)ng + 15) >> 4;
      const size_t nbytes = front + (size_t)cs_record_bytes(ng, nval, nr, false) - CS_HDR;
      if (nbytes > (size_t)CSP_SLOT) { csr_rows(); continue; }
      pk.assign(nbytes, 0);
      write_front();
      if (patterned) flags |= CS_HAS_GENERIC;
      unsigned char* g = pk.data() + front;
      int32_t segpad[CS_SEGS];
      for (int i = 0; i < CS_SEGS; ++i) segpad[i] = i < nseg ? seg[i] : 0;
      std::memcpy(g, segpad, sizeof segpad);
      std::memcpy(g + CS_SEGS * 4, w.data(), sizeof(uint64_t) * (size_t)nval);
      unsigned short* code = reinterpret_cast<unsigned short*>(g + CS_SEGS * 4 + (size_t)nvpad * 8);
      unsigned char* vid = g + CS_SEGS * 4 + (size_t)nvpad * 8 + ncode16 * 16;
      unsigned short* ro = reinterpret_cast<unsigned short*>(vid + nid16 * 16);
      for (int q = 0; q < ng; ++q) {
        const int32_t col = gcol[(size_t)q];
        const int sidx = (int)(std::upper_bound(seg, seg + nseg, col) - seg) - 1;
        code[q] = (unsigned short)((sidx << CS_OFFBITS) | (col - seg[sidx]));
        uint64_t bits;
        std::memcpy(&bits, &gval[(size_t)q], sizeof bits);
        vid[q] = (unsigned char)(std::lower_bound(w.begin(), w.end(), bits) - w.begin());
      }
      for (int r = 0; r <= nr; ++r) ro[r] = groff[(size_t)r];
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
  int64_t patrows = 0;
  for (size_t b = 0; b < nb; ++b) {
    if (desc[b].nnzb > RgCfgA::NNZ_CAP || (desc[b].roff_only() && desc[b].nnzb > 0)) ++nfall;
    desc[b].soff = (long long)total;
    total += packs[b].size();
    patrows += npatrows[b];
  }
  a->cs_pattern_rows = patrows;
  std::vector<unsigned char> stream(total + 64, 0);      // (a dummy DMA piece of the loader reads bytes 0..16)
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
  MXG_TRY(mxg_csr_materialize(const_cast<mxg_csr*>(a)));
  MXG_CHECK(a, "mxg_csr_compress: NULL argument");
  MXG_CHECK(!a->is_complex, "mxg_csr_compress: real matrices only");
  mxg_ctx* c = a->ctx;
  MXG_HIP(hipSetDevice(c->device));
  if (a->pipe_geo != 10) MXG_TRY(build_pipe_desc(a, 10, rowptr));     // 16-bit row offsets of the same blocks
  const bool want_csp = csp_wanted();
  if ((a->ncs == 0 || a->cs_patterned != (int)want_csp) && a->nrows > 0 && !a->d_tiledesc) {
    std::vector<int64_t> rp;
    std::vector<int32_t> ci;
    std::vector<double> va;
    if (!rowptr || !colidx || !vals) {       // no host copy offered: read the arrays back
      rp.resize((size_t)a->nrows + 1); ci.resize((size_t)a->nnz + 1); va.resize((size_t)a->nnz + 1);
      MXG_HIP(hipMemcpyAsync(rp.data(), a->d_rowptr, sizeof(int64_t) * (a->nrows + 1), hipMemcpyDeviceToHost, c->stream));
      if (a->nnz > 0) {
        MXG_HIP(hipMemcpyAsync(ci.data(), a->d_colidx, sizeof(int32_t) * a->nnz, hipMemcpyDeviceToHost, c->stream));
        MXG_HIP(hipMemcpyAsync(va.data(), a->d_vals, sizeof(double) * a->nnz, hipMemcpyDeviceToHost, c->stream));
      }
      MXG_HIP(hipStreamSynchronize(c->stream));
      rowptr = rp.data(); colidx = ci.data(); vals = va.data();
    }
    // the tile form first (matrices without ghosts; MXG_PAT=0 keeps the record kernels): a lattice operator needs no
    // compressed stream at all - its encoding pass (0.6 s for a 256^3 operator) and 0.5 GB of device memory are saved
    if (pat_mode() == 3 && a->nghost == 0 && a->nneigh == 0) MXG_TRY(mxg_tile_build(a, rowptr, colidx, vals));
    if (!a->d_tiledesc) {
      MXG_TRY(build_cs(a, rowptr, colidx, vals, want_csp));
      a->cs_patterned = (int)want_csp;
      if (pat_mode() == 2 && a->nghost == 0 && a->nneigh == 0 && !a->d_pairid) MXG_TRY(mxg_pair_build(a, rowptr, colidx, vals));
      else if (pat_mode() == 1 && a->nghost == 0 && a->nneigh == 0 && !a->d_rowpid) MXG_TRY(mxg_pat_build(a, rowptr, colidx, vals));
    }
  }
  // 0x8000000: the persistent kernel with a loader wave over patterned records (k_spmv_csp); MXG_CSP=0 at compress
  // time builds the plain records and keeps one block per workgroup (k_spmv_cs)
  a->kernel = MXG_KERNEL_DEFAULT | 0x1000000 | (want_csp ? 0x8000000 : 0) | ((pat_wanted() && (a->d_rowpid || a->d_pairid || a->d_tiledesc)) ? 0x10000000 : 0);
  if (a->d_tiledesc) {                // tile form: work units, no fall-back blocks, bytes of descriptors + patterns
    if (nblocks) *nblocks = a->pat_ntile + (a->pat_ngrows + 255) / 256;     // tiles + workgroups of irregular rows
    if (nfallback) *nfallback = 0;
    if (stream_bytes) *stream_bytes = a->pat_ntile * 16 + a->pat_ntilepat * 256;
    return 0;
  }
  if (nblocks) *nblocks = a->ncs;
  if (nfallback) *nfallback = a->cs_fallback;
  if (stream_bytes) *stream_bytes = a->cs_bytes;
  return 0;
}

extern "C" int mxg_csr_compress(mxg_csr* a, int64_t* nblocks, int64_t* nfallback, int64_t* stream_bytes) {
  MXG_TRY(mxg_csr_materialize(const_cast<mxg_csr*>(a)));
  return mxg_csr_compress_from(a, nullptr, nullptr, nullptr, nblocks, nfallback, stream_bytes);
}

extern "C" int mxg_csr_set_kernel(mxg_csr* a, int which) {
  MXG_TRY(mxg_csr_materialize(const_cast<mxg_csr*>(a)));
  const int old_geo = (a->kernel >> 4) & 3, geo = (which >> 4) & 3;
  a->kernel = which;
  if ((which & 0x30000) && !a->is_complex) {   // row-gather kernel: 0x10000 = 2048 nnz, 0x20000 = 3072 nnz
    const int pg = (which & 0x20000) ? 11 : 10;
    if (a->pipe_geo != pg) MXG_TRY(build_pipe_desc(a, pg, nullptr));
    return 0;
  }
  if ((which & 0xC000) && !a->is_complex) {
    const int xg = (which & 0x8000) ? 1 : 0;
    if (a->xw_geo != xg) {
      mxg_ctx* c = a->ctx;
      std::vector<int64_t> rp(a->nrows + 1);
      std::vector<int32_t> ci((size_t)a->nnz);
      MXG_HIP(hipMemcpyAsync(rp.data(), a->d_rowptr, sizeof(int64_t) * (a->nrows + 1), hipMemcpyDeviceToHost, c->stream));
      MXG_HIP(hipMemcpyAsync(ci.data(), a->d_colidx, sizeof(int32_t) * a->nnz, hipMemcpyDeviceToHost, c->stream));
      MXG_HIP(hipStreamSynchronize(c->stream));
      if (xg == 0) MXG_TRY(build_xwin<XwCfgA>(a, rp.data(), ci.data()));
      else MXG_TRY(build_xwin<XwCfgB>(a, rp.data(), ci.data()));
      a->xw_geo = xg;
    }
    return 0;
  }
  if ((which & 0xC0) && !a->is_complex) {
    const int pg = (which & 0x80) ? 1 : 0;
    if (a->pipe_geo != pg) MXG_TRY(build_pipe_desc(a, pg, nullptr));
    return 0;
  }
  if (geo != old_geo && !a->is_complex) {
    if (geo == 1) return rebuild_row_blocks(a, CfgWide::NNZ_CAP, CfgWide::ROWS_BLK);
    if (geo == 2) return rebuild_row_blocks(a, CfgSmall::NNZ_CAP, CfgSmall::ROWS_BLK);
    if (geo == 3) return rebuild_row_blocks(a, CfgTall::NNZ_CAP, CfgTall::ROWS_BLK);
    return rebuild_row_blocks(a, CfgDefault::NNZ_CAP, CfgDefault::ROWS_BLK);
  }
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
  const int64_t ldg = (a->nghost + 31) & ~(int64_t)31;
  if (k > a->halo_kcap || !a->d_ghost) {
    MXG_HIP(hipStreamSynchronize(c->stream));
    MXG_HIP(hipStreamSynchronize(c->comm_stream));
    if (a->d_sendbuf) hipFree(a->d_sendbuf);
    if (a->d_xext) { hipFree(a->d_xext); a->d_xext = nullptr; }
    if (a->d_ghost) hipFree(a->d_ghost);
    const int kk = std::max(k, a->halo_kcap);
    MXG_HIP(hipMalloc(&a->d_sendbuf, sizeof(double) * std::max<int64_t>(a->nsend, 1) * kk));
    MXG_HIP(hipMalloc(&a->d_ghost, sizeof(double) * std::max<int64_t>(ldg, 32) * kk));
    a->halo_kcap = kk;
    a->ldghost = ldg;
  }
  for (int j = 0; j < k && a->nsend > 0; ++j) {
    const int grid = mxg_grid_for(a->nsend, 256, 1, 4 * c->num_cu);
    hipLaunchKernelGGL(k_pack, dim3(grid), dim3(256), 0, c->stream, x->colptr(j),
                       (const long long*)a->d_send_idx, (long long)a->nsend, 1, a->d_sendbuf + a->nsend * j);
  }
  MXG_HIP(hipGetLastError());
  MXG_CHECK(mxg_comm_on(c), "halo exchange needs a communicator (nranks > 1)");
  MXG_HIP(hipEventRecord(c->ev_pack, c->stream));
  MXG_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
  std::vector<MxgXfer> xf;
  for (int j = 0; j < k; ++j)
    for (int p = 0; p < a->nneigh; ++p)
      xf.push_back({a->neigh[p], a->d_sendbuf + a->nsend * j + a->send_off[p], a->send_off[p + 1] - a->send_off[p],
                    a->d_ghost + ldg * j + a->recv_off[p], a->recv_off[p + 1] - a->recv_off[p]});
  MXG_TRY(mxg_comm_exchange(c, xf.data(), (int)xf.size(), c->comm_stream));
  MXG_HIP(hipEventRecord(c->ev_halo, c->comm_stream));
  return 0;
}

// interior blocks, wait for the halo, boundary blocks (row-gather kernel, real scalars)
template <bool DOT>
static int launch_rowgather_halo(mxg_ctx* c, mxg_csr* a, const MvDesc& X, const MvDesc& Y, const double* xdot,
                                 double* partial) {
  const unsigned short* roff = (a->kernel & 0x40000) ? (const unsigned short*)a->d_roff : (const unsigned short*)nullptr;
  const BlkDesc* desc = (const BlkDesc*)a->d_blkdesc;
  const long long ni = a->n_interior, nb = a->npipe - a->n_interior;
  if (ni > 0) {
    const unsigned grid = (unsigned)(((ni + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD);
    hipLaunchKernelGGL((k_spmv_rowgather<RgCfgA, 7, DOT, false>), dim3(grid), dim3(RgCfgA::SBLK), 0, c->stream, desc, ni,
                       (const long long*)a->d_rowptr, a->d_colidx, a->d_vals, X, Y, xdot, partial, roff,
                       (const double*)nullptr, 0LL, 0);
  }
  MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
  if (nb > 0) {
    const unsigned grid = (unsigned)(((nb + MXG_NUM_XCD - 1) / MXG_NUM_XCD) * MXG_NUM_XCD);
    hipLaunchKernelGGL((k_spmv_rowgather<RgCfgA, 7, DOT, true>), dim3(grid), dim3(RgCfgA::SBLK), 0, c->stream, desc + ni, nb,
                       (const long long*)a->d_rowptr, a->d_colidx, a->d_vals, X, Y, xdot, partial ? partial + ni : nullptr, roff,
                       (const double*)a->d_ghost, (long long)a->ldghost, (int)a->ncols_owned);
  }
  MXG_HIP(hipGetLastError());
  return 0;
}

// persistent compressed-stream kernel: two records per iteration, a ring of two iterations (32 KiB of LDS), four
// 320-thread workgroups per CU
constexpr int CSP_DEPTH = 2, CSP_NB = 2;
static inline bool csp_on(const mxg_csr* a) { return (a->kernel & 0x8000000) != 0; }
template <bool DOT, int EPI, bool HALO = false>
static void launch_csp(mxg_ctx* c, mxg_csr* a, const CsDesc* desc, long long nblocks, const double* x, double* y,
                       const double* xdot, double* partial, const double* rhs, const double* diag, double omega,
                       double beta, const double* xg = nullptr, int nowned = 0) {
  int wgs = 4;
  if (const char* e = getenv("MXG_CSP_WGS")) { const int v = atoi(e); if (v >= 1 && v <= 8) wgs = v; }
  long long grid = std::max(MXG_NUM_XCD, (c->num_cu * wgs / MXG_NUM_XCD) * MXG_NUM_XCD);
  const long long per = (nblocks + MXG_NUM_XCD - 1) / MXG_NUM_XCD;           // blocks per XCD
  grid = std::min(grid, ((per + CSP_NB - 1) / CSP_NB) * MXG_NUM_XCD);        // small matrices: one group per workgroup
  static const bool stamps = getenv("MXG_CSP_STAMPS") != nullptr;
  if (stamps && !DOT && EPI == 0 && !HALO) {
    hipLaunchKernelGGL((k_spmv_csp<7, false, 0, CSP_DEPTH, false, CSP_NB, true>), dim3((unsigned)grid), dim3(CSP_THREADS), 0, c->stream,
                       desc, nblocks, (const unsigned char*)a->d_csstream, (const int*)a->d_colidx,
                       (const double*)a->d_vals, x, y, xdot, partial, rhs, diag, omega, beta, xg, nowned);
    unsigned long long h[16];
    hipStreamSynchronize(c->stream);
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_csp_stamps), sizeof h) == hipSuccess)
      fprintf(stderr, "[csp stamps] loader: wait %llu barrier %llu issue %llu cycles over %llu iterations; consumer: barrier %llu gather-issue %llu gather-wait %llu finish %llu\n",
              h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    return;
  }
  hipLaunchKernelGGL((k_spmv_csp<7, DOT, EPI, CSP_DEPTH, HALO, CSP_NB>), dim3((unsigned)grid), dim3(CSP_THREADS), 0, c->stream,
                     desc, nblocks, (const unsigned char*)a->d_csstream, (const int*)a->d_colidx,
                     (const double*)a->d_vals, x, y, xdot, partial, rhs, diag, omega, beta, xg, nowned);
}

// compressed-stream kernel on a row-partitioned matrix (real, one column): interior blocks, wait for the halo,
// boundary blocks with ghost reads.  halo_start(c, a, x) must have been called.
template <bool DOT, int EPI>
static int launch_cs_halo(mxg_ctx* c, mxg_csr* a, const double* x, double* y, const double* xdot, double* partial,
                          const double* rhs, const double* diag, double omega, double beta = 0.0) {
  const CsDesc* desc = (const CsDesc*)a->d_csdesc;
  const long long ni = a->ncs_interior, nb = a->ncs - a->ncs_interior;
  if (csp_on(a)) {
    // persistent kernel: the blocks that reference no ghost column run while the halo is in flight
    if (ni > 0) launch_csp<DOT, EPI, false>(c, a, desc, ni, x, y, xdot, partial, rhs, diag, omega, beta);
    MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
    if (nb > 0) launch_csp<DOT, EPI, true>(c, a, desc + ni, nb, x, y, xdot, DOT ? partial + ni : partial, rhs, diag, omega, beta,
                                            (const double*)a->d_ghost, (int)a->ncols_owned);
    MXG_HIP(hipGetLastError());
    return 0;
  }
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

static inline bool cs_halo_ok(const mxg_csr* a, const mxg_mv* x, const mxg_mv* y) {
  return (a->kernel & 0x1000000) && !a->is_complex && a->ncs > 0 && a->pipe_geo == 10 && x->k == 1 && y->k == 1 &&
         x->base != y->base && (a->nghost > 0 || a->nneigh > 0);
}

extern "C" int mxg_csr_apply(mxg_ctx* c, const mxg_csr* a_, const mxg_mv* x, mxg_mv* y) {
  MXG_TRY(mxg_csr_materialize(const_cast<mxg_csr*>(a_)));
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
  const bool halo = a->nghost > 0 || a->nneigh > 0;
  if (cs_halo_ok(a, x, y)) {
    MXG_TRY(halo_start(c, a, x));
    if (a->nrows == 0) { MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0)); return 0; }
    (void)hipGetLastError();
    return launch_cs_halo<false, 0>(c, a, x->colptr(0), y->colptr(0), nullptr, nullptr, nullptr, nullptr, 0.0);
  }
  if (halo && !a->is_complex && a->pipe_geo == 10 && a->npipe > 0 && (a->kernel & 0x1030000) == 0x10000 &&
      !(a->kernel & 0xC0C0) && x->base != y->base) {
    // default kernel on a distributed matrix: overlapped ghost import, owned x read in place
    MXG_TRY(halo_start(c, a, x));
    if (a->nrows == 0) { MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0)); return 0; }
    (void)hipGetLastError();
    return launch_rowgather_halo<false>(c, a, X, Y, nullptr, nullptr);
  }
  if (halo) {
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
  (void)hipGetLastError();   // do not attribute a stale sticky error to this launch
  if ((a->kernel & 0x1000000) && !a->is_complex && (a->ncs > 0 || a->d_tiledesc) && a->pipe_geo == 10) {
    const long long per = (a->ncs + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
    const unsigned grid = (unsigned)(per * MXG_NUM_XCD);
    for (int j = 0; j < X.k; ++j)
      if (pat_on(a)) {
        if (mxg_pat_launch(c, a, (const double*)X.col(j), Y.col(j), 0, 0, nullptr, nullptr, nullptr, nullptr, 0.0, 0.0, 0, nullptr)) return 1;
      } else if (csp_on(a))
        launch_csp<false, 0>(c, a, (const CsDesc*)a->d_csdesc, a->ncs, (const double*)X.col(j), Y.col(j), nullptr, nullptr, nullptr, nullptr, 0.0, 0.0);
      else
      hipLaunchKernelGGL((k_spmv_cs<RgCfgA, 7, false>), dim3(grid), dim3(RgCfgA::SBLK / CS_RPT), 0, c->stream,
                         (const CsDesc*)a->d_csdesc, (long long)a->ncs, (const unsigned char*)a->d_csstream,
                         (const unsigned short*)a->d_roff, a->d_colidx, a->d_vals, (const double*)X.col(j), Y.col(j),
                         (const double*)nullptr, (double*)nullptr, (const double*)nullptr, (const double*)nullptr, 0.0,
                         (const double*)nullptr, 0LL, 0.0);
  } else if ((a->kernel & 0x2000000) && !a->is_complex && a->npipe > 0 && a->pipe_geo == 10 && a->d_roff16) {
    // persistent double-buffered row-gather kernel: 3 workgroups (2 x 24.5 KiB of LDS each) per CU
    const unsigned grid = (unsigned)std::max(MXG_NUM_XCD, (c->num_cu * 3 / MXG_NUM_XCD) * MXG_NUM_XCD);
    const int ch = (a->kernel >> 20) & 0xF;
#define MXG_RGP(CHV, AUX, KTV)                                                                       \
    hipLaunchKernelGGL((k_spmv_rgp<CHV, AUX, false, KTV>), dim3(grid), dim3(256), 0, c->stream, (const BlkDesc*)a->d_blkdesc, \
                       (long long)a->npipe, (const int*)a->d_colidx, (const double*)a->d_vals,          \
                       (const unsigned short*)a->d_roff16, (const double*)X.base + X.ld * xc0, (long long)X.ld, \
                       Y.base + Y.ld * yc0, (long long)Y.ld, kk)
    // contiguous column ranges of both views (one launch for a plain block of vectors)
    for (int j0 = 0; j0 < X.k;) {
      int kk = 1;
      while (j0 + kk < X.k && x->hcols[j0 + kk] == x->hcols[j0] + kk && y->hcols[j0 + kk] == y->hcols[j0] + kk) ++kk;
      const long long xc0 = X.base == x->base ? x->hcols[j0] : j0, yc0 = y->hcols[j0];
      if (a->kernel & 0x4000000) { if (ch == 4) MXG_RGP(4, 0, 1); else if (ch == 7) MXG_RGP(7, 0, 1); else MXG_RGP(8, 0, 1); }
      else if (kk >= 4 && ch == 7) MXG_RGP(7, 2, 4);
      else if (kk >= 2 && ch == 7) MXG_RGP(7, 2, 2);
      else { if (ch == 4) MXG_RGP(4, 2, 1); else if (ch == 7) MXG_RGP(7, 2, 1); else MXG_RGP(8, 2, 1); }
      j0 += kk;
    }
#undef MXG_RGP
  } else if ((a->kernel & 0x30000) && !a->is_complex && a->npipe > 0) {
    const long long per = (a->npipe + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
    const unsigned grid = (unsigned)(per * MXG_NUM_XCD);
    const int ch = (a->kernel >> 20) & 0xF;   // gather chunk: 0 -> 8
#define MXG_RG(CFG, CHV) MXG_RGK(CFG, CHV, 1)
#define MXG_RGK(CFG, CHV, KTV)                                                                     \
    hipLaunchKernelGGL((k_spmv_rowgather<CFG, CHV, false, false, KTV>), dim3(grid), dim3(CFG::SBLK), 0, c->stream, \
                       (const BlkDesc*)a->d_blkdesc, (long long)a->npipe, (const long long*)a->d_rowptr, \
                       a->d_colidx, a->d_vals, X, Y, (const double*)nullptr, (double*)nullptr,       \
                       (a->kernel & 0x40000) ? (const unsigned short*)a->d_roff : (const unsigned short*)nullptr)
    // (column tiles - KT = 2 / 4 gathers of a chunk in flight per column tile - measured slower here: 0.566 / 0.780 /
    //  1.344 ms against 0.516 / 0.748 / 1.206 ms at k = 2 / 4 / 8, the registers cost a block per CU; MXG_SPMM_TILE=1)
    static const bool tile = getenv("MXG_SPMM_TILE") != nullptr;
    if (tile && a->pipe_geo == 10 && ch == 7 && X.k >= 4) MXG_RGK(RgCfgA, 7, 4);
    else if (tile && a->pipe_geo == 10 && ch == 7 && X.k >= 2) MXG_RGK(RgCfgA, 7, 2);
    else if (a->pipe_geo == 10) { if (ch == 4) MXG_RG(RgCfgA, 4); else if (ch == 7) MXG_RG(RgCfgA, 7); else MXG_RG(RgCfgA, 8); }
    else { if (ch == 4) MXG_RG(RgCfgB, 4); else if (ch == 7) MXG_RG(RgCfgB, 7); else MXG_RG(RgCfgB, 8); }
#undef MXG_RG
#undef MXG_RGK
  } else if ((a->kernel & 0xC000) && !a->is_complex && a->nxw > 0) {
    const long long per = (a->nxw + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
    const unsigned grid = (unsigned)(per * MXG_NUM_XCD);
    if (a->xw_geo == 0)
      hipLaunchKernelGGL((k_spmv_xwin<XwCfgA>), dim3(grid), dim3(XwCfgA::SBLK), 0, c->stream,
                         (const BlkDescX*)a->d_xwdesc, (long long)a->nxw, (const long long*)a->d_rowptr,
                         a->d_colidx, a->d_lcol, a->d_vals, X, Y);
    else
      hipLaunchKernelGGL((k_spmv_xwin<XwCfgB>), dim3(grid), dim3(XwCfgB::SBLK), 0, c->stream,
                         (const BlkDescX*)a->d_xwdesc, (long long)a->nxw, (const long long*)a->d_rowptr,
                         a->d_colidx, a->d_lcol, a->d_vals, X, Y);
  } else if ((a->kernel & 0xC0) && !a->is_complex && X.k == 1 && a->npipe > 0) {
    int bpc = (a->kernel >> 8) & 0xF;
    if (bpc == 0) bpc = 8;
    long long grid = (long long)c->num_cu * bpc;
    grid = (grid / MXG_NUM_XCD) * MXG_NUM_XCD;
    if (grid < MXG_NUM_XCD) grid = MXG_NUM_XCD;
    const double* xp = X.col(0);
    double* yp = Y.col(0);
    if (a->pipe_geo == 0)
      hipLaunchKernelGGL((k_spmv_pipe<PipeCfgA>), dim3((unsigned)grid), dim3(PipeCfgA::SBLK), 0, c->stream,
                         (const BlkDesc*)a->d_blkdesc, (long long)a->npipe, (const long long*)a->d_rowptr,
                         a->d_colidx, a->d_vals, xp, yp);
    else
      hipLaunchKernelGGL((k_spmv_pipe<PipeCfgB>), dim3((unsigned)grid), dim3(PipeCfgB::SBLK), 0, c->stream,
                         (const BlkDesc*)a->d_blkdesc, (long long)a->npipe, (const long long*)a->d_rowptr,
                         a->d_colidx, a->d_vals, xp, yp);
  } else if ((a->kernel & 1) && !a->is_complex) {
    const long long threads = (long long)a->nrows * 8;
    hipLaunchKernelGGL((k_spmv_vector<8>), dim3((unsigned)((threads + SBLK - 1) / SBLK)), dim3(SBLK),
                       0, c->stream, (const long long*)a->d_rowptr, a->d_colidx, a->d_vals,
                       (long long)a->nrows, X, Y);
  } else if (a->is_complex && (a->kernel & 0x2000000) && a->pipe_geo == 12 && a->npipe > 0 && a->d_roff16 && !halo) {
    const unsigned grid = (unsigned)std::max(MXG_NUM_XCD, (c->num_cu * 3 / MXG_NUM_XCD) * MXG_NUM_XCD);
    for (int j0 = 0; j0 < X.k;) {
      int kk = 1;
      while (j0 + kk < X.k && x->hcols[j0 + kk] == x->hcols[j0] + kk && y->hcols[j0 + kk] == y->hcols[j0] + kk) ++kk;
      const long long xc0 = X.base == x->base ? x->hcols[j0] : j0, yc0 = y->hcols[j0];
      hipLaunchKernelGGL((k_spmv_rgp<7, 2, true, 1>), dim3(grid), dim3(256), 0, c->stream, (const BlkDesc*)a->d_blkdesc,
                         (long long)a->npipe, (const int*)a->d_colidx, (const double*)a->d_vals,
                         (const unsigned short*)a->d_roff16, (const double*)X.base + X.ld * xc0, (long long)X.ld,
                         Y.base + Y.ld * yc0, (long long)Y.ld, kk);
      j0 += kk;
    }
  } else if (a->is_complex) {
    launch_stream<true, CfgDefault, true, true>(c, a, X, Y);
  } else if (a->kernel & 0x3000) {
    if ((a->kernel & 0x3000) == 0x1000) launch_stream<false, CfgDefault, true, true, 1>(c, a, X, Y);
    else launch_stream<false, CfgDefault, true, true, 2>(c, a, X, Y);
  } else {
    // tuning variants (mxg_csr_set_kernel): bit 1 = plain loads, bit 2 = no XCD remap,
    // bits 4-5 = block geometry
    const bool nt = !(a->kernel & 2), xcd = !(a->kernel & 4);
    const int geo = (a->kernel >> 4) & 3;
#define MXG_LAUNCH(CFG)                                                        \
    do {                                                                       \
      if (nt && xcd) launch_stream<false, CFG, true, true>(c, a, X, Y);        \
      else if (nt) launch_stream<false, CFG, true, false>(c, a, X, Y);         \
      else if (xcd) launch_stream<false, CFG, false, true>(c, a, X, Y);        \
      else launch_stream<false, CFG, false, false>(c, a, X, Y);                \
    } while (0)
    if (geo == 1) MXG_LAUNCH(CfgWide);
    else if (geo == 2) MXG_LAUNCH(CfgSmall);
    else if (geo == 3) MXG_LAUNCH(CfgTall);
    else MXG_LAUNCH(CfgDefault);
#undef MXG_LAUNCH
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
  if (x->k != 1 || y->k != 1 || x->base == y->base || (getenv("MXG_ELL") && atoi(getenv("MXG_ELL")) == 0)) return 1;
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
  if (epi == 3 && x->k == 1 && y->k == 1 && x->base != y->base && !(getenv("MXG_INJECT") && atoi(getenv("MXG_INJECT")) == 0)) {
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
  if (!((a->kernel & 0x1000000) && !a->is_complex && (a->ncs > 0 || a->d_tiledesc) && a->pipe_geo == 10 && a->nghost == 0 &&
        a->nneigh == 0 && x->k == 1 && y->k == 1 && x->base != y->base))
    return 1;
  const MvDesc X = x->desc(), Y = y->desc();
  const long long per = (a->ncs + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  const unsigned grid = (unsigned)(per * MXG_NUM_XCD);
  (void)hipGetLastError();
#define MXG_CS_EPI(E)                                                                                          \
  hipLaunchKernelGGL((k_spmv_cs<RgCfgA, 7, false, E>), dim3(grid), dim3(RgCfgA::SBLK / CS_RPT), 0, c->stream,     \
                     (const CsDesc*)a->d_csdesc, (long long)a->ncs, (const unsigned char*)a->d_csstream,        \
                     (const unsigned short*)a->d_roff, a->d_colidx, a->d_vals, (const double*)X.col(0), Y.col(0), \
                     (const double*)nullptr, (double*)nullptr, rhs, diag, omega, (const double*)nullptr, 0LL, beta)
  if (pat_on(a)) {
    if (mxg_pat_launch(c, a, (const double*)X.col(0), Y.col(0), epi, 0, nullptr, nullptr, rhs, diag, omega, beta, 0, nullptr)) return 2;
    return 0;
  }
  if (csp_on(a)) {
    const double* xp = X.col(0);
    double* yp = Y.col(0);
    if (epi == 1) launch_csp<false, 1>(c, a, (const CsDesc*)a->d_csdesc, a->ncs, xp, yp, nullptr, nullptr, rhs, diag, omega, beta);
    else if (epi == 2) launch_csp<false, 2>(c, a, (const CsDesc*)a->d_csdesc, a->ncs, xp, yp, nullptr, nullptr, rhs, diag, omega, beta);
    else if (epi == 3) launch_csp<false, 3>(c, a, (const CsDesc*)a->d_csdesc, a->ncs, xp, yp, nullptr, nullptr, rhs, diag, omega, beta);
    else if (epi == 4) launch_csp<false, 4>(c, a, (const CsDesc*)a->d_csdesc, a->ncs, xp, yp, nullptr, nullptr, rhs, diag, omega, beta);
    else if (epi == 5) launch_csp<false, 5>(c, a, (const CsDesc*)a->d_csdesc, a->ncs, xp, yp, nullptr, nullptr, rhs, diag, omega, beta);
    else launch_csp<false, 0>(c, a, (const CsDesc*)a->d_csdesc, a->ncs, xp, yp, nullptr, nullptr, rhs, diag, omega, beta);
  } else
  if (epi == 1) MXG_CS_EPI(1); else if (epi == 2) MXG_CS_EPI(2); else if (epi == 3) MXG_CS_EPI(3); else if (epi == 4) MXG_CS_EPI(4);
  else if (epi == 5) MXG_CS_EPI(5); else MXG_CS_EPI(0);
#undef MXG_CS_EPI
  if (hipGetLastError() != hipSuccess) { mxg_set_error("mxg_dev_csr_apply_epi: launch failed"); return 2; }
  return 0;
}

// y = A x and d_out[0] = x . y in one pass over the matrix (real, one column, row-gather geometry).
// x . y uses the owned rows only; the caller all-reduces d_out across ranks if needed.
int mxg_dev_csr_apply_dot(mxg_ctx* c, mxg_csr* a, const mxg_mv* x, mxg_mv* y, double* d_out) {
  MXG_TRY(mxg_csr_materialize(const_cast<mxg_csr*>(a)));
  MXG_CHECK(!a->is_complex && x->k == 1 && y->k == 1 && a->nrows == a->ncols_owned,
            "mxg_dev_csr_apply_dot: real square single-column case only");
  MXG_CHECK(a->pipe_geo == 10, "mxg_dev_csr_apply_dot: row-gather descriptors missing");
  if (a->npipe == 0) {              // no rows on this rank: take part in the ghost import (nothing to send), x . y = 0
    if (a->nneigh > 0) { MXG_TRY(halo_start(c, a, x)); MXG_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0)); }
    MXG_HIP(hipMemsetAsync(d_out, 0, sizeof(double), c->stream));
    return 0;
  }
  MvDesc X = x->desc();
  const MvDesc Y = y->desc();
  if (!a->d_dotpartial || a->dotpartial_cap < a->npipe) {
    if (a->d_dotpartial) hipFree(a->d_dotpartial);
    MXG_HIP(hipMalloc(&a->d_dotpartial, sizeof(double) * a->npipe));
    a->dotpartial_cap = a->npipe;
  }
  if (a->nghost > 0 || a->nneigh > 0) {
    MXG_TRY(halo_start(c, a, x));
    (void)hipGetLastError();
    if (cs_halo_ok(a, x, y) && a->ncs == a->npipe)
      MXG_TRY((launch_cs_halo<true, 0>(c, a, x->colptr(0), y->colptr(0), (const double*)x->colptr(0), a->d_dotpartial,
                                       nullptr, nullptr, 0.0)));
    else
    MXG_TRY((launch_rowgather_halo<true>(c, a, X, Y, (const double*)x->colptr(0), a->d_dotpartial)));
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, c->stream, (const double*)a->d_dotpartial,
                       (long long)a->npipe, d_out);
    MXG_HIP(hipGetLastError());
    return 0;
  }
  const long long per = (a->npipe + MXG_NUM_XCD - 1) / MXG_NUM_XCD;
  const unsigned grid = (unsigned)(per * MXG_NUM_XCD);
  (void)hipGetLastError();
  if ((a->kernel & 0x1000000) && pat_on(a)) {
    const int64_t np = mxg_pat_partials(a);
    if (a->dotpartial_cap < np) {
      if (a->d_dotpartial) hipFree(a->d_dotpartial);
      MXG_HIP(hipMalloc(&a->d_dotpartial, sizeof(double) * np));
      a->dotpartial_cap = np;
    }
    if (mxg_pat_launch(c, a, (const double*)X.col(0), Y.col(0), 0, 1, (const double*)x->colptr(0), a->d_dotpartial, nullptr, nullptr, 0.0, 0.0, 0, nullptr)) return 1;
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, c->stream, (const double*)a->d_dotpartial, (long long)np, d_out);
    MXG_HIP(hipGetLastError());
    return 0;
  }
  if ((a->kernel & 0x1000000) && a->ncs == a->npipe && csp_on(a))
    launch_csp<true, 0>(c, a, (const CsDesc*)a->d_csdesc, a->ncs, (const double*)X.col(0), Y.col(0), (const double*)x->colptr(0), a->d_dotpartial, nullptr, nullptr, 0.0, 0.0);
  else if ((a->kernel & 0x1000000) && a->ncs == a->npipe)
    hipLaunchKernelGGL((k_spmv_cs<RgCfgA, 7, true>), dim3(grid), dim3(RgCfgA::SBLK / CS_RPT), 0, c->stream,
                       (const CsDesc*)a->d_csdesc, (long long)a->ncs, (const unsigned char*)a->d_csstream,
                       (const unsigned short*)a->d_roff, a->d_colidx, a->d_vals, (const double*)X.col(0), Y.col(0),
                       (const double*)x->colptr(0), a->d_dotpartial, (const double*)nullptr, (const double*)nullptr, 0.0,
                       (const double*)nullptr, 0LL, 0.0);
  else
  hipLaunchKernelGGL((k_spmv_rowgather<RgCfgA, 7, true>), dim3(grid), dim3(RgCfgA::SBLK), 0, c->stream,
                     (const BlkDesc*)a->d_blkdesc, (long long)a->npipe, (const long long*)a->d_rowptr,
                     a->d_colidx, a->d_vals, X, Y, (const double*)x->colptr(0), a->d_dotpartial,
                     (a->kernel & 0x40000) ? (const unsigned short*)a->d_roff : (const unsigned short*)nullptr);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, c->stream, (const double*)a->d_dotpartial,
                     (long long)a->npipe, d_out);
  MXG_HIP(hipGetLastError());
  return 0;
}
