// Multivector storage and the block BLAS-1 / tall-skinny BLAS-3 kernels of the Krylov basis.
// Replaces Epetra_MultiVector behind MxMultiVector / MxAnasaziMV (reference
// src/MxMultiVector.cpp, src/MxAnasaziMV.cpp; call sites K2-K9 of SURVEY section 2.2).
//
// All kernels are HBM-bound: 16-byte loads where the layout allows, thread-private
// accumulation over a grid-stride loop, one wavefront shuffle reduction + one LDS pass per
// block at the end, and a fixed-order second stage so results are run-to-run reproducible.
#include "mxg_internal.h"

namespace {

constexpr int BLK = 256;

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block-wide sum of one value per thread; result valid in thread 0
__device__ inline double block_sum(double v, double* lds /* BLK/64 doubles */) {
  v = wave_sum(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < BLK / 64; ++i) r += lds[i];
  }
  return r;
}

__global__ void k_reduce_final(const double* __restrict__ partial, int nblocks, double* out) {
  __shared__ double lds[BLK / 64];
  const double* p = partial + (size_t)blockIdx.x * MXG_MAX_RED_BLOCKS;
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += BLK) s += p[i];
  s = block_sum(s, lds);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// ---------------------------------------------------------------- elementwise
// modes of the generic elementwise kernel (real data handled as raw doubles,
// complex data as (re, im) pairs)
enum EwOp { EW_FILL, EW_SCALE, EW_SCALE_COLS, EW_CONJ, EW_MASK, EW_AXPBY, EW_ADD, EW_COPY };

struct EwArgs {
  MvDesc y;       // destination
  MvDesc a;       // first source (AXPBY/ADD/COPY/MASK)
  MvDesc b;       // second source (ADD)
  double ar, ai;  // scalar a
  double br, bi;  // scalar b
  const double* alphas;  // per-column scalars (SCALE_COLS), device
  const int* ycols;      // optional destination column remap (set_block), device
};

template <int OP, bool CPLX>
__global__ void k_elementwise(EwArgs g) {
  const int j = blockIdx.y;
  const long long n = g.y.n;  // scalar rows
  double* y = g.ycols ? g.y.base + (long long)g.ycols[j] * g.y.ld : g.y.col(j);
  const double* a = nullptr;
  const double* b = nullptr;
  if (OP == EW_AXPBY || OP == EW_ADD || OP == EW_COPY) a = g.a.col(j);
  if (OP == EW_MASK) a = g.a.col(0);
  if (OP == EW_ADD) b = g.b.col(j);
  double sr = g.ar, si = g.ai;
  if (OP == EW_SCALE_COLS) {
    sr = g.alphas[CPLX ? 2 * j : j];
    si = CPLX ? g.alphas[2 * j + 1] : 0.0;
  }
  // every element is a pair of doubles: two independent reals, or one complex number
  const long long npair = CPLX ? n : n / 2;
  double2* y2 = reinterpret_cast<double2*>(y);
  const double2* a2 = reinterpret_cast<const double2*>(a);
  const double2* b2 = reinterpret_cast<const double2*>(b);
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < npair; i += stride) {
    double2 r;
    if (OP == EW_FILL) {
      r.x = sr; r.y = CPLX ? si : sr;
    } else if (OP == EW_SCALE || OP == EW_SCALE_COLS) {
      double2 v = y2[i];
      if (CPLX) { r.x = sr * v.x - si * v.y; r.y = sr * v.y + si * v.x; }
      else { r.x = sr * v.x; r.y = sr * v.y; }
    } else if (OP == EW_CONJ) {
      double2 v = y2[i]; r.x = v.x; r.y = -v.y;
    } else if (OP == EW_MASK) {
      double2 v = y2[i];
      if (CPLX) { double m = a[i]; r.x = v.x * m; r.y = v.y * m; }
      else { double2 m = a2[i]; r.x = v.x * m.x; r.y = v.y * m.y; }
    } else if (OP == EW_COPY) {
      r = a2[i];
    } else if (OP == EW_AXPBY) {
      double2 va = a2[i];
      const bool useb = (g.br != 0.0) || (CPLX && g.bi != 0.0);
      double2 vy = useb ? y2[i] : make_double2(0.0, 0.0);
      if (CPLX) {
        r.x = (g.ar * va.x - g.ai * va.y) + (g.br * vy.x - g.bi * vy.y);
        r.y = (g.ar * va.y + g.ai * va.x) + (g.br * vy.y + g.bi * vy.x);
      } else {
        r.x = g.ar * va.x + g.br * vy.x; r.y = g.ar * va.y + g.br * vy.y;
      }
      if (!useb) {
        if (CPLX) { r.x = g.ar * va.x - g.ai * va.y; r.y = g.ar * va.y + g.ai * va.x; }
        else { r.x = g.ar * va.x; r.y = g.ar * va.y; }
      }
    } else {  // EW_ADD
      double2 va = a2[i], vb = b2[i];
      if (CPLX) {
        r.x = (g.ar * va.x - g.ai * va.y) + (g.br * vb.x - g.bi * vb.y);
        r.y = (g.ar * va.y + g.ai * va.x) + (g.br * vb.y + g.bi * vb.x);
      } else {
        r.x = g.ar * va.x + g.br * vb.x; r.y = g.ar * va.y + g.br * vb.y;
      }
    }
    y2[i] = r;
  }
  // odd tail of a real column
  if (!CPLX && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const long long i = n - 1;
    double r;
    if (OP == EW_FILL) r = sr;
    else if (OP == EW_SCALE || OP == EW_SCALE_COLS) r = sr * y[i];
    else if (OP == EW_CONJ) r = y[i];
    else if (OP == EW_MASK) r = y[i] * a[i];
    else if (OP == EW_COPY) r = a[i];
    else if (OP == EW_AXPBY) r = (g.br != 0.0) ? g.ar * a[i] + g.br * y[i] : g.ar * a[i];
    else r = g.ar * a[i] + g.br * b[i];
    y[i] = r;
  }
}

template <int OP>
int launch_ew(mxg_ctx* c, const EwArgs& g) {
  const long long n = g.y.n;
  if (n == 0 || g.y.k == 0) return 0;
  const long long npair = g.y.cplx ? n : (n + 1) / 2;
  dim3 grid(mxg_grid_for(npair, BLK, 4, 8 * c->num_cu), g.y.k);
  if (g.y.cplx) hipLaunchKernelGGL((k_elementwise<OP, true>), grid, dim3(BLK), 0, c->stream, g);
  else hipLaunchKernelGGL((k_elementwise<OP, false>), grid, dim3(BLK), 0, c->stream, g);
  MXG_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- dot products (K2/K3)
// partial[(2j+c) * MAXB + blockIdx.x]; c = 0 real part, 1 imaginary part (complex only).  Four 16-byte loads per
// operand in flight per thread.  SELF: norm2, the operand is read once.  The second stage (k_reduce_final) adds the
// partials in block order and stores the result - straight into the pinned host buffer when the caller wants it there.
// (A fused "last block adds the partials" variant was measured and dropped: the device-scope fence each block needs
// before it signals writes back / invalidates the XCD's L2 and doubled the kernel time, profiles/r02c.)
template <bool CPLX, bool SELF>
__global__ void __launch_bounds__(BLK) k_dot(MvDesc A, MvDesc B, double* __restrict__ partial) {
  __shared__ double lds[BLK / 64];
  const int j = blockIdx.y;
  const double* a = A.col(j);
  const double* b = B.col(j);
  const long long n = A.n;
  const long long npair = CPLX ? n : n / 2;
  const double2* a2 = reinterpret_cast<const double2*>(a);
  const double2* b2 = reinterpret_cast<const double2*>(b);
  double sr = 0.0, si = 0.0;
  const long long stride = (long long)gridDim.x * BLK;
  long long i = (long long)blockIdx.x * BLK + threadIdx.x;
  constexpr int U = 4;
  for (; i + (U - 1) * stride < npair; i += U * stride) {
    double2 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      va[u] = a2[i + u * stride];
      if (!SELF) vb[u] = b2[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double2 x = va[u], y = SELF ? va[u] : vb[u];
      sr += x.x * y.x + x.y * y.y;
      if (CPLX && !SELF) si += x.x * y.y - x.y * y.x;
    }
  }
  for (; i < npair; i += stride) {
    const double2 x = a2[i], y = SELF ? x : b2[i];
    sr += x.x * y.x + x.y * y.y;
    if (CPLX && !SELF) si += x.x * y.y - x.y * y.x;
  }
  if (!CPLX && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) sr += a[n - 1] * b[n - 1];
  const int width = CPLX ? 2 : 1;
  double r = block_sum(sr, lds);
  if (threadIdx.x == 0) partial[(size_t)(width * j) * MXG_MAX_RED_BLOCKS + blockIdx.x] = r;
  if (CPLX) {
    double q = block_sum(si, lds);
    if (threadIdx.x == 0) partial[(size_t)(2 * j + 1) * MXG_MAX_RED_BLOCKS + blockIdx.x] = q;
  }
}

// ---------------------------------------------------------------- B = [A A2]^H T (K7)
// One pass over the rows for every m: all four wavefronts of a block walk the same rows, wavefront w owns the LT
// columns of [A A2] of tile w (then w + 4, ...) and all KT columns of T - the columns of T are fetched from HBM once
// per 4 LT columns and served to the other three wavefronts by the caches, every column of A exactly once.  Thread-
// private LT x KT accumulators over a grid-stride loop, one wavefront shuffle reduction per accumulator at the end
// (no block-level step: the wavefronts own different entries).  A2 (m2 columns, may be 0) is appended after A: the
// Krylov-Schur step gets V^H w and w^H w from one launch.  partial[((l) + m*(j0+j)) * width + c] per block.
template <int LT, int KT, bool CPLX>
__global__ void __launch_bounds__(BLK) k_trans_mv(MvDesc A, MvDesc A2, MvDesc T, int j0, int m, double* __restrict__ partial) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long n = T.n;
  const int m1 = A.k;
  const double* tp[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) tp[j] = T.col(min(j0 + j, T.k - 1));
  const long long stride = (long long)gridDim.x * 64;
  const int width = CPLX ? 2 : 1;
  const long long nelem = CPLX ? n : n / 2;
  for (int l0 = wave * LT; l0 < m; l0 += (BLK / 64) * LT) {
    const double* ap[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      const int ll = min(l0 + l, m - 1);
      ap[l] = ll < m1 ? A.col(ll) : A2.col(ll - m1);
    }
    double acc[LT][KT][CPLX ? 2 : 1];
#pragma unroll
    for (int l = 0; l < LT; ++l)
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        acc[l][j][0] = 0.0;
        if (CPLX) acc[l][j][1] = 0.0;
      }
    for (long long i = (long long)blockIdx.x * 64 + lane; i < nelem; i += stride) {
      double2 tv[KT], av[LT];
#pragma unroll
      for (int j = 0; j < KT; ++j) tv[j] = reinterpret_cast<const double2*>(tp[j])[i];
#pragma unroll
      for (int l = 0; l < LT; ++l) av[l] = reinterpret_cast<const double2*>(ap[l])[i];
#pragma unroll
      for (int l = 0; l < LT; ++l)
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          if (CPLX) {
            acc[l][j][0] += av[l].x * tv[j].x + av[l].y * tv[j].y;
            acc[l][j][1] += av[l].x * tv[j].y - av[l].y * tv[j].x;
          } else {
            acc[l][j][0] += av[l].x * tv[j].x + av[l].y * tv[j].y;
          }
        }
    }
    if (!CPLX && (n & 1) && blockIdx.x == 0 && lane == 0) {
#pragma unroll
      for (int l = 0; l < LT; ++l)
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[l][j][0] += ap[l][n - 1] * tp[j][n - 1];
    }
#pragma unroll
    for (int l = 0; l < LT; ++l)
#pragma unroll
      for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int c = 0; c < width; ++c) {
          const double r = wave_sum(acc[l][j][c]);
          if (lane == 0 && l0 + l < m && j0 + j < T.k) {
            const size_t slot = ((size_t)(l0 + l) + (size_t)m * (j0 + j)) * width + c;
            partial[slot * MXG_MAX_RED_BLOCKS + blockIdx.x] = r;
          }
        }
  }
}

// ---------------------------------------------------------------- Y = alpha A B + beta Y (K6)
// One thread per element (a complex number, or two consecutive real rows: 16-byte accesses either way), KT
// accumulators, a single pass over A and Y for every m.  The m x KT coefficients (already scaled by alpha) sit in
// LDS, zero-padded to a multiple of UL rows, and are read as broadcasts; UL columns of A are loaded before their
// FMAs so that UL 16-byte loads per thread are in flight (the loop over m used to issue one load at a time).
constexpr int TM_UL = 6;
constexpr int TM_LDS_DOUBLES = 6144;   // m * KT * (1 or 2) coefficients; larger products read them from global memory

template <int KT, bool CPLX, bool INLDS>
__global__ void __launch_bounds__(BLK) k_times_mat(MvDesc A, const double* __restrict__ Bs, int m, int j0, int kcols,
                                                    double br, double bi, MvDesc Y) {
  constexpr int W = CPLX ? 2 : 1;
  __shared__ double sB[INLDS ? TM_LDS_DOUBLES : 1];
  const int mpad = (m + TM_UL - 1) / TM_UL * TM_UL;
  if (INLDS) {
    for (int q = threadIdx.x; q < mpad * KT * W; q += BLK) {
      const int c = q % W, j = (q / W) % KT, l = q / (W * KT);
      const int jj = min(j0 + j, kcols - 1);
      sB[q] = l < m ? Bs[((size_t)l + (size_t)m * jj) * W + c] : 0.0;
    }
    __syncthreads();
  }
  const long long n = Y.n;
  const long long nelem = CPLX ? n : n / 2;
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < nelem; i += stride) {
    double2 acc[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[j] = make_double2(0.0, 0.0);
    for (int l0 = 0; l0 < m; l0 += TM_UL) {
      double2 av[TM_UL];
#pragma unroll
      for (int u = 0; u < TM_UL; ++u) av[u] = reinterpret_cast<const double2*>(A.col(min(l0 + u, m - 1)))[i];
#pragma unroll
      for (int u = 0; u < TM_UL; ++u) {
        const int l = l0 + u;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          double bre, bim = 0.0;
          if (INLDS) {
            bre = sB[(l * KT + j) * W];
            if (CPLX) bim = sB[(l * KT + j) * W + 1];
          } else {
            const int jj = min(j0 + j, kcols - 1);
            const bool in = l < m;
            bre = in ? Bs[((size_t)l + (size_t)m * jj) * W] : 0.0;
            if (CPLX) bim = in ? Bs[((size_t)l + (size_t)m * jj) * W + 1] : 0.0;
          }
          if (CPLX) {
            acc[j].x += av[u].x * bre - av[u].y * bim;
            acc[j].y += av[u].x * bim + av[u].y * bre;
          } else {
            acc[j].x += av[u].x * bre;
            acc[j].y += av[u].y * bre;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      if (j0 + j >= kcols) break;
      double2* y2 = reinterpret_cast<double2*>(Y.col(j0 + j));
      double2 r = acc[j];
      if (br != 0.0 || (CPLX && bi != 0.0)) {
        const double2 v = y2[i];
        if (CPLX) { r.x += br * v.x - bi * v.y; r.y += br * v.y + bi * v.x; }
        else { r.x += br * v.x; r.y += br * v.y; }
      }
      y2[i] = r;
    }
  }
  if (!CPLX && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // odd tail of real columns
    const long long i = n - 1;
    for (int j = 0; j < KT; ++j) {
      if (j0 + j >= kcols) break;
      double acc1 = 0.0;
      for (int l = 0; l < m; ++l) acc1 += A.col(l)[i] * Bs[(size_t)l + (size_t)m * (j0 + j)];
      double* y = Y.col(j0 + j);
      y[i] = br != 0.0 ? acc1 + br * y[i] : acc1;
    }
  }
}

// ---------------------------------------------------------------- random (K9)
__device__ inline unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void k_random(MvDesc Y, unsigned long long seed, long long ndoubles) {
  const int j = blockIdx.y;
  double* y = Y.col(j);
  const int colid = Y.cols ? Y.cols[j] : Y.col0 + j;
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < ndoubles; i += stride) {
    unsigned long long h = splitmix64(seed ^ splitmix64((unsigned long long)colid * 0x100000001B3ull + 1));
    h = splitmix64(h + (unsigned long long)i);
    // 53 random bits -> (0,1) -> (-1,1), never exactly +-1
    double u = ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    y[i] = 2.0 * u - 1.0;
  }
}

bool same_shape(const mxg_mv* a, const mxg_mv* b) {
  return a->nlocal == b->nlocal && a->k == b->k && a->is_complex == b->is_complex;
}

}  // namespace

// ====================================================================== C ABI
extern "C" int mxg_mv_create(mxg_ctx* c, int64_t nlocal, int k, int is_complex, mxg_mv** out) {
  MXG_CHECK(c && out, "mxg_mv_create: NULL argument");
  MXG_CHECK(nlocal >= 0 && k >= 1, "mxg_mv_create: bad shape %lld x %d", (long long)nlocal, k);
  MXG_HIP(hipSetDevice(c->device));
  mxg_mv* v = new mxg_mv();
  v->ctx = c;
  v->nlocal = nlocal;
  v->is_complex = is_complex ? 1 : 0;
  int64_t doubles = nlocal * (is_complex ? 2 : 1);
  v->ld = (doubles + 31) & ~(int64_t)31;  // 256-byte aligned columns
  if (v->ld == 0) v->ld = 32;
  v->ktotal = v->k = k;
  v->owner = true;
  v->contiguous = true;
  v->col0 = 0;
  v->hcols.resize(k);
  for (int j = 0; j < k; ++j) v->hcols[j] = j;
  if (mxg_vec_alloc(c, sizeof(double) * v->ld * k, (void**)&v->base) != 0) { delete v; return 1; }
  MXG_HIP(hipMemsetAsync(v->base, 0, sizeof(double) * v->ld * k, c->stream));
  *out = v;
  return 0;
}

extern "C" int mxg_mv_view(mxg_mv* p, const int* cols, int n, mxg_mv** out) {
  MXG_CHECK(p && cols && out && n >= 1, "mxg_mv_view: bad argument");
  MXG_HIP(hipSetDevice(p->ctx->device));
  mxg_mv* v = new mxg_mv();
  v->ctx = p->ctx;
  v->base = p->base;
  v->nlocal = p->nlocal;
  v->ld = p->ld;
  v->ktotal = p->ktotal;
  v->k = n;
  v->is_complex = p->is_complex;
  v->owner = false;
  v->hcols.resize(n);
  bool contig = true;
  for (int j = 0; j < n; ++j) {
    if (cols[j] < 0 || cols[j] >= p->k) {
      delete v;
      mxg_set_error("mxg_mv_view: column %d out of range [0,%d)", cols[j], p->k);
      return 1;
    }
    v->hcols[j] = p->hcols[cols[j]];
    if (j > 0 && v->hcols[j] != v->hcols[j - 1] + 1) contig = false;
  }
  v->contiguous = contig;
  v->col0 = v->hcols[0];
  if (!contig) {
    MXG_HIP(hipMalloc(&v->dcols, sizeof(int) * n));
    MXG_HIP(hipMemcpyAsync(v->dcols, v->hcols.data(), sizeof(int) * n, hipMemcpyHostToDevice,
                           v->ctx->stream));
    MXG_HIP(hipStreamSynchronize(v->ctx->stream));
  }
  *out = v;
  return 0;
}

extern "C" int mxg_mv_clone(const mxg_mv* s, const int* cols, int n, mxg_mv** out) {
  MXG_CHECK(s && out, "mxg_mv_clone: NULL argument");
  int k = cols ? n : s->k;
  MXG_CHECK(k >= 1, "mxg_mv_clone: no columns");
  mxg_mv* v = nullptr;
  MXG_TRY(mxg_mv_create(s->ctx, s->nlocal, k, s->is_complex, &v));
  for (int j = 0; j < k; ++j) {
    int sc = cols ? cols[j] : j;
    if (sc < 0 || sc >= s->k) {
      mxg_mv_destroy(v);
      mxg_set_error("mxg_mv_clone: column %d out of range", sc);
      return 1;
    }
    MXG_HIP(hipMemcpyAsync(v->colptr(j), s->colptr(sc), sizeof(double) * s->ld,
                           hipMemcpyDeviceToDevice, s->ctx->stream));
  }
  *out = v;
  return 0;
}

extern "C" int mxg_mv_destroy(mxg_mv* v) {
  if (!v) return 0;
  if (!mxg_ctx_alive(v->ctx)) { delete v; return 0; }   // context already gone: nothing to release safely
  hipSetDevice(v->ctx->device);
  if (v->owner && v->base) mxg_vec_release(v->ctx, v->base, sizeof(double) * v->ld * v->ktotal);
  if (v->dcols) { hipStreamSynchronize(v->ctx->stream); hipFree(v->dcols); }
  delete v;
  return 0;
}

extern "C" int mxg_mv_info(const mxg_mv* v, int64_t* nlocal, int* k, int* is_complex) {
  if (nlocal) *nlocal = v->nlocal;
  if (k) *k = v->k;
  if (is_complex) *is_complex = v->is_complex;
  return 0;
}

extern "C" int mxg_mv_put(mxg_mv* v, int col, int64_t row0, int64_t nrows, const double* host) {
  MXG_CHECK(col >= 0 && col < v->k && row0 >= 0 && row0 + nrows <= v->nlocal,
            "mxg_mv_put: out of range");
  const int s = v->is_complex ? 2 : 1;
  MXG_HIP(hipMemcpyAsync(v->colptr(col) + row0 * s, host, sizeof(double) * nrows * s,
                         hipMemcpyHostToDevice, v->ctx->stream));
  MXG_HIP(hipStreamSynchronize(v->ctx->stream));
  return 0;
}

extern "C" int mxg_mv_fetch(const mxg_mv* v, int col, int64_t row0, int64_t nrows, double* host) {
  MXG_CHECK(col >= 0 && col < v->k && row0 >= 0 && row0 + nrows <= v->nlocal,
            "mxg_mv_fetch: out of range");
  const int s = v->is_complex ? 2 : 1;
  MXG_HIP(hipMemcpyAsync(host, v->colptr(col) + row0 * s, sizeof(double) * nrows * s,
                         hipMemcpyDeviceToHost, v->ctx->stream));
  MXG_HIP(hipStreamSynchronize(v->ctx->stream));
  return 0;
}

extern "C" int mxg_mv_copy(mxg_mv* d, const mxg_mv* s) {
  MXG_CHECK(same_shape(d, s), "mxg_mv_copy: shape mismatch");
  EwArgs g{}; g.y = d->desc(); g.a = s->desc();
  return launch_ew<EW_COPY>(d->ctx, g);
}

extern "C" int mxg_mv_fill(mxg_mv* v, double re, double im) {
  EwArgs g{}; g.y = v->desc(); g.ar = re; g.ai = im;
  return launch_ew<EW_FILL>(v->ctx, g);
}

extern "C" int mxg_mv_random(mxg_mv* v, uint64_t seed) {
  const long long nd = v->nlocal * (v->is_complex ? 2 : 1);
  if (nd == 0) return 0;
  dim3 grid(mxg_grid_for(nd, BLK, 4, 8 * v->ctx->num_cu), v->k);
  // decorrelate ranks: each rank owns different rows of the global vector
  unsigned long long s = seed + 0x51ED27ull * (unsigned long long)v->ctx->rank;
  hipLaunchKernelGGL(k_random, grid, dim3(BLK), 0, v->ctx->stream, v->desc(), s, nd);
  MXG_HIP(hipGetLastError());
  return 0;
}

extern "C" int mxg_mv_conj(mxg_mv* v) {
  if (!v->is_complex) return 0;
  EwArgs g{}; g.y = v->desc();
  return launch_ew<EW_CONJ>(v->ctx, g);
}

extern "C" int mxg_mv_scale(mxg_mv* v, double re, double im) {
  EwArgs g{}; g.y = v->desc(); g.ar = re; g.ai = im;
  return launch_ew<EW_SCALE>(v->ctx, g);
}

extern "C" int mxg_mv_scale_cols(mxg_mv* v, const double* alphas) {
  mxg_ctx* c = v->ctx;
  const size_t w = (size_t)v->k * (v->is_complex ? 2 : 1);
  MXG_TRY(mxg_ctx_reserve_small(c, w));
  MXG_HIP(hipMemcpyAsync(c->d_small, alphas, sizeof(double) * w, hipMemcpyHostToDevice, c->stream));
  EwArgs g{}; g.y = v->desc(); g.alphas = c->d_small;
  MXG_TRY(launch_ew<EW_SCALE_COLS>(c, g));
  MXG_HIP(hipStreamSynchronize(c->stream));  // d_small / caller's buffer are reusable on return
  return 0;
}

extern "C" int mxg_mv_mask_rows(mxg_mv* v, const mxg_mv* mask) {
  MXG_CHECK(mask->nlocal == v->nlocal && !mask->is_complex, "mxg_mv_mask_rows: mask shape");
  EwArgs g{}; g.y = v->desc(); g.a = mask->desc();
  return launch_ew<EW_MASK>(v->ctx, g);
}

extern "C" int mxg_mv_axpby(double ar, double ai, const mxg_mv* A, double br, double bi,
                            mxg_mv* self) {
  MXG_CHECK(same_shape(A, self), "mxg_mv_axpby: shape mismatch");
  EwArgs g{}; g.y = self->desc(); g.a = A->desc(); g.ar = ar; g.ai = ai; g.br = br; g.bi = bi;
  return launch_ew<EW_AXPBY>(self->ctx, g);
}

extern "C" int mxg_mv_add(double ar, double ai, const mxg_mv* A, double br, double bi,
                          const mxg_mv* B, mxg_mv* self) {
  MXG_CHECK(same_shape(A, self) && same_shape(B, self), "mxg_mv_add: shape mismatch");
  EwArgs g{}; g.y = self->desc(); g.a = A->desc(); g.b = B->desc();
  g.ar = ar; g.ai = ai; g.br = br; g.bi = bi;
  return launch_ew<EW_ADD>(self->ctx, g);
}

int mxg_dev_dot(mxg_ctx* c, const MvDesc& A, const MvDesc& B, double* d_out) {
  const int width = A.cplx ? 2 : 1;
  MXG_TRY(mxg_ctx_reserve_reduce(c, (size_t)A.k * width));
  const long long npair = A.cplx ? A.n : (A.n + 1) / 2;
  const int gx = mxg_grid_for(npair, BLK, 8, 1024);
  dim3 grid(gx, A.k);
  const bool self = A.base == B.base && A.ld == B.ld && A.col0 == B.col0 && A.cols == B.cols;
  if (A.cplx) {
    if (self) hipLaunchKernelGGL((k_dot<true, true>), grid, dim3(BLK), 0, c->stream, A, B, c->d_partial);
    else hipLaunchKernelGGL((k_dot<true, false>), grid, dim3(BLK), 0, c->stream, A, B, c->d_partial);
  } else {
    if (self) hipLaunchKernelGGL((k_dot<false, true>), grid, dim3(BLK), 0, c->stream, A, B, c->d_partial);
    else hipLaunchKernelGGL((k_dot<false, false>), grid, dim3(BLK), 0, c->stream, A, B, c->d_partial);
  }
  hipLaunchKernelGGL(k_reduce_final, dim3(A.k * width), dim3(BLK), 0, c->stream, c->d_partial, gx, d_out);
  MXG_HIP(hipGetLastError());
  return 0;
}

// dot / norm to the host: without a communicator the second stage stores straight into the pinned result buffer (no
// copy call: two launches + one stream wait per call); with one, the device buffer is all-reduced first
static int dot_to_host(mxg_ctx* c, const MvDesc& A, const MvDesc& B, size_t w, double* host_out) {
  MXG_TRY(mxg_ctx_reserve_reduce(c, w));
  static const bool hostout = !(getenv("MXG_DOT_HOSTOUT") && atoi(getenv("MXG_DOT_HOSTOUT")) == 0);
  if (mxg_comm_on(c) || !c->h_result_dev || !hostout) {
    MXG_TRY(mxg_dev_dot(c, A, B, c->d_result));
    return mxg_ctx_finish_reduce(c, w, host_out);
  }
  MXG_TRY(mxg_dev_dot(c, A, B, c->h_result_dev));
  MXG_HIP(hipStreamSynchronize(c->stream));
  memcpy(host_out, c->h_result, sizeof(double) * w);
  return 0;
}

extern "C" int mxg_mv_dot(const mxg_mv* A, const mxg_mv* B, double* res) {
  MXG_CHECK(same_shape(A, B), "mxg_mv_dot: shape mismatch");
  const size_t w = (size_t)A->k * (A->is_complex ? 2 : 1);
  return dot_to_host(A->ctx, A->desc(), B->desc(), w, res);
}

extern "C" int mxg_mv_norm2(const mxg_mv* v, double* res) {
  const int width = v->is_complex ? 2 : 1;
  const size_t w = (size_t)v->k * width;
  std::vector<double> tmp(w);
  MXG_TRY(dot_to_host(v->ctx, v->desc(), v->desc(), w, tmp.data()));
  for (int j = 0; j < v->k; ++j) res[j] = sqrt(tmp[(size_t)j * width]);
  return 0;
}

extern "C" int mxg_mv_set_block(mxg_mv* self, const mxg_mv* A, const int* index, int n) {
  MXG_CHECK(A->nlocal == self->nlocal && A->is_complex == self->is_complex && n <= A->k,
            "mxg_mv_set_block: shape mismatch");
  mxg_ctx* c = self->ctx;
  // translate destination view columns to storage columns and stage them on the device
  std::vector<int> dst(n);
  for (int j = 0; j < n; ++j) {
    MXG_CHECK(index[j] >= 0 && index[j] < self->k, "mxg_mv_set_block: index %d out of range",
              index[j]);
    dst[j] = self->hcols[index[j]];
  }
  MXG_TRY(mxg_ctx_reserve_small(c, (size_t)n));
  int* dcols = reinterpret_cast<int*>(c->d_small);
  MXG_HIP(hipMemcpyAsync(dcols, dst.data(), sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
  EwArgs g{}; g.y = self->desc(); g.y.k = n; g.a = A->desc(); g.ycols = dcols;
  MXG_TRY(launch_ew<EW_COPY>(c, g));
  MXG_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

namespace {
template <int KT, bool CPLX>
void launch_times_mat(mxg_ctx* c, int grid, const MvDesc& A, const double* Bs, int m, int j0,
                      int kcols, double br, double bi, const MvDesc& Y) {
  const int mpad = (m + TM_UL - 1) / TM_UL * TM_UL;
  if ((size_t)mpad * KT * (CPLX ? 2 : 1) <= (size_t)TM_LDS_DOUBLES)
    hipLaunchKernelGGL((k_times_mat<KT, CPLX, true>), dim3(grid), dim3(BLK), 0, c->stream, A, Bs, m, j0, kcols, br, bi, Y);
  else
    hipLaunchKernelGGL((k_times_mat<KT, CPLX, false>), dim3(grid), dim3(BLK), 0, c->stream, A, Bs, m, j0, kcols, br, bi, Y);
}
template <bool CPLX>
void dispatch_times_mat(mxg_ctx* c, int grid, const MvDesc& A, const double* Bs, int m, int kcols,
                        double br, double bi, const MvDesc& Y) {
  int j0 = 0;
  while (j0 < kcols) {
    int rem = kcols - j0;
    if (rem >= 8 && !CPLX) { launch_times_mat<8, CPLX>(c, grid, A, Bs, m, j0, kcols, br, bi, Y); j0 += 8; }
    else if (rem >= 4) { launch_times_mat<4, CPLX>(c, grid, A, Bs, m, j0, kcols, br, bi, Y); j0 += 4; }
    else if (rem >= 2) { launch_times_mat<2, CPLX>(c, grid, A, Bs, m, j0, kcols, br, bi, Y); j0 += 2; }
    else { launch_times_mat<1, CPLX>(c, grid, A, Bs, m, j0, kcols, br, bi, Y); j0 += 1; }
  }
}
}  // namespace

extern "C" int mxg_mv_times_mat(double alr, double ali, const mxg_mv* A, const double* B, int ldb,
                                int m, int k, double ber, double bei, mxg_mv* self) {
  MXG_CHECK(A->nlocal == self->nlocal && A->is_complex == self->is_complex,
            "mxg_mv_times_mat: shape mismatch");
  MXG_CHECK(m == A->k && k == self->k && ldb >= m, "mxg_mv_times_mat: B is %d x %d (ld %d) but A has %d, this %d columns",
            m, k, ldb, A->k, self->k);
  mxg_ctx* c = self->ctx;
  const bool cplx = self->is_complex;
  const int w = cplx ? 2 : 1;
  // pack alpha * B densely (m x k) on the host, then stage it on the device
  std::vector<double> Bs((size_t)m * k * w);
  for (int j = 0; j < k; ++j)
    for (int l = 0; l < m; ++l) {
      if (cplx) {
        double bre = B[((size_t)l + (size_t)ldb * j) * 2], bim = B[((size_t)l + (size_t)ldb * j) * 2 + 1];
        Bs[((size_t)l + (size_t)m * j) * 2] = alr * bre - ali * bim;
        Bs[((size_t)l + (size_t)m * j) * 2 + 1] = alr * bim + ali * bre;
      } else {
        Bs[(size_t)l + (size_t)m * j] = alr * B[(size_t)l + (size_t)ldb * j];
      }
    }
  MXG_TRY(mxg_ctx_reserve_small(c, Bs.size()));
  MXG_HIP(hipMemcpyAsync(c->d_small, Bs.data(), sizeof(double) * Bs.size(), hipMemcpyHostToDevice,
                         c->stream));
  const int grid = mxg_grid_for(self->is_complex ? self->nlocal : (self->nlocal + 1) / 2, BLK, 1, 8 * c->num_cu);
  if (self->nlocal > 0) {
    if (cplx) dispatch_times_mat<true>(c, grid, A->desc(), c->d_small, m, k, ber, bei, self->desc());
    else dispatch_times_mat<false>(c, grid, A->desc(), c->d_small, m, k, ber, bei, self->desc());
  }
  MXG_HIP(hipGetLastError());
  MXG_HIP(hipStreamSynchronize(c->stream));  // Bs (host) and d_small are reusable on return
  return 0;
}

namespace {
template <int LT, int KT, bool CPLX>
void launch_trans_mv(mxg_ctx* c, int gx, const MvDesc& A, const MvDesc& A2, const MvDesc& T, int m, int k) {
  for (int j0 = 0; j0 < k; j0 += KT)
    hipLaunchKernelGGL((k_trans_mv<LT, KT, CPLX>), dim3(gx), dim3(BLK), 0, c->stream, A, A2, T, j0, m, c->d_partial);
}
}  // namespace

// B = alpha [A A2]^H self, (A.k + A2.k) x self.k; A2 may be NULL
extern "C" int mxg_mv_trans_mv2(double alr, double ali, const mxg_mv* A, const mxg_mv* A2, const mxg_mv* self,
                                double* B, int ldb) {
  MXG_CHECK(A->nlocal == self->nlocal && A->is_complex == self->is_complex,
            "mxg_mv_trans_mv: shape mismatch");
  MXG_CHECK(!A2 || (A2->nlocal == self->nlocal && A2->is_complex == self->is_complex && A2->ctx == self->ctx),
            "mxg_mv_trans_mv2: shape mismatch of the appended columns");
  mxg_ctx* c = self->ctx;
  const int m = A->k + (A2 ? A2->k : 0), k = self->k;
  MXG_CHECK(ldb >= m, "mxg_mv_trans_mv: ldb %d < m %d", ldb, m);
  const bool cplx = self->is_complex;
  const int w = cplx ? 2 : 1;
  const size_t width = (size_t)m * k * w;
  MXG_TRY(mxg_ctx_reserve_reduce(c, width));
  // each block walks 64 elements per step (the four wavefronts share the rows)
  const int gx = mxg_grid_for(self->is_complex ? self->nlocal : (self->nlocal + 1) / 2, 64, 8, 1024);
  MvDesc Ad = A->desc(), A2d = A2 ? A2->desc() : A->desc(), Td = self->desc();
  // single vectors (the Krylov-Schur step with block size 1): the tile width that spreads the m columns evenly over the
  // four wavefronts and the fewest passes, e.g. m = 100 -> 4 passes of 7 columns (15 tiles), not 13 tiles of 8
  const int passes = (m + 31) / 32, lt1 = std::max(1, (m + 4 * passes - 1) / (4 * passes));
  if (cplx) {
    if (k >= 4) launch_trans_mv<4, 4, true>(c, gx, Ad, A2d, Td, m, k);
    else if (k >= 2) launch_trans_mv<8, 2, true>(c, gx, Ad, A2d, Td, m, k);
    else switch (lt1) {
      case 1: launch_trans_mv<1, 1, true>(c, gx, Ad, A2d, Td, m, k); break;
      case 2: launch_trans_mv<2, 1, true>(c, gx, Ad, A2d, Td, m, k); break;
      case 3: launch_trans_mv<3, 1, true>(c, gx, Ad, A2d, Td, m, k); break;
      case 4: launch_trans_mv<4, 1, true>(c, gx, Ad, A2d, Td, m, k); break;
      case 5: launch_trans_mv<5, 1, true>(c, gx, Ad, A2d, Td, m, k); break;
      case 6: launch_trans_mv<6, 1, true>(c, gx, Ad, A2d, Td, m, k); break;
      case 7: launch_trans_mv<7, 1, true>(c, gx, Ad, A2d, Td, m, k); break;
      default: launch_trans_mv<8, 1, true>(c, gx, Ad, A2d, Td, m, k); break;
    }
  } else {
    if (k >= 8) launch_trans_mv<4, 8, false>(c, gx, Ad, A2d, Td, m, k);
    else if (k >= 4) launch_trans_mv<8, 4, false>(c, gx, Ad, A2d, Td, m, k);
    else if (k >= 2) launch_trans_mv<8, 2, false>(c, gx, Ad, A2d, Td, m, k);
    else switch (lt1) {
      case 1: launch_trans_mv<1, 1, false>(c, gx, Ad, A2d, Td, m, k); break;
      case 2: launch_trans_mv<2, 1, false>(c, gx, Ad, A2d, Td, m, k); break;
      case 3: launch_trans_mv<3, 1, false>(c, gx, Ad, A2d, Td, m, k); break;
      case 4: launch_trans_mv<4, 1, false>(c, gx, Ad, A2d, Td, m, k); break;
      case 5: launch_trans_mv<5, 1, false>(c, gx, Ad, A2d, Td, m, k); break;
      case 6: launch_trans_mv<6, 1, false>(c, gx, Ad, A2d, Td, m, k); break;
      case 7: launch_trans_mv<7, 1, false>(c, gx, Ad, A2d, Td, m, k); break;
      default: launch_trans_mv<8, 1, false>(c, gx, Ad, A2d, Td, m, k); break;
    }
  }
  hipLaunchKernelGGL(k_reduce_final, dim3((unsigned)width), dim3(BLK), 0, c->stream, c->d_partial, gx,
                     c->d_result);
  MXG_HIP(hipGetLastError());
  std::vector<double> tmp(width);
  MXG_TRY(mxg_ctx_finish_reduce(c, width, tmp.data()));
  for (int j = 0; j < k; ++j)
    for (int l = 0; l < m; ++l) {
      const size_t s = ((size_t)l + (size_t)m * j) * w;
      if (cplx) {
        B[((size_t)l + (size_t)ldb * j) * 2] = alr * tmp[s] - ali * tmp[s + 1];
        B[((size_t)l + (size_t)ldb * j) * 2 + 1] = alr * tmp[s + 1] + ali * tmp[s];
      } else {
        B[(size_t)l + (size_t)ldb * j] = alr * tmp[s];
      }
    }
  return 0;
}

extern "C" int mxg_mv_trans_mv(double alr, double ali, const mxg_mv* A, const mxg_mv* self,
                               double* B, int ldb) {
  return mxg_mv_trans_mv2(alr, ali, A, nullptr, self, B, ldb);
}
