// CPU backend of the device C ABI (include/mxg.h) - TEST / BASELINE INFRASTRUCTURE, not a product path.
//
// SURVEY section 8(d) asks for the reference's CPU path timed beside the GPU: the reference's CPU path is Epetra's
// serial CSR row loop under AztecOO / ML / Anasazi, one MPI rank per core.  Trilinos is not available, so the
// baseline is THIS repository's own solver stack - the C++ drop-in classes of include/mx/ (MxSolver, the block
// Krylov-Schur manager, MxMagWaveOp::Apply), the Krylov loops of maxwell_amd/csrc/device/mxg_krylov.h and the
// multigrid hierarchy builder - running unchanged over a plain host implementation of the same ABI: CSR row loops,
// vector loops, the same V-cycle (Chebyshev smoothing, over-correction 1.9), OpenMP threads standing in for the
// ranks.  bench.py's cpu_baseline leg and tests/test_cpu_backend.py load it from oracle/_cpu/ (libmxg.so +
// a libmxhost.so linked against it); nothing under maxwell_amd/ knows it exists, and the product still fails loudly
// without the HIP library (tests/test_abi.py).
// Single rank only; no ghosts; the setup-time sparse algebra is include/mx/mx_sparse.hpp (the host chain).
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mxg.h"
#include "mx/mx_sparse.hpp"

static thread_local std::string g_err;
static void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
#define CHECK(cond, ...) do { if (!(cond)) { set_error(__VA_ARGS__); return 1; } } while (0)
#define TRY(call) do { int rc_ = (call); if (rc_ != 0) return rc_; } while (0)
#define UNSUPPORTED(name) do { set_error("%s: not available in the CPU backend (single rank, no device)", name); return 1; } while (0)

typedef std::complex<double> cplx;

struct mxg_ctx {
  std::chrono::steady_clock::time_point t0;
};
struct mxg_mv {
  mxg_ctx* ctx = nullptr;
  double* base = nullptr;
  int64_t nlocal = 0, ld = 0;
  int k = 0, is_complex = 0;
  bool owner = false;
  std::vector<int> cols;
  double* col(int j) const { return base + (int64_t)cols[j] * ld; }
  int w() const { return is_complex ? 2 : 1; }
};
struct mxg_csr {
  mxg_ctx* ctx = nullptr;
  int64_t nrows = 0, ncols = 0;
  int is_complex = 0;
  std::vector<int64_t> rp;
  std::vector<int32_t> ci;
  std::vector<double> va, diag;
  int64_t nnz() const { return rp.empty() ? 0 : rp.back(); }
};

extern "C" {

const char* mxg_last_error(void) { return g_err.c_str(); }
int mxg_get_unique_id(void* out) { memset(out, 0, MXG_UNIQUE_ID_BYTES); return 0; }
int mxg_ctx_create(int, int rank, int nranks, const void*, mxg_ctx** out) {
  CHECK(out && rank == 0 && nranks == 1, "mxg_ctx_create: the CPU backend is single-rank");
  *out = new mxg_ctx();
  return 0;
}
int mxg_ctx_destroy(mxg_ctx* c) { delete c; return 0; }
int mxg_ctx_sync(mxg_ctx*) { return 0; }
int mxg_ctx_rank(const mxg_ctx*) { return 0; }
int mxg_ctx_nranks(const mxg_ctx*) { return 1; }
int mxg_ctx_comm_ranks(const mxg_ctx*) { return 1; }
int mxg_timer_start(mxg_ctx* c) { c->t0 = std::chrono::steady_clock::now(); return 0; }
int mxg_timer_stop(mxg_ctx* c, float* ms) {
  *ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c->t0).count();
  return 0;
}
int mxg_barrier(mxg_ctx*) { return 0; }
int mxg_bench_stream(mxg_ctx*, int64_t, int64_t, int, float*) { UNSUPPORTED("mxg_bench_stream"); }
int mxg_local_world_create(int, mxg_local_world**) { UNSUPPORTED("mxg_local_world_create"); }
int mxg_local_world_destroy(mxg_local_world*) { return 0; }
int mxg_ctx_create_local(int, int, int, mxg_local_world*, mxg_ctx**) { UNSUPPORTED("mxg_ctx_create_local"); }
int mxg_comm_allgather_i64(mxg_ctx*, int64_t v, int64_t* out) { out[0] = v; return 0; }
int mxg_comm_allgatherv(mxg_ctx*, const void* send, int64_t nbytes, const int64_t*, void* recv) {
  if (nbytes > 0 && recv != send) memcpy(recv, send, (size_t)nbytes);
  return 0;
}
int mxg_comm_allreduce_host(mxg_ctx*, double*, int) { return 0; }

// ---------------------------------------------------------------------------------------------------- CSR
int mxg_csr_create(mxg_ctx* c, int64_t nrows, int64_t ncols_owned, int64_t nghost, const int64_t* rowptr,
                   const int32_t* colidx, const double* vals, int is_complex, const mxg_halo_plan*, mxg_csr** out) {
  CHECK(c && rowptr && out, "mxg_csr_create: NULL argument");
  CHECK(nghost == 0, "mxg_csr_create: the CPU backend has no ghost import");
  CHECK(rowptr[0] == 0, "mxg_csr_create: rowptr[0] must be 0");
  const int64_t nnz = rowptr[nrows];
  for (int64_t q = 0; q < nnz; ++q)
    CHECK(colidx[q] >= 0 && colidx[q] < ncols_owned, "mxg_csr_create: column index %d out of range [0,%lld) at entry %lld",
          colidx[q], (long long)ncols_owned, (long long)q);
  mxg_csr* a = new mxg_csr();
  const int w = is_complex ? 2 : 1;
  a->ctx = c; a->nrows = nrows; a->ncols = ncols_owned; a->is_complex = is_complex ? 1 : 0;
  a->rp.assign(rowptr, rowptr + nrows + 1);
  a->ci.assign(colidx, colidx + nnz);
  a->va.assign(vals, vals + nnz * w);
  a->diag.assign((size_t)std::max<int64_t>(nrows, 1) * w, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrows; ++r)
    for (int64_t q = rowptr[r]; q < rowptr[r + 1]; ++q)
      if (colidx[q] == r) { a->diag[r * w] += vals[q * w]; if (w == 2) a->diag[r * w + 1] += vals[q * w + 1]; }
  *out = a;
  return 0;
}
int mxg_csr_destroy(mxg_csr* a) { delete a; return 0; }
int mxg_csr_info(const mxg_csr* a, int64_t* nrows, int64_t* ncols_owned, int64_t* nghost, int64_t* nnz, int* is_complex) {
  if (nrows) *nrows = a->nrows;
  if (ncols_owned) *ncols_owned = a->ncols;
  if (nghost) *nghost = 0;
  if (nnz) *nnz = a->nnz();
  if (is_complex) *is_complex = a->is_complex;
  return 0;
}
int64_t mxg_csr_apply_bytes(const mxg_csr* a, int k) {
  const int64_t sb = a->is_complex ? 16 : 8;
  return a->nnz() * (sb + 4) + a->nrows * (8 + sb * k) + a->ncols * sb * k;
}
int mxg_csr_set_kernel(mxg_csr*, int) { return 0; }
const char* mxg_csr_kernel_name(const mxg_csr*) { return "serial CSR row loop, OpenMP static rows (CPU backend)"; }
int mxg_csr_compress(mxg_csr*, int64_t* nb, int64_t* nf, int64_t* sb) {
  if (nb) *nb = 0;
  if (nf) *nf = 0;
  if (sb) *sb = 0;
  return 0;
}
int mxg_csr_compress_from(mxg_csr* a, const int64_t*, const int32_t*, const double*, int64_t* nb, int64_t* nf, int64_t* sb) {
  return mxg_csr_compress(a, nb, nf, sb);
}
int mxg_csr_product_form(const mxg_csr*, int64_t* t, int64_t* p, int64_t* g, double* cov) {
  if (t) *t = 0;
  if (p) *p = 0;
  if (g) *g = 0;
  if (cov) *cov = 0.0;
  return 0;
}
int mxg_debug_tile_plan_apply(int64_t, int64_t, const int64_t*, const int32_t*, const double*, int, const double*, const double*,
                              double*, int64_t*) { UNSUPPORTED("mxg_debug_tile_plan_apply"); }
int mxg_csr_download(const mxg_csr* a, int64_t* rowptr, int32_t* colidx, double* vals) {
  if (rowptr) memcpy(rowptr, a->rp.data(), sizeof(int64_t) * a->rp.size());
  if (colidx && !a->ci.empty()) memcpy(colidx, a->ci.data(), sizeof(int32_t) * a->ci.size());
  if (vals && !a->va.empty()) memcpy(vals, a->va.data(), sizeof(double) * a->va.size());
  return 0;
}

}  // extern "C"

// the entries of one row in storage order, every product rounded before it is added (g++ -ffp-contract=off):
// the arithmetic of oracle/spmv_ref.c
static void csr_apply_col(const mxg_csr* a, const double* x, double* y) {
  const int64_t n = a->nrows;
  const int64_t* rp = a->rp.data();
  const int32_t* ci = a->ci.data();
  const double* va = a->va.data();
  if (!a->is_complex) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
      double s = 0.0;
      for (int64_t q = rp[r]; q < rp[r + 1]; ++q) s += va[q] * x[ci[q]];
      y[r] = s;
    }
  } else {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
      double sr = 0.0, si = 0.0;
      for (int64_t q = rp[r]; q < rp[r + 1]; ++q) {
        const double ar = va[2 * q], ai = va[2 * q + 1], xr = x[2 * (int64_t)ci[q]], xi = x[2 * (int64_t)ci[q] + 1];
        sr += ar * xr - ai * xi;
        si += ar * xi + ai * xr;
      }
      y[2 * r] = sr; y[2 * r + 1] = si;
    }
  }
}

template <typename S>
static mx::Csr<S> to_host(const mxg_csr* a) {
  mx::Csr<S> m;
  m.nrows = a->nrows; m.ncols = a->ncols;
  m.rowptr.resize(a->rp.size());
  std::copy(a->rp.begin(), a->rp.end(), m.rowptr.begin());
  m.col.resize(a->ci.size());
  std::copy(a->ci.begin(), a->ci.end(), m.col.begin());
  m.val.resize(a->ci.size());
  memcpy(reinterpret_cast<double*>(m.val.data()), a->va.data(), sizeof(double) * a->va.size());
  return m;
}
static int from_host(mxg_ctx* c, const mx::Csr<double>& m, mxg_csr** out) {
  return mxg_csr_create(c, m.nrows, m.ncols, 0, m.rowptr.data(), m.col.data(), m.val.data(), 0, nullptr, out);
}
static int from_host(mxg_ctx* c, const mx::Csr<std::complex<double> >& m, mxg_csr** out) {
  return mxg_csr_create(c, m.nrows, m.ncols, 0, m.rowptr.data(), m.col.data(), reinterpret_cast<const double*>(m.val.data()), 1, nullptr, out);
}

extern "C" {

// the host versions of include/mx/mx_sparse.hpp ARE the definition the device kernels reproduce
int mxg_csr_multiply(mxg_ctx* c, const mxg_csr* a, const mxg_csr* b, mxg_csr** out) {
  CHECK(a->is_complex == b->is_complex, "mxg_csr_multiply: one operand is real, the other complex");
  if (a->is_complex) return from_host(c, mx::multiply(to_host<std::complex<double> >(a), to_host<std::complex<double> >(b)), out);
  return from_host(c, mx::multiply(to_host<double>(a), to_host<double>(b)), out);
}
int mxg_csr_add_z(mxg_ctx* c, const mxg_csr* a, const double sa[2], const mxg_csr* b, const double sb[2], mxg_csr** out) {
  CHECK(a->is_complex == b->is_complex, "mxg_csr_add: one operand is real, the other complex");
  if (a->is_complex)
    return from_host(c, mx::add(to_host<std::complex<double> >(a), std::complex<double>(sa[0], sa[1]), to_host<std::complex<double> >(b),
                                std::complex<double>(sb[0], sb[1])), out);
  CHECK(sa[1] == 0.0 && sb[1] == 0.0, "mxg_csr_add: complex scalars on real matrices");
  return from_host(c, mx::add(to_host<double>(a), sa[0], to_host<double>(b), sb[0]), out);
}
int mxg_csr_add(mxg_ctx* c, const mxg_csr* a, double sa, const mxg_csr* b, double sb, mxg_csr** out) {
  const double za[2] = {sa, 0.0}, zb[2] = {sb, 0.0};
  return mxg_csr_add_z(c, a, za, b, zb, out);
}
int mxg_csr_purge(mxg_ctx* c, const mxg_csr* a, double, mxg_csr** out) {
  if (a->is_complex) return from_host(c, mx::purge(to_host<std::complex<double> >(a)), out);
  return from_host(c, mx::purge(to_host<double>(a)), out);
}

int mxg_csr_apply(mxg_ctx*, const mxg_csr* a, const mxg_mv* x, mxg_mv* y) {
  CHECK(a && x && y, "mxg_csr_apply: NULL argument");
  CHECK(x->k == y->k, "mxg_csr_apply: x has %d columns, y has %d", x->k, y->k);
  CHECK(x->nlocal == a->ncols, "mxg_csr_apply: x has %lld rows, matrix has %lld owned columns", (long long)x->nlocal, (long long)a->ncols);
  CHECK(y->nlocal == a->nrows, "mxg_csr_apply: y has %lld rows, matrix has %lld rows", (long long)y->nlocal, (long long)a->nrows);
  CHECK(x->is_complex == a->is_complex && y->is_complex == a->is_complex, "mxg_csr_apply: scalar type mismatch");
  std::vector<double> tmp;
  for (int j = 0; j < x->k; ++j) {
    const double* xp = x->col(j);
    bool alias = false;                         // apply(x, x): work from a copy, like Epetra
    for (int i = 0; i < y->k && !alias; ++i) alias = x->base == y->base && y->cols[i] == x->cols[j];
    if (alias) { tmp.assign(xp, xp + x->nlocal * x->w()); xp = tmp.data(); }
    // (all columns of x are read before a later column of y could overwrite them only when views do not overlap
    //  out of order; the solver never does that - same contract as the device library's scratch copy)
    csr_apply_col(a, xp, y->col(j));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------- multivector
int mxg_mv_create(mxg_ctx* c, int64_t nlocal, int k, int is_complex, mxg_mv** out) {
  CHECK(c && out && nlocal >= 0 && k >= 1, "mxg_mv_create: bad argument");
  mxg_mv* v = new mxg_mv();
  v->ctx = c; v->nlocal = nlocal; v->is_complex = is_complex ? 1 : 0; v->k = k; v->owner = true;
  const int64_t doubles = nlocal * v->w();
  v->ld = std::max<int64_t>((doubles + 31) & ~(int64_t)31, 32);
  v->base = static_cast<double*>(aligned_alloc(256, sizeof(double) * (size_t)v->ld * k));
  CHECK(v->base, "mxg_mv_create: out of memory");
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < v->ld * k; ++i) v->base[i] = 0.0;     // (first touch by the threads that will use it)
  v->cols.resize(k);
  for (int j = 0; j < k; ++j) v->cols[j] = j;
  *out = v;
  return 0;
}
int mxg_mv_view(mxg_mv* p, const int* cols, int n, mxg_mv** out) {
  CHECK(p && cols && out && n >= 1, "mxg_mv_view: bad argument");
  mxg_mv* v = new mxg_mv(*p);
  v->owner = false; v->k = n;
  v->cols.resize(n);
  for (int j = 0; j < n; ++j) {
    CHECK(cols[j] >= 0 && cols[j] < p->k, "mxg_mv_view: column %d out of range", cols[j]);
    v->cols[j] = p->cols[cols[j]];
  }
  *out = v;
  return 0;
}
int mxg_mv_clone(const mxg_mv* src, const int* cols, int n, mxg_mv** out) {
  const int k = cols ? n : src->k;
  TRY(mxg_mv_create(src->ctx, src->nlocal, k, src->is_complex, out));
  for (int j = 0; j < k; ++j) memcpy((*out)->col(j), src->col(cols ? cols[j] : j), sizeof(double) * src->nlocal * src->w());
  return 0;
}
int mxg_mv_destroy(mxg_mv* v) {
  if (v && v->owner) free(v->base);
  delete v;
  return 0;
}
int mxg_mv_info(const mxg_mv* v, int64_t* nlocal, int* k, int* is_complex) {
  if (nlocal) *nlocal = v->nlocal;
  if (k) *k = v->k;
  if (is_complex) *is_complex = v->is_complex;
  return 0;
}
int mxg_mv_put(mxg_mv* v, int col, int64_t row0, int64_t nrows, const double* host) {
  CHECK(col >= 0 && col < v->k && row0 >= 0 && row0 + nrows <= v->nlocal, "mxg_mv_put: out of range");
  memcpy(v->col(col) + row0 * v->w(), host, sizeof(double) * nrows * v->w());
  return 0;
}
int mxg_mv_fetch(const mxg_mv* v, int col, int64_t row0, int64_t nrows, double* host) {
  CHECK(col >= 0 && col < v->k && row0 >= 0 && row0 + nrows <= v->nlocal, "mxg_mv_fetch: out of range");
  memcpy(host, v->col(col) + row0 * v->w(), sizeof(double) * nrows * v->w());
  return 0;
}

}  // extern "C"

static bool same_shape(const mxg_mv* a, const mxg_mv* b) { return a->nlocal == b->nlocal && a->k == b->k && a->is_complex == b->is_complex; }

// y[i] = f(i) over the doubles (real) or the complex entries of every column
template <class F>
static void each_real(mxg_mv* y, F f) {
  for (int j = 0; j < y->k; ++j) {
    double* p = y->col(j);
    const int64_t n = y->nlocal * y->w();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) f(j, i, p);
  }
}
template <class F>
static void each_cplx(mxg_mv* y, F f) {
  for (int j = 0; j < y->k; ++j) {
    cplx* p = reinterpret_cast<cplx*>(y->col(j));
    const int64_t n = y->nlocal;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) f(j, i, p);
  }
}

extern "C" {

int mxg_mv_copy(mxg_mv* dst, const mxg_mv* src) {
  CHECK(same_shape(dst, src), "mxg_mv_copy: shape mismatch");
  each_real(dst, [&](int j, int64_t i, double* p) { p[i] = src->col(j)[i]; });
  return 0;
}
int mxg_mv_fill(mxg_mv* v, double re, double im) {
  if (v->is_complex) each_cplx(v, [&](int, int64_t i, cplx* p) { p[i] = cplx(re, im); });
  else each_real(v, [&](int, int64_t i, double* p) { p[i] = re; });
  return 0;
}
static inline unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
int mxg_mv_random(mxg_mv* v, uint64_t seed) {     // the counter-based generator of the device library: same start vectors
  each_real(v, [&](int j, int64_t i, double* p) {
    unsigned long long h = splitmix64(seed ^ splitmix64((unsigned long long)v->cols[j] * 0x100000001B3ull + 1));
    h = splitmix64(h + (unsigned long long)i);
    const double u = ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    p[i] = 2.0 * u - 1.0;
  });
  return 0;
}
int mxg_mv_conj(mxg_mv* v) {
  if (v->is_complex) each_cplx(v, [&](int, int64_t i, cplx* p) { p[i] = std::conj(p[i]); });
  return 0;
}
int mxg_mv_scale(mxg_mv* v, double re, double im) {
  if (v->is_complex) each_cplx(v, [&](int, int64_t i, cplx* p) { p[i] *= cplx(re, im); });
  else each_real(v, [&](int, int64_t i, double* p) { p[i] *= re; });
  return 0;
}
int mxg_mv_scale_cols(mxg_mv* v, const double* alphas) {
  if (v->is_complex) each_cplx(v, [&](int j, int64_t i, cplx* p) { p[i] *= cplx(alphas[2 * j], alphas[2 * j + 1]); });
  else each_real(v, [&](int j, int64_t i, double* p) { p[i] *= alphas[j]; });
  return 0;
}
int mxg_mv_mask_rows(mxg_mv* v, const mxg_mv* mask) {
  CHECK(mask->nlocal == v->nlocal && !mask->is_complex, "mxg_mv_mask_rows: mask shape");
  const double* m = mask->col(0);
  if (v->is_complex) each_cplx(v, [&](int, int64_t i, cplx* p) { p[i] *= m[i]; });
  else each_real(v, [&](int, int64_t i, double* p) { p[i] *= m[i]; });
  return 0;
}
int mxg_mv_axpby(double ar, double ai, const mxg_mv* A, double br, double bi, mxg_mv* self) {
  CHECK(same_shape(A, self), "mxg_mv_axpby: shape mismatch");
  if (self->is_complex) {
    const cplx a(ar, ai), b(br, bi);
    each_cplx(self, [&](int j, int64_t i, cplx* p) { p[i] = a * reinterpret_cast<const cplx*>(A->col(j))[i] + b * p[i]; });
  } else {
    each_real(self, [&](int j, int64_t i, double* p) { p[i] = ar * A->col(j)[i] + br * p[i]; });
  }
  return 0;
}
int mxg_mv_add(double ar, double ai, const mxg_mv* A, double br, double bi, const mxg_mv* B, mxg_mv* self) {
  CHECK(same_shape(A, self) && same_shape(B, self), "mxg_mv_add: shape mismatch");
  if (self->is_complex) {
    const cplx a(ar, ai), b(br, bi);
    each_cplx(self, [&](int j, int64_t i, cplx* p) {
      p[i] = a * reinterpret_cast<const cplx*>(A->col(j))[i] + b * reinterpret_cast<const cplx*>(B->col(j))[i];
    });
  } else {
    each_real(self, [&](int j, int64_t i, double* p) { p[i] = ar * A->col(j)[i] + br * B->col(j)[i]; });
  }
  return 0;
}
int mxg_mv_dot(const mxg_mv* A, const mxg_mv* B, double* res) {
  CHECK(same_shape(A, B), "mxg_mv_dot: shape mismatch");
  for (int j = 0; j < A->k; ++j) {
    const int64_t n = A->nlocal;
    if (A->is_complex) {
      const cplx *a = reinterpret_cast<const cplx*>(A->col(j)), *b = reinterpret_cast<const cplx*>(B->col(j));
      double sr = 0.0, si = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : sr, si)
      for (int64_t i = 0; i < n; ++i) { const cplx t = std::conj(a[i]) * b[i]; sr += t.real(); si += t.imag(); }
      res[2 * j] = sr; res[2 * j + 1] = si;
    } else {
      const double *a = A->col(j), *b = B->col(j);
      double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
      for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
      res[j] = s;
    }
  }
  return 0;
}
int mxg_mv_norm2(const mxg_mv* v, double* res) {
  for (int j = 0; j < v->k; ++j) {
    const double* a = v->col(j);
    const int64_t n = v->nlocal * v->w();
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (int64_t i = 0; i < n; ++i) s += a[i] * a[i];
    res[j] = std::sqrt(s);
  }
  return 0;
}
int mxg_mv_set_block(mxg_mv* self, const mxg_mv* A, const int* index, int n) {
  CHECK(A->nlocal == self->nlocal && A->is_complex == self->is_complex && n <= A->k, "mxg_mv_set_block: shape mismatch");
  for (int j = 0; j < n; ++j) {
    CHECK(index[j] >= 0 && index[j] < self->k, "mxg_mv_set_block: index out of range");
    memmove(self->col(index[j]), A->col(j), sizeof(double) * A->nlocal * A->w());
  }
  return 0;
}
// this = alpha A B + beta this, B m x k column-major (ldb), complex B interleaved
int mxg_mv_times_mat(double alr, double ali, const mxg_mv* A, const double* B, int ldb, int m, int k, double ber, double bei,
                     mxg_mv* self) {
  CHECK(A->k == m && self->k == k && A->nlocal == self->nlocal && A->is_complex == self->is_complex, "mxg_mv_times_mat: shape mismatch");
  const int64_t n = self->nlocal;
  // A may alias this (Anasazi's MvTimesMatAddMv contract allows it only for distinct objects, but the solver's
  // views overlap): work row-wise through a private copy of the row of A
  if (!self->is_complex) {
#pragma omp parallel
    {
      std::vector<double> row(m), outv(k);
#pragma omp for schedule(static)
      for (int64_t i = 0; i < n; ++i) {
        for (int l = 0; l < m; ++l) row[l] = A->col(l)[i];
        for (int j = 0; j < k; ++j) {
          double s = 0.0;
          for (int l = 0; l < m; ++l) s += row[l] * B[l + (size_t)ldb * j];
          outv[j] = alr * s + (ber != 0.0 ? ber * self->col(j)[i] : 0.0);
        }
        for (int j = 0; j < k; ++j) self->col(j)[i] = outv[j];
      }
    }
  } else {
    const cplx al(alr, ali), be(ber, bei);
    const cplx* Bc = reinterpret_cast<const cplx*>(B);
#pragma omp parallel
    {
      std::vector<cplx> row(m), outv(k);
#pragma omp for schedule(static)
      for (int64_t i = 0; i < n; ++i) {
        for (int l = 0; l < m; ++l) row[l] = reinterpret_cast<const cplx*>(A->col(l))[i];
        for (int j = 0; j < k; ++j) {
          cplx s = 0.0;
          for (int l = 0; l < m; ++l) s += row[l] * Bc[l + (size_t)ldb * j];
          outv[j] = al * s + (be != cplx(0.0) ? be * reinterpret_cast<cplx*>(self->col(j))[i] : cplx(0.0));
        }
        for (int j = 0; j < k; ++j) reinterpret_cast<cplx*>(self->col(j))[i] = outv[j];
      }
    }
  }
  return 0;
}
// B((m + m2) x k) = alpha [A A2]^H this
int mxg_mv_trans_mv2(double alr, double ali, const mxg_mv* A, const mxg_mv* A2, const mxg_mv* self, double* B, int ldb) {
  const int m = A->k, m2 = A2 ? A2->k : 0, k = self->k;
  const int64_t n = self->nlocal;
  CHECK(A->nlocal == n && A->is_complex == self->is_complex && (!A2 || (A2->nlocal == n && A2->is_complex == self->is_complex)),
        "mxg_mv_trans_mv: shape mismatch");
  auto acol = [&](int l) { return l < m ? A->col(l) : A2->col(l - m); };
  for (int l = 0; l < m + m2; ++l)
    for (int j = 0; j < k; ++j) {
      if (!self->is_complex) {
        const double *a = acol(l), *t = self->col(j);
        double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
        for (int64_t i = 0; i < n; ++i) s += a[i] * t[i];
        B[l + (size_t)ldb * j] = alr * s;
      } else {
        const cplx *a = reinterpret_cast<const cplx*>(acol(l)), *t = reinterpret_cast<const cplx*>(self->col(j));
        double sr = 0.0, si = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : sr, si)
        for (int64_t i = 0; i < n; ++i) { const cplx v = std::conj(a[i]) * t[i]; sr += v.real(); si += v.imag(); }
        const cplx r = cplx(alr, ali) * cplx(sr, si);
        B[2 * (l + (size_t)ldb * j)] = r.real(); B[2 * (l + (size_t)ldb * j) + 1] = r.imag();
      }
    }
  return 0;
}
int mxg_mv_trans_mv(double alr, double ali, const mxg_mv* A, const mxg_mv* self, double* B, int ldb) {
  return mxg_mv_trans_mv2(alr, ali, A, nullptr, self, B, ldb);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------- multigrid
// The V-cycle of maxwell_amd/csrc/device/mxg_solve.hip (mg_cycle / mg_smooth_cheb), same parameters, on host loops.
struct mxg_mg {
  mxg_ctx* ctx = nullptr;
  int nlevels = 0;
  std::vector<mxg_csr*> A, P, R;
  std::vector<std::vector<double> > r, z, t, u;
  int pre = 2, coarse = 20;
  double omega = 0.7, cscale = 1.9;
  bool cheb = false;
  double cheb_ratio = 20.0, coarse_cheb_ratio = 50.0;
  int coarse_cheb_degree = 0;
  std::vector<double> lmax;
};

static void mg_smooth_cheb(mxg_mg* mg, int l, int sweeps, bool zero_guess, double ratio) {
  const mxg_csr* A = mg->A[l];
  const int64_t n = A->nrows;
  const double* diag = A->diag.data();
  const double lmax = mg->lmax[l], lmin = lmax / ratio;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma1 = theta / delta;
  double rho = 1.0 / sigma1;
  bool prev_zero = false;
  std::vector<double>&rr = mg->r[l], &u = mg->u[l];
  for (int s = 0; s < sweeps; ++s) {
    double omega, beta;
    if (s == 0) { omega = 1.0 / theta; beta = 0.0; }
    else { const double rn = 1.0 / (2.0 * sigma1 - rho); omega = 2.0 * rn / delta; beta = rn * rho; rho = rn; }
    double* z = mg->z[l].data();
    double* t = mg->t[l].data();
    if (s == 0 && zero_guess) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) z[i] = diag[i] != 0.0 ? omega * rr[i] / diag[i] : 0.0;
      prev_zero = true;
      continue;
    }
    csr_apply_col(A, z, u.data());
    const bool pz = s == 0 || prev_zero;
    const double b2 = s == 0 ? 0.0 : beta;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      const double d = diag[i], zi = z[i], prev = pz ? 0.0 : t[i];
      t[i] = d != 0.0 ? zi + b2 * (zi - prev) + omega * (rr[i] - u[i]) / d : zi;
    }
    prev_zero = false;
    mg->z[l].swap(mg->t[l]);
  }
}

static void mg_smooth(mxg_mg* mg, int l, int sweeps, bool zero_guess) {
  if (mg->cheb && l < mg->nlevels - 1) return mg_smooth_cheb(mg, l, sweeps, zero_guess, mg->cheb_ratio);
  if (mg->cheb && mg->coarse_cheb_degree > 0) return mg_smooth_cheb(mg, l, mg->coarse_cheb_degree, zero_guess, mg->coarse_cheb_ratio);
  const mxg_csr* A = mg->A[l];
  const int64_t n = A->nrows;
  const double* diag = A->diag.data();
  for (int s = 0; s < sweeps; ++s) {
    double* z = mg->z[l].data();
    if (s == 0 && zero_guess) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) z[i] = diag[i] != 0.0 ? mg->omega * mg->r[l][i] / diag[i] : 0.0;
      continue;
    }
    csr_apply_col(A, z, mg->t[l].data());
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
      if (diag[i] != 0.0) z[i] += mg->omega * (mg->r[l][i] - mg->t[l][i]) / diag[i];
  }
}

static void mg_cycle(mxg_mg* mg, int l) {
  if (l == mg->nlevels - 1) return mg_smooth(mg, l, mg->coarse, true);
  const int64_t n = mg->A[l]->nrows;
  mg_smooth(mg, l, mg->pre, true);
  csr_apply_col(mg->A[l], mg->z[l].data(), mg->t[l].data());
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) mg->t[l][i] = mg->r[l][i] - mg->t[l][i];
  csr_apply_col(mg->R[l], mg->t[l].data(), mg->r[l + 1].data());
  mg_cycle(mg, l + 1);
  csr_apply_col(mg->P[l], mg->z[l + 1].data(), mg->t[l].data());
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) mg->z[l][i] += mg->cscale * mg->t[l][i];
  mg_smooth(mg, l, mg->pre, false);
}

extern "C" {

int mxg_mg_create(mxg_ctx* c, int nlevels, mxg_csr* const* A, mxg_csr* const* P, mxg_csr* const* R, int pre, int post,
                  int coarse, double omega, double cscale, mxg_mg** out) {
  CHECK(c && A && out && nlevels >= 1 && pre == post, "mxg_mg_create: bad argument");
  mxg_mg* mg = new mxg_mg();
  mg->ctx = c; mg->nlevels = nlevels; mg->pre = pre; mg->coarse = coarse; mg->omega = omega; mg->cscale = cscale;
  for (int l = 0; l < nlevels; ++l) {
    CHECK(A[l] && !A[l]->is_complex && A[l]->nrows == A[l]->ncols, "mxg_mg_create: level %d must be real and square", l);
    mg->A.push_back(A[l]);
    if (l + 1 < nlevels) { mg->P.push_back(P[l]); mg->R.push_back(R[l]); }
    const size_t n = (size_t)std::max<int64_t>(A[l]->nrows, 1);
    mg->r.emplace_back(n, 0.0); mg->z.emplace_back(n, 0.0); mg->t.emplace_back(n, 0.0); mg->u.emplace_back(n, 0.0);
  }
  *out = mg;
  return 0;
}
int mxg_mg_destroy(mxg_mg* mg) { delete mg; return 0; }
int mxg_mg_set_chebyshev(mxg_mg* mg, double ratio) {
  CHECK(mg && ratio > 1.0, "mxg_mg_set_chebyshev: bad argument");
  mg->cheb_ratio = ratio;
  mg->lmax.assign(mg->nlevels, 2.0);
  for (int l = 0; l < mg->nlevels; ++l) {       // lmax(D^-1 A) by 15 power iterations from the device library's start vector
    const mxg_csr* A = mg->A[l];
    const int64_t n = A->nrows;
    std::vector<double> v((size_t)n), w((size_t)n), u((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
      unsigned long long h = splitmix64((unsigned long long)(777 + l) ^ splitmix64(1));
      h = splitmix64(h + (unsigned long long)i);
      v[i] = 2.0 * (((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0)) - 1.0;
    }
    double lam = 1.0;
    for (int it = 0; it < 15; ++it) {
      double nv = 0.0;
      for (int64_t i = 0; i < n; ++i) nv += v[i] * v[i];
      nv = std::sqrt(nv);
      if (!(nv > 0.0)) break;
      for (int64_t i = 0; i < n; ++i) v[i] /= nv;
      csr_apply_col(A, v.data(), u.data());
      double nw = 0.0;
      for (int64_t i = 0; i < n; ++i) { w[i] = A->diag[i] != 0.0 ? u[i] / A->diag[i] : 0.0; nw += w[i] * w[i]; }
      lam = std::sqrt(nw);
      v.swap(w);
    }
    mg->lmax[l] = 1.1 * lam;
  }
  mg->coarse_cheb_degree = 16;
  mg->coarse_cheb_ratio = 50.0;
  mg->cheb = true;
  return 0;
}
int mxg_mg_set_tail(mxg_mg*, mxg_mg*, const int64_t*) { UNSUPPORTED("mxg_mg_set_tail"); }
int mxg_mg_apply(mxg_ctx*, mxg_mg* mg, const mxg_mv* r, mxg_mv* z) {
  CHECK(mg && r && z && r->k == 1 && z->k == 1 && r->nlocal == mg->A[0]->nrows && z->nlocal == r->nlocal && r->is_complex == z->is_complex,
        "mxg_mg_apply: shape mismatch");
  const int64_t n = r->nlocal;
  if (r->is_complex) {          // real hierarchy on a complex residual: one cycle per part
    for (int part = 0; part < 2; ++part) {
      for (int64_t i = 0; i < n; ++i) mg->r[0][i] = r->col(0)[2 * i + part];
      mg_cycle(mg, 0);
      for (int64_t i = 0; i < n; ++i) z->col(0)[2 * i + part] = mg->z[0][i];
    }
    return 0;
  }
  memcpy(mg->r[0].data(), r->col(0), sizeof(double) * n);
  mg_cycle(mg, 0);
  memcpy(z->col(0), mg->z[0].data(), sizeof(double) * n);
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------- inner solves
// the loops are the product's own (mxg_krylov.h is written on this ABI); only the preconditioner hooks are local
#include "mxg_krylov.h"

static int apply_prec(mxg_ctx* c, const mxg_csr* A, const mxg_solve_opts* o, const mxg_mv* r, mxg_mv* z) {
  if (o->precond == 1) {
    const int64_t n = A->nrows;
    const double* d = A->diag.data();
    const double* rp = r->col(0);
    double* zp = z->col(0);
    if (!A->is_complex) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) zp[i] = d[i] != 0.0 ? rp[i] / d[i] : 0.0;
    } else {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) {
        const double dr = d[2 * i], di = d[2 * i + 1], den = dr * dr + di * di, rr = rp[2 * i], ri = rp[2 * i + 1];
        zp[2 * i] = den != 0.0 ? (rr * dr + ri * di) / den : 0.0;
        zp[2 * i + 1] = den != 0.0 ? (ri * dr - rr * di) / den : 0.0;
      }
    }
    return 0;
  }
  if (o->precond == 3) {
    CHECK(o->mg != nullptr, "precond == 3 needs opts->mg");
    return mxg_mg_apply(c, o->mg, r, z);
  }
  return mxg_mv_copy(z, r);
}

extern "C" {

int mxg_solve_cg(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, mxg_mv* x, const mxg_solve_opts* o, mxg_solve_info* info) {
  CHECK(c && A && b && x && o && info && A->nrows == A->ncols && b->k == x->k && b->nlocal == A->nrows, "mxg_solve_cg: bad argument");
  return mxg_krylov::solve_cg(c, A, b, x, o, info, [&](const mxg_mv* r, mxg_mv* z) { return apply_prec(c, A, o, r, z); },
                              [](mxg_mv*, mxg_mv*, mxg_mv*, mxg_mv*, mxg_mv*, mxg_mv*, int, int*, double*) { return -1; });
}
int mxg_solve_bicgstab(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, mxg_mv* x, const mxg_solve_opts* o, mxg_solve_info* info) {
  CHECK(c && A && b && x && o && info && A->nrows == A->ncols && b->k == x->k && b->nlocal == A->nrows, "mxg_solve_bicgstab: bad argument");
  return mxg_krylov::solve_bicgstab(c, A, b, x, o, info, [&](const mxg_mv* r, mxg_mv* z) { return apply_prec(c, A, o, r, z); });
}
int mxg_solve_gmres(mxg_ctx* c, const mxg_csr* A, const mxg_mv* b, mxg_mv* x, const mxg_solve_opts* o, mxg_solve_info* info) {
  CHECK(c && A && b && x && o && info && A->nrows == A->ncols && b->k == x->k && b->nlocal == A->nrows, "mxg_solve_gmres: bad argument");
  return mxg_krylov::solve_gmres(c, A, b, x, o, info, [&](mxg_mv* v, mxg_mv* z, mxg_mv** zc) {
    *zc = z;
    return apply_prec(c, A, o, v, z);
  });
}

}  // extern "C"
