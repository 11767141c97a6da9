// Internal declarations shared by the translation units of libmxg.so.
#ifndef MXG_INTERNAL_H
#define MXG_INTERNAL_H

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <mutex>
#include <utility>
#include <vector>

#include "mxg.h"

void mxg_set_error(const char* fmt, ...);

#define MXG_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      mxg_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,                 \
                    hipGetErrorString(e_));                                            \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

#define MXG_NCCL(call)                                                                 \
  do {                                                                                 \
    ncclResult_t e_ = (call);                                                          \
    if (e_ != ncclSuccess) {                                                           \
      mxg_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,                 \
                    ncclGetErrorString(e_));                                           \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

#define MXG_CHECK(cond, ...)                                                           \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      mxg_set_error(__VA_ARGS__);                                                      \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

#define MXG_TRY(call)                                                                  \
  do {                                                                                 \
    int rc_ = (call);                                                                  \
    if (rc_ != 0) return rc_;                                                          \
  } while (0)

constexpr int MXG_NUM_XCD = 8;          // MI355X: 8 XCDs, workgroups dealt round-robin
constexpr int MXG_MAX_RED_BLOCKS = 2048; // partial-sum slots of the two-stage reductions

struct mxg_ctx {
  int device = 0;
  int rank = 0;
  int nranks = 1;
  hipStream_t stream = nullptr;
  ncclComm_t comm = nullptr;
  struct mxg_local_world* lw = nullptr;   // in-process world (ranks = threads sharing the device), mxg_comm.hip
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // halo exchange runs on its own stream so that interior rows overlap it (ev_pack / ev_halo order the two)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
  hipEvent_t ev_scal = nullptr;           // "the scalars of this iteration are on the host" (multigrid-CG convergence test)
  // released vector storage, kept for the next vector of the same size (mxg_vec_alloc / mxg_vec_release): the work
  // vectors of an inner solve and the clones of the eigensolver come and go per call, and hipFree stalls the device
  std::vector<std::pair<size_t, void*> > vec_cache;
  size_t vec_cache_bytes = 0;
  std::mutex vec_cache_mutex;             // a binding's finaliser may release a vector from another thread
  // reduction workspace: partial sums + results (device), results (pinned host)
  double* d_partial = nullptr;   // MXG_MAX_RED_BLOCKS * red_width doubles
  double* d_result = nullptr;    // red_width doubles
  double* h_result = nullptr;    // pinned, red_width doubles
  double* h_result_dev = nullptr;   // the same buffer as the device sees it (kernels store results straight into it)
  size_t red_width = 0;          // capacity (number of simultaneous scalar results)
  // small dense operand staging (B of K6) and int index staging
  double* d_small = nullptr;
  size_t small_cap = 0;          // doubles
  // scratch vector storage for aliasing-safe applies (grown on demand)
  double* d_scratch = nullptr;
  size_t scratch_cap = 0;        // doubles
  int num_cu = 256;
};

// View descriptor passed by value to kernels.
struct MvDesc {
  double* base;        // storage of column 0 of the owning allocation
  long long ld;        // leading dimension in doubles (already x2 for complex)
  long long n;         // local scalar rows
  int k;               // columns in this view
  int cplx;            // 0 real, 1 complex (interleaved)
  int col0;            // first column when cols == nullptr (contiguous range)
  const int* cols;     // device array of k column indices, or nullptr
  __host__ __device__ inline double* col(int j) const {
    long long c = cols ? (long long)cols[j] : (long long)(col0 + j);
    return base + c * ld;
  }
};

struct mxg_mv {
  mxg_ctx* ctx = nullptr;
  double* base = nullptr;     // allocation (owner) or parent's allocation (view)
  int64_t nlocal = 0;         // scalar rows
  int64_t ld = 0;             // doubles between columns
  int ktotal = 0;             // columns in the allocation
  int k = 0;                  // columns in this object
  int is_complex = 0;
  bool owner = false;
  bool contiguous = true;
  int col0 = 0;
  std::vector<int> hcols;     // host copy of the column list (always filled, size k)
  int* dcols = nullptr;       // device copy when !contiguous
  MvDesc desc() const {
    MvDesc d;
    d.base = base; d.ld = ld; d.n = nlocal; d.k = k; d.cplx = is_complex;
    d.col0 = col0; d.cols = contiguous ? nullptr : dcols;
    return d;
  }
  double* colptr(int j) const { return base + (int64_t)hcols[j] * ld; }
};

struct mxg_csr {
  mxg_ctx* ctx = nullptr;
  int64_t nrows = 0, ncols_owned = 0, nghost = 0, nnz = 0;
  int is_complex = 0;
  int64_t* d_rowptr = nullptr;
  int32_t* d_colidx = nullptr;
  double* d_vals = nullptr;
  // row blocks of the stream kernel: block b covers rows [rowblk[b], rowblk[b+1])
  int32_t* d_rowblk = nullptr;
  int64_t nblocks = 0;
  int kernel = 0;
  // pipelined kernel: 32-byte block descriptors (built for the active geometry)
  void* d_blkdesc = nullptr;
  int64_t npipe = 0;
  int pipe_geo = -1;
  double* d_dotpartial = nullptr;   // per-block partials of the fused SpMV + dot
  int64_t dotpartial_cap = 0;
  unsigned short* d_roff = nullptr; // row-gather kernel: 16-bit in-block row offsets (nrows + nblocks + 1)
  unsigned short* d_roff16 = nullptr; // persistent row-gather kernel: the same, 264 entries per block (16-byte aligned pieces)
  // compressed-stream kernel (mxg_csr_compress): 32-byte block descriptors + packed per-block streams
  void* d_csdesc = nullptr;
  unsigned char* d_csstream = nullptr;
  int64_t ncs = 0, cs_fallback = 0, cs_bytes = 0;
  // injection form (every row has at most one entry: the piecewise-constant prolongations of the multigrid): built on
  // the first product with the coarse-correction epilogue; inj_state 0 unknown, 1 in use, -1 not an injection
  int inj_state = 0;
  int32_t* d_injcol = nullptr;          // column of the row's entry, -1 without one
  double* d_injval = nullptr;           // its value (nullptr when every entry has the value inj_uniform)
  double inj_uniform = 0.0;
  // ELL form of a short-row gather matrix (the multigrid restrictions: <= 16 entries per row), built on first use by
  // mxg_dev_csr_apply_ell; ell_state 0 unknown, 1 in use, -1 not suitable
  int ell_state = 0, ell_width = 0;
  int32_t* d_ellcol = nullptr;          // [width][nrows], -1 = no entry
  double* d_ellval = nullptr;           // [width][nrows]; nullptr when every entry has the value ell_uniform
  double ell_uniform = 0.0;
  // a matrix adopted from device arrays (the results of mxg_csr_multiply / add / purge) gets its apply-side structures
  // (row blocks, diagonal, pipeline descriptors) only when something applies / compresses it: the intermediates of the
  // operator chain never do (mxg_csr_materialize)
  int lazy = 0;
  int small_maxlen = -1;                // longest row, computed on demand for small matrices (mxg_solve.hip: one-workgroup smoother)
  // tile form (mxg_tile.hip): wavefront-uniform tiles of short (A: <= 8 entries) and long (B: 9..16) patterns, the
  // irregular rows as ELL groups of 64 rows (ghost-free groups first) and rows of more than 40 entries as "long rows"
  int tile_on = 0, tile_tried = 0;      // tile_tried: mxg_tile_build has looked at this matrix (it may have declined: coverage)
  void* d_tiledescA = nullptr; void* d_tiledictA = nullptr;
  void* d_tiledescB = nullptr; void* d_tiledictB = nullptr;
  int64_t tile_ntA = 0, tile_ntB = 0;
  int32_t* d_grow = nullptr;            // row ids, 64 per group (-1 padding)
  void* d_ggrp = nullptr;               // 16-byte group descriptors
  int32_t* d_gcol = nullptr;            // [group][width][64]
  double* d_gval = nullptr;
  int32_t* d_lrows = nullptr;
  int64_t gen_ngrp = 0, gen_ngrp_interior = 0, gen_nlong = 0, gen_nlong_interior = 0;
  int64_t pat_ntile = 0, pat_ntilepat = 0, pat_ngrows = 0, tile_bytes = 0;
  double pat_tile_coverage = 0.0;       // fraction of the rows inside tiles
  int64_t ncs_interior = 0;      // compressed descriptors [0, ncs_interior) reference no ghost column
  // diagonal (for Jacobi) extracted at creation
  double* d_diag = nullptr;
  // halo
  int nneigh = 0;
  std::vector<int> neigh;
  std::vector<int64_t> send_off, recv_off;
  int64_t* d_send_idx = nullptr;
  int64_t nsend = 0;
  double* d_sendbuf = nullptr;   // nsend * kcap scalars
  double* d_xext = nullptr;      // (ncols_owned + nghost) * kcap scalars: x with ghosts appended (kernels without the split)
  double* d_ghost = nullptr;     // nghost * kcap scalars: received ghosts only (row-gather kernel reads owned x in place)
  int64_t ldghost = 0;
  int64_t n_interior = 0;        // row-gather descriptors [0, n_interior) reference no ghost column
  int halo_kcap = 0;
};

bool mxg_ctx_alive(const mxg_ctx* c);

// collectives (mxg_comm.hip): RCCL, or the in-process world when the ctx was made by mxg_ctx_create_local
struct MxgXfer { int peer; const double* send; int64_t ns; double* recv; int64_t nr; };
bool mxg_comm_on(const mxg_ctx* c);
int mxg_comm_allreduce(mxg_ctx* c, double* d, size_t count, hipStream_t s);
int mxg_comm_exchange(mxg_ctx* c, const MxgXfer* x, int n, hipStream_t s);
// recv[r * n .. r * n + n) = send[0 .. n) of rank r (device buffers, equal counts)
int mxg_comm_allgather_dev(mxg_ctx* c, const double* send, size_t n, double* recv, hipStream_t s);

// reductions / workspace helpers (mxg_ctx.hip)
int mxg_ctx_reserve_reduce(mxg_ctx* ctx, size_t width);
int mxg_ctx_reserve_small(mxg_ctx* ctx, size_t doubles);
int mxg_ctx_reserve_scratch(mxg_ctx* ctx, size_t doubles);
// sum `width` device results across ranks (no-op for 1 rank), copy to host, sync
int mxg_ctx_finish_reduce(mxg_ctx* ctx, size_t width, double* host_out);

// device-side entry points used across translation units (results stay on device)
// d_out[j] = sum_i conj(A_j[i]) * B_j[i]; complex results interleaved. No all-reduce, no sync.
int mxg_dev_dot(mxg_ctx* ctx, const MvDesc& A, const MvDesc& B, double* d_out);
// vector storage through the context's cache (stream-ordered reuse on ctx->stream; contents undefined)
int mxg_vec_alloc(mxg_ctx* c, size_t bytes, void** out);
void mxg_vec_release(mxg_ctx* c, void* p, size_t bytes);
void mxg_vec_cache_flush(mxg_ctx* c);
// build the apply-side structures of an adopted matrix if that has not happened yet
int mxg_csr_materialize(mxg_csr* a);
// y = A x through the ELL form of a short-row matrix (0 done, 1 not applicable: use mxg_csr_apply, 2 error)
int mxg_dev_csr_apply_ell(mxg_ctx* c, mxg_csr* a, const mxg_mv* x, mxg_mv* y);

// y = A x with the p.Ap of CG fused in (mxg_csr.hip); d_out is a device scalar
// 0: launched; 1: matrix not compressed / shape not supported (nothing done); 2: launch error
int mxg_dev_csr_apply_epi(mxg_ctx* c, mxg_csr* a, const mxg_mv* x, mxg_mv* y, int epi, const double* rhs,
                          const double* diag, double omega, double beta = 0.0);
int mxg_dev_csr_apply_epi_dot(mxg_ctx* c, mxg_csr* a, const mxg_mv* x, mxg_mv* y, int epi, const double* rhs,
                              const double* diag, double omega, double beta, double* d_out);
int mxg_dev_csr_apply_dot(mxg_ctx* ctx, mxg_csr* a, const mxg_mv* x, mxg_mv* y, double* d_out);

// tile product (mxg_tile.hip)
int mxg_tile_build(mxg_csr* a, const int64_t* rowptr, const int32_t* colidx, const double* vals);
void mxg_tile_free(mxg_csr* a);
int64_t mxg_tile_partials(const mxg_csr* a);
int mxg_tile_launch(mxg_ctx* c, mxg_csr* a, const double* x, long long ldx, double* y, long long ldy, int k, int epi,
                    int dot, const double* xdot, double* partial, const double* rhs, const double* diag, double omega,
                    double beta, int halo, const double* xg, long long ldg);

// MXG_TRACE_SETUP=1: wall time of the setup-side entry points on stderr (where do the seconds of solver setup go)
struct MxgSetupTrace {
  const char* what; long long n; std::chrono::steady_clock::time_point t0; bool on;
  MxgSetupTrace(const char* w, long long n_) : what(w), n(n_), t0(std::chrono::steady_clock::now()) {
    static const bool enabled = getenv("MXG_TRACE_SETUP") != nullptr;
    on = enabled;
  }
  ~MxgSetupTrace() {
    if (on) { hipDeviceSynchronize(); fprintf(stderr, "[mxg setup] %-28s %12lld  %.3f s\n", what, n, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
  }
};

inline int mxg_grid_for(int64_t n, int block, int per_thread, int cap) {
  int64_t g = (n + (int64_t)block * per_thread - 1) / ((int64_t)block * per_thread);
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

#endif
