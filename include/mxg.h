/*
 * mxg.h -- C ABI of the MI355X device library (libmxg.so) for maxwell's Krylov hot path.
 *
 * This is the drop-in boundary: every entry point replaces one Trilinos call site that the
 * reference's MxCrsMatrix / MxMultiVector / MxAnasaziMV / MxMagWaveOp wrappers forward to
 * (citations are file:line in the upstream bauerca/maxwell tree).  Plain pointers and sizes
 * only; no C++ or torch types.  The C++ host classes in include/mx/ are written on top of
 * exactly these functions, INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; mxg_last_error() gives the text
 *   - one mxg_ctx per GPU / rank; all objects belong to a ctx and run on its stream
 *   - device memory is owned by the library; host data crosses only through the put/fetch calls,
 *     small dense blocks (m x k) and scalar results
 *   - multivectors are column-major: column j is one contiguous array of nlocal scalars
 *     (Epetra_MultiVector layout, src/MxMultiVector.cpp:12-24); complex scalars are interleaved
 *     (re, im) pairs, i.e. rows 2i / 2i+1 of the reference's doubled real map
 *     (src/MxMultiVector.hpp:127-130,175-182, src/MxMap.cpp:97-100)
 *   - complex scalar arguments are passed as (re, im) double pairs; complex arrays as interleaved
 *     doubles; for real objects the imaginary parts are ignored
 *   - collective calls (dot, norm2, trans_mv, csr_apply with a halo, solves) must be made by all
 *     ranks in the same order, like the MPI collectives inside Epetra
 */
#ifndef MXG_H
#define MXG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxg_ctx mxg_ctx;
typedef struct mxg_csr mxg_csr;
typedef struct mxg_mv mxg_mv;

#define MXG_UNIQUE_ID_BYTES 128

/* ------------------------------------------------------------------------------------------
 * context  (replaces MxComm / Epetra_MpiComm, src/MxComm.hpp:14-37)
 * ---------------------------------------------------------------------------------------- */
const char* mxg_last_error(void);
/* fills MXG_UNIQUE_ID_BYTES bytes with an RCCL unique id (rank 0 calls it, then broadcasts) */
int mxg_get_unique_id(void* out_id);
/* nccl_unique_id may be NULL when nranks == 1 (no communicator is created) */
int mxg_ctx_create(int device, int rank, int nranks, const void* nccl_unique_id, mxg_ctx** out);
int mxg_ctx_destroy(mxg_ctx* ctx);
int mxg_ctx_sync(mxg_ctx* ctx);
int mxg_ctx_rank(const mxg_ctx* ctx);
int mxg_ctx_nranks(const mxg_ctx* ctx);
/* ranks the RCCL communicator reports (ncclCommCount); 1 when there is none, -1 on error */
int mxg_ctx_comm_ranks(const mxg_ctx* ctx);
/* HIP-event timer on the ctx stream (the stream every kernel of this ctx is launched on) */
int mxg_timer_start(mxg_ctx* ctx);
int mxg_timer_stop(mxg_ctx* ctx, float* elapsed_ms);
/* Comm::Barrier (src/MxMagWaveOp.cpp:71,288): stream sync + zero-byte all-reduce */
int mxg_barrier(mxg_ctx* ctx);
/* Measurement aid (bench.py): what a pure stream reaches on THIS device - read_bytes read once with 16-byte
 * non-temporal loads, write_bytes written once, no gathers - averaged over `reps` passes (HIP events).  The devices
 * of a pool differ by several per cent; the SpMV roofline fraction is reported beside this ceiling. */
int mxg_bench_stream(mxg_ctx* ctx, int64_t read_bytes, int64_t write_bytes, int reps, float* ms_per_pass);

/* In-process world: the ranks are threads of one process sharing `device` (several MxGridDomain boxes on one
 * GPU, src/MxGrid.cpp:171-193).  Same collectives as the RCCL world, staged through the host; every rank
 * thread must issue them in the same order.  Create one world, then one ctx per rank thread. */
typedef struct mxg_local_world mxg_local_world;
int mxg_local_world_create(int nranks, mxg_local_world** out);
int mxg_local_world_destroy(mxg_local_world* w);
int mxg_ctx_create_local(int device, int rank, int nranks, mxg_local_world* w, mxg_ctx** out);
/* Setup-time collectives on host data (the plan agreement Epetra's FillComplete does with MPI,
 * src/MxCrsMatrix.cpp:325-342): one int64 per rank; variable-length byte strings concatenated in rank
 * order (counts[r] = bytes of rank r); sum of doubles.  No-ops on a single rank. */
int mxg_comm_allgather_i64(mxg_ctx* ctx, int64_t v, int64_t* out);
int mxg_comm_allgatherv(mxg_ctx* ctx, const void* send, int64_t nbytes, const int64_t* counts, void* recv);
int mxg_comm_allreduce_host(mxg_ctx* ctx, double* v, int n);

/* ------------------------------------------------------------------------------------------
 * halo plan  (replaces the Epetra_Import built by FillComplete, src/MxCrsMatrix.cpp:325-342)
 *
 * Local column numbering: [0, nowned) owned entries, [nowned, nowned + nghost) ghosts grouped
 * by source neighbour in the order of `neigh`.  send_idx lists the owned local rows packed for
 * each neighbour (send_off[i] .. send_off[i+1]); recv_off gives the ghost range of each
 * neighbour relative to nowned.
 * ---------------------------------------------------------------------------------------- */
typedef struct mxg_halo_plan {
  int nneigh;
  const int* neigh;           /* [nneigh] peer ranks */
  const int64_t* send_off;    /* [nneigh+1] */
  const int64_t* send_idx;    /* [send_off[nneigh]] owned local indices to pack */
  const int64_t* recv_off;    /* [nneigh+1] ghost offsets (relative to the first ghost) */
} mxg_halo_plan;

/* ------------------------------------------------------------------------------------------
 * CSR matrix  (replaces Epetra_CrsMatrix behind MxCrsMatrix, src/MxCrsMatrix.hpp:11-104)
 * ---------------------------------------------------------------------------------------- */
/* rowptr: nrows+1 int64; colidx: local column indices (< ncols_owned + nghost); vals: nnz doubles
 * (real) or nnz (re, im) pairs (is_complex).  plan may be NULL when nghost == 0.
 * Duplicates must already be summed and rows sorted or not -- order inside a row is kept and is
 * the summation order of the product (bit-exact with the serial row loop of the oracle). */
int mxg_csr_create(mxg_ctx* ctx, int64_t nrows, int64_t ncols_owned, int64_t nghost,
                   const int64_t* rowptr, const int32_t* colidx, const double* vals,
                   int is_complex, const mxg_halo_plan* plan, mxg_csr** out);
int mxg_csr_destroy(mxg_csr* a);
/* Setup-time sparse algebra on the device (matrices without ghosts; both operands real or both complex), bit-identical
 * to the host versions (include/mx/mx_sparse.hpp; complex products are (ac - bd, ad + bc), |a| is hypot(re, im)):
 *   multiply: C = A B        MxCrsMatrix::matrixMatrixMultiply (src/MxCrsMatrix.cpp:361-382; the chain of
 *                            MxUtil::Trilinos::massiveCrsMultiply, src/MxUtil.cpp:318-371)
 *   add:      C = sa A + sb B MxCrsMatrix::matrixMatrixAdd (src/MxCrsMatrix.cpp:404-457; setShift, src/MxMagWaveOp.cpp:248-255)
 *   purge:    keep |a| > tol MxCrsMatrix::purgeZeros (src/MxCrsMatrix.cpp:84-116)
 * Operands: columns ascending and unique inside every row (what fillComplete leaves; the diagonal-factor fast path of
 * multiply copies the other factor's pattern as it stands).  The result is a new matrix owned by the caller (mxg_csr_destroy).  download copies the CSR arrays to the host
 * (NULL = not wanted): MxCrsMatrix::getLocalRow / save and the multigrid setup read them. */
int mxg_csr_multiply(mxg_ctx* ctx, const mxg_csr* a, const mxg_csr* b, mxg_csr** out);
int mxg_csr_add(mxg_ctx* ctx, const mxg_csr* a, double sa, const mxg_csr* b, double sb, mxg_csr** out);
/* the same with (re, im) scalars - the Scalar arguments of matrixMatrixAdd on complex matrices */
int mxg_csr_add_z(mxg_ctx* ctx, const mxg_csr* a, const double sa[2], const mxg_csr* b, const double sb[2], mxg_csr** out);
int mxg_csr_purge(mxg_ctx* ctx, const mxg_csr* a, double tol, mxg_csr** out);
int mxg_csr_download(const mxg_csr* a, int64_t* rowptr, int32_t* colidx, double* vals);
int mxg_csr_info(const mxg_csr* a, int64_t* nrows, int64_t* ncols_owned, int64_t* nghost,
                 int64_t* nnz, int* is_complex);
/* y = A x  (SpMV for 1 column, SpMM for k columns).  MxCrsMatrix<S>::apply,
 * src/MxCrsMatrix.cpp:347-353 -> Epetra_CrsMatrix::Apply; x may alias y
 * (src/MxMagWaveOp.cpp:650,1249).  With a halo plan the ghost import (C1) runs first. */
int mxg_csr_apply(mxg_ctx* ctx, const mxg_csr* a, const mxg_mv* x, mxg_mv* y);
/* algorithmic bytes of one apply with k columns (SURVEY section 8(d) formula):
 * nnz*(vb+4) + nrows*(8 + sb*k) + ncols*sb*k, vb = sb = 8 (real) or 16 (complex) */
int64_t mxg_csr_apply_bytes(const mxg_csr* a, int k);
/* Which CSR kernel mxg_csr_apply runs on the plain arrays: 0 default (real: row-gather kernel with 16-bit in-block
 * row offsets; complex: persistent LDS-DMA row-gather kernel), 1 real row-gather kernel reading the int64 row
 * pointers of the CSR arrays, 2 complex one-block-per-workgroup stream kernel.  Any call makes a compressed form
 * inactive again (products run on the CSR arrays) until the next mxg_csr_compress. */
int mxg_csr_set_kernel(mxg_csr* a, int which);
/* name of the kernel mxg_csr_apply launches for this matrix in its present state (for bench / profile records) */
const char* mxg_csr_kernel_name(const mxg_csr* a);
/* Opt-in re-encoding of a matrix for the products of the solver loops.  Lattice operators (rows that repeat along
 * grid lines, real or complex) take the TILE form: 16-byte tile descriptors + a dictionary of wavefront-uniform
 * patterns, irregular rows (cavity wall, line ends, rows reading ghost columns) as ELL groups of 64 rows - no matrix
 * stream at all for the bulk.  Other real matrices take the COMPRESSED STREAM: per block of <= 256 rows the column
 * indices become 16-bit band codes and the values 8-bit indices into the block's table of distinct values (3 bytes per
 * entry instead of 12; blocks that do not fit keep plain CSR).  Same values, columns and summation order either way:
 * results are bit-identical to the CSR kernels.  Returns work units (tiles + groups, or blocks), fall-back blocks and
 * the size of the re-encoded form.  Used by the eigen-driver for its operators and multigrid levels; the headline
 * SpMV measurement of bench.py stays on the plain CSR arrays. */
int mxg_csr_compress(mxg_csr* a, int64_t* nblocks, int64_t* nfallback, int64_t* stream_bytes);
/* same, encoding from the caller's host copy of the CSR arrays given to mxg_csr_create (saves reading them back) */
/* Which product kernel a compressed matrix runs: returns 0 plain CSR, 1 compressed stream (k_spmv_cs), 3 tile patterns
 * (k_spmv_tile: lattice operators whose rows repeat along grid lines; MXG_PAT=0 at compress time keeps form 1).
 * tiles / patterns / irregular_rows / coverage (fraction of the rows inside tiles) describe form 3; any may be NULL. */
int mxg_csr_product_form(const mxg_csr* a, int64_t* tiles, int64_t* patterns, int64_t* irregular_rows, double* coverage);
int mxg_csr_compress_from(mxg_csr* a, const int64_t* rowptr, const int32_t* colidx, const double* vals,
                          int64_t* nblocks, int64_t* nfallback, int64_t* stream_bytes);
/* TEST HOOK, no GPU needed, not a product path (nothing in the library calls it): builds the tile form's host-side
 * plan for the given CSR arrays (columns >= ncols_owned are ghosts, read from xghost) and evaluates y = A x by
 * walking the PLAN with the kernels' arithmetic, so that the setup code can be checked against the oracle in a
 * CPU-only container (tests/test_tile_plan.py).  stats[9] = {short tiles, long tiles, patterns, groups, ghost-free
 * groups, long rows, rows in tiles, rows the plan missed, plan errors}.  Returns 0 when the plan is consistent. */
int mxg_debug_tile_plan_apply(int64_t nrows, int64_t ncols_owned, const int64_t* rowptr, const int32_t* colidx,
                              const double* vals, int is_complex, const double* x, const double* xghost, double* y,
                              int64_t* stats);

/* ------------------------------------------------------------------------------------------
 * multivector  (replaces Epetra_MultiVector behind MxMultiVector / MxAnasaziMV,
 *               src/MxMultiVector.hpp:14-109, src/MxAnasaziMV.hpp:11-138)
 * ---------------------------------------------------------------------------------------- */
int mxg_mv_create(mxg_ctx* ctx, int64_t nlocal, int k, int is_complex, mxg_mv** out);
/* aliasing view of arbitrary columns (Epetra View ctor, src/MxMultiVector.cpp:29-43) */
int mxg_mv_view(mxg_mv* parent, const int* cols, int n, mxg_mv** out);
/* deep copy of selected columns (cols == NULL: all) (Epetra Copy ctor, same lines; :48-58) */
int mxg_mv_clone(const mxg_mv* src, const int* cols, int n, mxg_mv** out);
int mxg_mv_destroy(mxg_mv* v);
int mxg_mv_info(const mxg_mv* v, int64_t* nlocal, int* k, int* is_complex);
/* host <-> device slabs of one column: rows [row0, row0+nrows) scalars
 * (replaceLocalValue / operator(), src/MxMultiVector.hpp:119-183; MxIO, src/MxIO.cpp:167-197) */
int mxg_mv_put(mxg_mv* v, int col, int64_t row0, int64_t nrows, const double* host);
int mxg_mv_fetch(const mxg_mv* v, int col, int64_t row0, int64_t nrows, double* host);
/* this = src (same shape); operator=, src/MxMultiVector.cpp:61-83 */
int mxg_mv_copy(mxg_mv* dst, const mxg_mv* src);
/* set / MvInit, src/MxMultiVector.cpp:86-91, src/MxAnasaziMV.hpp:125-132 */
int mxg_mv_fill(mxg_mv* v, double re, double im);
/* Epetra_MultiVector::Random, src/MxMultiVector.hpp:41: uniform (-1,1), counter based */
int mxg_mv_random(mxg_mv* v, uint64_t seed);
/* conj, src/MxMultiVector.cpp:131-142 */
int mxg_mv_conj(mxg_mv* v);
/* this *= a, src/MxMultiVector.cpp:102-117 */
int mxg_mv_scale(mxg_mv* v, double re, double im);
/* this[:,j] *= alphas[j], src/MxMultiVector.cpp:120-126 (alphas: k scalars, interleaved if complex) */
int mxg_mv_scale_cols(mxg_mv* v, const double* alphas);
/* zero the rows flagged in mask (device-resident copy is kept by the caller as an mv of 1 column
 * with 0/1 entries): this[i,:] *= mask[i]; zeroUnusedComponents, src/MxSolver.cpp:65 */
int mxg_mv_mask_rows(mxg_mv* v, const mxg_mv* mask);
/* this = a*A + b*this; update, src/MxMultiVector.cpp:212-227 (A may alias this) */
int mxg_mv_axpby(double a_re, double a_im, const mxg_mv* A, double b_re, double b_im, mxg_mv* self);
/* this = a*A + b*B; MvAddMv contract, src/MxAnasaziMV.cpp:89-111 (any aliasing allowed) */
int mxg_mv_add(double a_re, double a_im, const mxg_mv* A, double b_re, double b_im, const mxg_mv* B,
               mxg_mv* self);
/* res[j] = A_j^H . B_j (k scalars); dot / MvDot, src/MxMultiVector.cpp:179-203; all-reduced */
int mxg_mv_dot(const mxg_mv* A, const mxg_mv* B, double* res);
/* res[j] = ||v_j||_2 (k doubles); norm2 / MvNorm, src/MxMultiVector.cpp:148-155; all-reduced */
int mxg_mv_norm2(const mxg_mv* v, double* res);
/* this[:, index[j]] = A[:, j]; SetBlock, src/MxAnasaziMV.cpp:201-213 */
int mxg_mv_set_block(mxg_mv* self, const mxg_mv* A, const int* index, int n);
/* this = alpha*A*B + beta*this, B is m x k column-major with leading dimension ldb;
 * MvTimesMatAddMv, src/MxAnasaziMV.cpp:8-86 */
int mxg_mv_times_mat(double alpha_re, double alpha_im, const mxg_mv* A, const double* B, int ldb,
                     int m, int k, double beta_re, double beta_im, mxg_mv* self);
/* B(m x k) = alpha * A^H * this, column-major ldb; MvTransMv, src/MxAnasaziMV.cpp:114-197;
 * all-reduced */
int mxg_mv_trans_mv(double alpha_re, double alpha_im, const mxg_mv* A, const mxg_mv* self,
                    double* B, int ldb);
/* B((m + m2) x k) = alpha * [A A2]^H * this in ONE pass over the rows (A2 may be NULL).  The block Krylov-Schur step
 * of Anasazi (called from src/MxSolver.cpp:100-104) needs V^H w (MvTransMv) and w^H w (MvNorm / MvDot) of the same
 * w back to back; with A2 = w both come from one launch.  All-reduced. */
int mxg_mv_trans_mv2(double alpha_re, double alpha_im, const mxg_mv* A, const mxg_mv* A2, const mxg_mv* self,
                     double* B, int ldb);

/* ------------------------------------------------------------------------------------------
 * inner Krylov solves  (replaces AztecOO::recursiveIterate + TrueResidual loop,
 *                       src/MxMagWaveOp.cpp:874-885,907-913; options :326-353)
 * ---------------------------------------------------------------------------------------- */
/* Multigrid preconditioner (K11): stands in for ML's smoothed-aggregation AMG
 * (src/MxMagWaveOp.cpp:425-449,524-531) with SpMV-only ingredients.  The caller supplies the
 * hierarchy (level operators A[0..nlevels), prolongations P[l]: n_l x n_{l+1}, restrictions
 * R[l]: n_{l+1} x n_l); one application is a V(pre, post) cycle with damped-Jacobi smoothing,
 * `coarse_sweeps` Jacobi sweeps on the coarsest level and the coarse correction scaled by
 * `coarse_scale` (1.8-2 compensates the under-correction of piecewise-constant aggregation).
 * The cycle is a fixed symmetric positive operator, so it is a legal CG preconditioner. */
typedef struct mxg_mg mxg_mg;
int mxg_mg_create(mxg_ctx* ctx, int nlevels, mxg_csr* const* A, mxg_csr* const* P, mxg_csr* const* R,
                  int pre_sweeps, int post_sweeps, int coarse_sweeps, double omega, double coarse_scale,
                  mxg_mg** out);
int mxg_mg_destroy(mxg_mg* mg);
/* Chebyshev instead of damped-Jacobi smoothing on every level but the coarsest (ML's default smoother,
 * src/MxProblem.cpp:56 "smoother" = "Chebyshev"): `sweeps` steps of the Chebyshev iteration for D^-1 A on
 * [lmax / ratio, lmax], lmax estimated per level by power iteration here (ML's `smoother: Chebyshev alpha`). */
int mxg_mg_set_chebyshev(mxg_mg* mg, double ratio);
/* z = M^-1 r (one column), one V-cycle */
int mxg_mg_apply(mxg_ctx* ctx, mxg_mg* mg, const mxg_mv* r, mxg_mv* z);
/* Decomposed grid: below the last row-partitioned level the hierarchy continues REPLICATED on every rank.
 * `tail` is a single-rank hierarchy whose level 0 is the last level of `mg` gathered over the ranks in rank order
 * (counts[r] = rows of rank r); per V-cycle one all-gather of that level's right-hand side replaces the ghost
 * imports of every sweep on every level below (they are latency-bound there).  `tail` stays owned by the caller. */
int mxg_mg_set_tail(mxg_mg* mg, mxg_mg* tail, const int64_t* counts);

#define MXG_SOLVE_ZERO_GUESS 1
typedef struct mxg_solve_opts {
  double tol;          /* unscaled true-residual 2-norm target (AZ_noscaled, :346,352) */
  int restart;         /* Krylov subspace (AZ_kspace, "linear solver : basis") */
  int max_iters;       /* total iteration cap; 0 = library default */
  int precond;         /* 0 none, 1 Jacobi (diagonal), 3 multigrid V-cycle (`mg`) */
  int flags;           /* bit 0 (MXG_SOLVE_ZERO_GUESS): x is zero on entry - the initial residual is b, no product with A */
  double lmin, lmax;   /* reserved */
  mxg_mg* mg;          /* hierarchy for precond == 3 (not owned) */
} mxg_solve_opts;

typedef struct mxg_solve_info {
  int iters;           /* iterations summed over all columns */
  int converged;       /* 1 if every column met tol */
  double true_res;     /* max over columns of ||b - A x||_2 at exit */
} mxg_solve_info;

/* solve A x = b column by column, x is the initial guess on entry.  CG needs A Hermitian
 * positive (semi-)definite on the span of b; zero rows of A must carry zero entries in b
 * (reference behaviour, SURVEY section 7 "structurally empty rows"). */
int mxg_solve_cg(mxg_ctx* ctx, const mxg_csr* A, const mxg_mv* b, mxg_mv* x,
                 const mxg_solve_opts* opts, mxg_solve_info* info);
int mxg_solve_gmres(mxg_ctx* ctx, const mxg_csr* A, const mxg_mv* b, mxg_mv* x,
                    const mxg_solve_opts* opts, mxg_solve_info* info);
/* AZ_bicgstab (src/MxMagWaveOp.cpp:333-334): right-preconditioned BiCGStab, same stopping rule */
int mxg_solve_bicgstab(mxg_ctx* ctx, const mxg_csr* A, const mxg_mv* b, mxg_mv* x,
                    const mxg_solve_opts* opts, mxg_solve_info* info);

#ifdef __cplusplus
}
#endif
#endif /* MXG_H */
