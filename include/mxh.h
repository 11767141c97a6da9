/*
 * mxh.h -- C ABI of the host-side library (libmxhost.so): problem setup (Yee / Dey-Mittra
 * operator assembly, the MxMagWaveOp matrices) and the eigen-driver, written in C++ above the
 * device ABI of mxg.h.  This is what a Python / ctypes front end (tests, bench.py, the
 * example/*.py drivers) binds instead of shelling out to the `maxwell` binary
 * (reference example/pillbox.py:44-50, src/maxwell.cpp:64-119, src/MxProblem.cpp:111-201).
 *
 * All functions return 0 on success; mxh_last_error() gives the message otherwise.
 */
#ifndef MXH_H
#define MXH_H

#include <stddef.h>
#include <stdint.h>

#include "mxg.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxh_sim mxh_sim;
typedef struct mxh_solver mxh_solver;

const char* mxh_last_error(void);

/* Grid + boundary conditions + PEC shape  (MxGrid, MxEMSim::setup; src/MxEMSim.cpp:54-225).
 *   lower/upper: 3 strings out of "periodic" | "pec" | "pmc"  (src/MxProblem.cpp:441-470)
 *   pec_shape:   NULL/"" for none, else primitives joined by '&' (ShapeIntersection):
 *                  cylinder(r; ax,ay,az; x,y,z)   slab(t; nx,ny,nz; x,y,z)
 *                  sphere(r; x,y,z)               halfspace(nx,ny,nz; x,y,z)
 *                  ellipsoid(a,b,c; x,y,z)        torus(R; r; ax,ay,az; x,y,z)
 *                  cone(angle; ax,ay,az; x,y,z)   (vertex at x,y,z)
 *                unions, differences, mirrors, repeats and transformed shapes come in through a deck
 *                (mxh_problem_parse, all shape tags of src/MxProblem.cpp:8-17)
 *   strict:      1 = reference behaviour incl. the periodic-wrap quirk of
 *                MxYeeFitEField::getCompFactor, 0 = wrap-corrected */
int mxh_sim_create(const int n[3], const double origin[3], const double size[3],
                   const char* const lower[3], const char* const upper[3], const char* pec_shape,
                   int strict, mxh_sim** out);
/* mxh_sim_create_box plus materials: `dielectrics` = "e11,e22,e33[,o0,o1,o2] : shape-spec | ..." (real
 * symmetric permittivity tensors in the deck's `eps diag` / `eps off diag` order, src/MxProblem.cpp:487-527;
 * background vacuum), `dm_frac` = the <PEC> block's D-M frac.  With dielectrics the operator list gains
 * invEps (FIT inverse permittivity, src/MxYeeFitInvEps.cpp) and invEpsVolAve, and curlCurl / gradDiv
 * include them (src/MxMagWaveOp.cpp:140-179). */
int mxh_sim_create_ex(const int n[3], const double origin[3], const double size[3],
                      const char* const lower[3], const char* const upper[3], const char* pec_shape,
                      const char* dielectrics, double dm_frac, int strict, const int box_lo[3],
                      const int box_hi[3], mxh_sim** out);
int mxh_sim_destroy(mxh_sim* s);

/* ---------------------------------------------------------------------------------------
 * Domain decomposition: one GPU per MxGridDomain box (src/MxGrid.cpp:100-193)
 * ------------------------------------------------------------------------------------- */
/* MxGrid::partition + getGridDomain: process grid minimising sum_i p_i / n_i over all ordered
 * factorisations of nranks (first minimum wins), and the cell box [lo, hi) of `rank`
 * (the last rank along an axis also owns the upper node layer, hi = n + 1). */
int mxh_partition(const int n[3], int nranks, int rank, int proc_grid[3], int box_lo[3], int box_hi[3]);
/* rank owning global cell (i, j, k) under that partition */
int mxh_owner_of_cell(const int n[3], const int proc_grid[3], const int cell[3]);
/* like mxh_sim_create, for the sub-box [box_lo, box_hi) of this rank: operators are assembled on
 * the box plus a two-cell halo, so their owned rows are exact without communication */
int mxh_sim_create_box(const int n[3], const double origin[3], const double size[3],
                       const char* const lower[3], const char* const upper[3], const char* pec_shape,
                       int strict, const int box_lo[3], const int box_hi[3], mxh_sim** out);
/* The operator chain (curlCurl, gradDiv, vecLapl, scaLapl, mRhs: MxMagWaveOp::initMatrices, src/MxMagWaveOp.cpp:138-245)
 * of a sim that so far holds only its single operators, formed on the GPU of `ctx` (mxg_csr_multiply / add / purge,
 * bit-identical to the host chain) and read back.  mxh_solver_create calls it for the box + halo window of a rank of a
 * decomposed grid; a no-op once the chain exists.  Real problems. */
int mxh_sim_device_chain(mxg_ctx* ctx, mxh_sim* s);
/* owned rows of a named operator (or of vecLapl - shift*mRhs when name == "shifted") with GLOBAL
 * column ids: the row-partitioned form Epetra would hold.  row_gids / col_field_owned_gids are the
 * ascending global ids of the owned row entries and of the owned entries of the column field. */
int mxh_sim_dist_op(mxh_sim* s, const char* name, double shift, int64_t* nrows, int64_t* nnz,
                    const int64_t** rowptr, const int64_t** col_gids, const double** vals,
                    const int64_t** row_gids, int64_t* n_owned_cols, const int64_t** col_field_owned_gids,
                    int* col_ncomp);

/* The same through the C++ classes (MxEMSim maps of the owned entries, MxCrsMatrix with the ghost import agreed
 * between the ranks with mxg_comm_* collectives - what Epetra's FillComplete does, src/MxCrsMatrix.cpp:325-342).
 * `sim` is this rank's box simulation (mxh_sim_create_box with the box of mxh_partition); every call is COLLECTIVE
 * over the ranks of `ctx`.  mxh_dist_ops_matrix returns the device matrix (owned by the handle) of a named operator
 * or of vecLapl - shift*mRhs ("shifted"), its owned row ids and the owned ids of its column field (ascending). */
typedef struct mxh_dist_ops mxh_dist_ops;
int mxh_dist_ops_create(mxg_ctx* ctx, mxh_sim* sim, mxh_dist_ops** out);
int mxh_dist_ops_matrix(mxh_dist_ops* d, const char* name, double shift, mxg_csr** A, int64_t* nrows,
                        const int64_t** row_gids, int64_t* ncols_owned, const int64_t** col_gids, int64_t* nghost);
int mxh_dist_ops_destroy(mxh_dist_ops* d);

/* Borrowed views (valid until mxh_sim_destroy) of a named operator in local CSR form:
 * curlE curlB divB gradPsi dmA dmL dmVInv curlCurl gradDiv vecLapl scaLapl mRhs
 * (MxEMOps::getOp, src/MxEMOps.cpp:171-183; MxMagWaveOp::initMatrices, src/MxMagWaveOp.cpp:138-245) */
int mxh_sim_op(mxh_sim* s, const char* name, int64_t* nrows, int64_t* ncols, int64_t* nnz,
               const int64_t** rowptr, const int32_t** colidx, const double** vals);
/* vecLapl - shift * mRhs (MxMagWaveOp::setShift, src/MxMagWaveOp.cpp:248-255); cached per shift */
int mxh_sim_shifted(mxh_sim* s, double shift, int64_t* nrows, int64_t* nnz, const int64_t** rowptr,
                    const int32_t** colidx, const double** vals);
/* global ids of a field map: "bfield" | "efield" | "psifield" (MxGridField::getMap) */
int mxh_sim_map(mxh_sim* s, const char* field, int64_t* n, const int64_t** gids);
/* 0/1 per B-map row: 0 where the component is unused (zero area fraction / zero BC factor);
 * what MxGridField::zeroUnusedComponents clears (src/MxSolver.cpp:65) */
int mxh_sim_used_mask(mxh_sim* s, int64_t* n, const double** mask);
/* seconds spent in assembly (fractions + primitive operators + products) */
double mxh_sim_setup_seconds(const mxh_sim* s);

/* ---------------------------------------------------------------------------------------
 * Multigrid preconditioner setup (K11): geometric aggregation on the Yee lattice, Galerkin coarse
 * operators on the host, V-cycle on the GPU (mxg_mg_*).  Replaces the ML hierarchy of
 * src/MxMagWaveOp.cpp:425-449,524-531 ("linear solver : prec type" = amg).
 * ------------------------------------------------------------------------------------- */
typedef struct mxh_mg mxh_mg;
/* op: "shifted" (vecLapl - shift*mRhs, B field) or "scaLapl" (Psi field) */
int mxh_mg_create(mxg_ctx* ctx, mxh_sim* sim, const char* op, double shift, int64_t min_coarse,
                  int max_levels, int sweeps, int coarse_sweeps, double omega, double coarse_scale,
                  mxh_mg** out);
int mxh_mg_destroy(mxh_mg* mg);
mxg_mg* mxh_mg_handle(mxh_mg* mg);          /* the device V-cycle (for mxg_solve_opts.mg) */
mxg_csr* mxh_mg_fine_matrix(mxh_mg* mg);    /* level-0 operator on the device */
/* a matrix of the hierarchy, borrowed (tests): which = 'A' the operator of level l, 'P' the prolongation from level l + 1
 * to level l, 'R' the restriction from level l to level l + 1; NULL when there is none */
mxg_csr* mxh_mg_level_matrix(mxh_mg* mg, int level, char which);
int mxh_mg_level_sizes(mxh_mg* mg, int64_t* rows, int64_t* nnz, int* nlevels);   /* arrays of >= 16 */

/* ---------------------------------------------------------------------------------------
 * Eigen-driver: MxSolver + MxMagWaveOp on one GPU context
 * (src/MxSolver.cpp:24-94 ctor, :100-239 solve; src/MxMagWaveOp.cpp:826-943 Apply)
 * ------------------------------------------------------------------------------------- */
typedef struct mxh_solver_params {
  int nev;               /* "eigensolver : nev"            (default 10, src/MxProblem.cpp:31) */
  int basis;             /* "eigensolver : basis"          (default 20) = Anasazi Num Blocks */
  int block_size;        /* "eigensolver : block size"     (default 1) */
  int max_restarts;      /* "eigensolver : max restarts"   (default 100) */
  double tol;            /* "eigensolver : tol"            (default 1e-6) */
  double shift;          /* "eigensolver : shift" */
  const char* lin_type;  /* "linear solver : type": "gmres" | "cg"  (default gmres) */
  int lin_basis;         /* "linear solver : basis"        (default 40) */
  double lin_tol;        /* "linear solver : tol"          (default 1e-6) */
  const char* prec_type; /* "linear solver : prec type": "none" | "jacobi" | "ilu" (-> Jacobi) | "amg" (multigrid V-cycle) */
  int verbose;
  uint64_t seed;
  const char* amg_smoother; /* "linear solver : amg : smoother": "Chebyshev" (default, src/MxProblem.cpp:56) | "Jacobi" */
  int amg_sweeps;           /* "linear solver : amg : sweeps" (default 2) */
  const char* eig_type;     /* "eigensolver : type": "krylov-schur" (default) | "davidson" (src/MxSolver.cpp:85-94) */
} mxh_solver_params;

void mxh_solver_default_params(mxh_solver_params* p);
int mxh_solver_create(mxg_ctx* ctx, mxh_sim* sim, const mxh_solver_params* p, mxh_solver** out);
int mxh_solver_destroy(mxh_solver* s);
/* runs MxSolver::solve(); returns the number of converged eigenpairs in *nconv */
int mxh_solver_solve(mxh_solver* s, int* nconv);
/* per converged pair i: Anasazi eigenvalue theta (re, im), k^2 = 1/theta + shift, frequency in Hz
 * (eigValsToFreqs), relative residual (checkEigensolution) and divergence norm (checkDivergences) */
int mxh_solver_result(mxh_solver* s, int i, double theta[2], double k2[2], double freq[2],
                      double* residual, double* divergence);
/* counters of src/MxMagWaveOp.cpp:97-115 plus the eigensolve wall time of src/MxSolver.cpp:102-113 */
int mxh_solver_stats(mxh_solver* s, int* num_applies, int* vec_lin_iters, int* sca_lin_iters,
                     double* vec_solve_seconds, double* sca_solve_seconds, double* eigensolve_seconds);
/* copy eigenvector i of the B field (local rows) to host */
int mxh_solver_eigvec(mxh_solver* s, int i, double* out);

/* ---------------------------------------------------------------------------------------
 * `.mx` input deck front end: MxProblem (src/MxProblem.cpp:201-706) behind the `maxwell` binary
 * (src/maxwell.cpp:41-120, `maxwell --infile=pillbox.mx`).  Parsing needs no GPU.
 * ------------------------------------------------------------------------------------- */
typedef struct mxh_problem mxh_problem;
/* parse deck text / a deck file; `strict` as in mxh_sim_create */
int mxh_problem_parse(const char* mx_text, int strict, mxh_problem** out);
int mxh_problem_load(const char* path, int strict, mxh_problem** out);
int mxh_problem_destroy(mxh_problem* p);
/* value of a parameter-list entry as text ("eigensolver : nev", "linear solver : prec type",
 * "linear solver : ilu : drop tol", "polarization", ...; the keys of src/MxProblem.cpp:82-105).
 * Returns 1 if the key is not set. */
int mxh_problem_param(mxh_problem* p, const char* key, char* buf, size_t buflen);
/* <Grid>, <BoundaryConditions> (0 periodic, 1 pec, 2 pmc per side) and the <PEC> D-M frac */
int mxh_problem_grid(mxh_problem* p, int n[3], double origin[3], double size[3], int lower[3],
                     int upper[3], double* dm_frac, int* has_pec);
/* level-set value of the <PEC> shape at a point (MxShape::func, > 0 inside) */
int mxh_problem_pec_func(mxh_problem* p, const double xyz[3], double* f);
/* 1 if the deck needs complex arithmetic (non-zero Bloch phase shifts, src/MxProblem.cpp:640-642) */
int mxh_problem_is_complex(mxh_problem* p);
/* a named operator of the deck (as mxh_sim_op); for a complex deck `vals` are interleaved (re, im) pairs */
int mxh_problem_op(mxh_problem* p, const char* name, int64_t* nrows, int64_t* ncols, int64_t* nnz,
                   const int64_t** rowptr, const int32_t** colidx, const double** vals, int* is_complex);
/* assemble (once) and return the simulation; owned by the problem (real decks only) */
int mxh_problem_sim(mxh_problem* p, mxh_sim** sim);
/* MxSolver with the deck's parameters on `ctx`; owned by the caller (mxh_solver_destroy) and
 * valid while the problem lives.  `verbose` < 0 keeps the deck's "output" level. */
int mxh_problem_solver(mxg_ctx* ctx, mxh_problem* p, int verbose, mxh_solver** out);

/* ---------------------------------------------------------------------------------------
 * Field output in the layout of MxIO::save (src/MxIO.cpp:94-222): a 4-d array of doubles
 * (Nx+1, Ny+1, Nz+1, ncomp) per vector.  `<stem>.h5` with the named dataset when an HDF5 library
 * can be loaded at run time (it is not a build dependency), otherwise `<stem>.npy`.
 * ------------------------------------------------------------------------------------- */
int mxh_hdf5_available(void);
int mxh_field_write(const char* stem, const char* dataset, const int64_t dims[4], const double* data,
                    char* written, size_t written_len);
/* a further dataset in the same file ("imag" next to "real", src/MxIO.cpp:199-222); `<stem>_<dataset>.npy` without HDF5 */
int mxh_field_append(const char* stem, const char* dataset, const int64_t dims[4], const double* data,
                     char* written, size_t written_len);
/* read a dataset back (dims always; data when non-NULL and capacity >= product of dims) */
int mxh_field_read_h5(const char* path, const char* dataset, int64_t dims[4], double* data, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* MXH_H */
