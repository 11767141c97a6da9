/* Oracle (TEST INFRASTRUCTURE ONLY): plain-C restatement of the serial CSR row loop that
 * Epetra_CrsMatrix::Apply runs per MPI rank behind MxCrsMatrix<S>::apply
 * (reference src/MxCrsMatrix.cpp:347-353; Trilinos itself is not vendored: this is the
 * published textbook loop, sum += val[j] * x[col[j]] in storage order, one row after the other).
 * Compiled with -ffp-contract=off so a product is rounded before it is added, which is what
 * the GPU kernel reproduces bit for bit.  Also timed by bench.py as the CPU baseline
 * ("port"): oracle_spmv_omp splits the rows statically over OpenMP threads, the stand-in for
 * one MPI rank per core.
 */
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void oracle_spmv(int64_t nrows, const int64_t* rowptr, const int32_t* col, const double* val,
                 const double* x, double* y) {
  for (int64_t i = 0; i < nrows; ++i) {
    double s = 0.0;
    for (int64_t q = rowptr[i]; q < rowptr[i + 1]; ++q) s += val[q] * x[col[q]];
    y[i] = s;
  }
}

/* k right-hand sides, column-major with leading dimensions ldx / ldy (Epetra_MultiVector layout) */
void oracle_spmm(int64_t nrows, const int64_t* rowptr, const int32_t* col, const double* val,
                 int k, const double* x, int64_t ldx, double* y, int64_t ldy) {
  for (int j = 0; j < k; ++j) oracle_spmv(nrows, rowptr, col, val, x + j * ldx, y + j * ldy);
}

/* complex: values and vectors are interleaved (re, im) pairs */
void oracle_spmv_complex(int64_t nrows, const int64_t* rowptr, const int32_t* col,
                         const double* val, const double* x, double* y) {
  for (int64_t i = 0; i < nrows; ++i) {
    double sr = 0.0, si = 0.0;
    for (int64_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
      const double ar = val[2 * q], ai = val[2 * q + 1];
      const double xr = x[2 * (int64_t)col[q]], xi = x[2 * (int64_t)col[q] + 1];
      sr += ar * xr - ai * xi;
      si += ar * xi + ai * xr;
    }
    y[2 * i] = sr;
    y[2 * i + 1] = si;
  }
}

void oracle_spmv_omp(int64_t nrows, const int64_t* rowptr, const int32_t* col, const double* val,
                     const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nrows; ++i) {
    double s = 0.0;
    for (int64_t q = rowptr[i]; q < rowptr[i + 1]; ++q) s += val[q] * x[col[q]];
    y[i] = s;
  }
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
