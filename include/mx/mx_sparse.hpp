// Host-side CSR container and the setup-time sparse algebra behind MxCrsMatrix
// (replaces Epetra_CrsMatrix assembly + EpetraExt::MatrixMatrix; reference
// src/MxCrsMatrix.cpp:84-116 purgeZeros, :130-170 insertRowValues, :325-342 fillComplete,
// :361-382 matrixMatrixMultiply, :404-457 matrixMatrixAdd, :461-466 save).
// These run once per problem on the host; the hot path only sees the finished arrays.
#ifndef MX_SPARSE_HPP
#define MX_SPARSE_HPP

#include <sys/mman.h>

#include <algorithm>
#include <cstdlib>
#include <new>
#include <memory>
#include <type_traits>
#include <utility>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace mx {

// Storage for the big setup arrays: a std::vector whose allocator (a) leaves new elements
// uninitialised, so resize() does not sweep the memory serially, and (b) takes blocks of 1 MiB
// and more straight from mmap with MAP_POPULATE, so the kernel maps the pages in one call
// instead of one soft page fault per 4 KiB (in sandboxed containers a fault costs ~100 us,
// i.e. 20+ s per GiB, which would dominate the whole assembly).
template <typename T>
class BigAllocator {
 public:
  typedef T value_type;
  BigAllocator() noexcept {}
  template <typename U> BigAllocator(const BigAllocator<U>&) noexcept {}
  template <typename U> struct rebind { typedef BigAllocator<U> other; };
  static constexpr size_t kMmapBytes = 1u << 20;
  static constexpr size_t kHugeBytes = (size_t)64 << 20;
  static bool hugePages() {
    static const bool on = !(std::getenv("MXH_HUGE_PAGES") && std::atoi(std::getenv("MXH_HUGE_PAGES")) == 0);   // experiments
    return on;
  }
  T* allocate(size_t n) {
    const size_t bytes = n * sizeof(T);
    if (bytes >= kHugeBytes && hugePages()) {
      // the arrays of a 256^3 problem (0.1 - 2 GB each): 2 MiB pages where the system hands them out on request
      // (transparent_hugepage = madvise) - populate and unmap take half the time (0.037 + 0.055 against 0.06 + 0.10 s
      // per GB on the MI355X hosts) and the assembly's random lookups into lattice-sized tables stop missing the TLB
      void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (p == MAP_FAILED) throw std::bad_alloc();
      madvise(p, bytes, MADV_HUGEPAGE);
      if (madvise(p, bytes, 23 /* MADV_POPULATE_WRITE, Linux 5.14 */) != 0) {
        munmap(p, bytes);                       // older kernel: the plain populated mapping below
      } else {
        return static_cast<T*>(p);
      }
    }
    if (bytes >= kMmapBytes) {
      void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
      if (p == MAP_FAILED) throw std::bad_alloc();
      return static_cast<T*>(p);
    }
    void* p = std::malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return static_cast<T*>(p);
  }
  void deallocate(T* p, size_t n) noexcept {
    const size_t bytes = n * sizeof(T);
    if (bytes >= kMmapBytes) munmap(p, bytes); else std::free(p);
  }
  template <typename U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) {
    ::new (static_cast<void*>(p)) U;   // default-init: no zero fill
  }
  template <typename U, typename... Args> void construct(U* p, Args&&... args) {
    ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
  }
  template <typename U> bool operator==(const BigAllocator<U>&) const noexcept { return true; }
  template <typename U> bool operator!=(const BigAllocator<U>&) const noexcept { return false; }
};
template <typename T> using UVec = std::vector<T, BigAllocator<T>>;

template <typename S>
struct Csr {
  int64_t nrows = 0, ncols = 0;
  UVec<int64_t> rowptr;   // nrows + 1
  UVec<int32_t> col;      // local column indices, ascending inside a row
  UVec<S> val;
  int64_t nnz() const { return rowptr.empty() ? 0 : rowptr.back(); }
};

// Build from unsorted triplets; duplicate (row, col) pairs are summed in insertion order,
// like Epetra's FillComplete does for repeated InsertGlobalValues.
template <typename S>
Csr<S> fromTriplets(int64_t nrows, int64_t ncols, const std::vector<int64_t>& r,
                    const std::vector<int32_t>& c, const std::vector<S>& v) {
  Csr<S> m;
  m.nrows = nrows; m.ncols = ncols;
  std::vector<int64_t> cnt(nrows + 1, 0);
  for (int64_t x : r) cnt[x + 1]++;
  for (int64_t i = 0; i < nrows; ++i) cnt[i + 1] += cnt[i];
  std::vector<int32_t> tc(c.size());
  std::vector<S> tv(v.size());
  std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
  for (size_t i = 0; i < r.size(); ++i) { int64_t p = pos[r[i]]++; tc[p] = c[i]; tv[p] = v[i]; }
  m.rowptr.assign(nrows + 1, 0);
  std::vector<int64_t> newcnt(nrows, 0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nrows; ++i) {
    const int64_t a = cnt[i], e = cnt[i + 1];
    // stable insertion sort by column (rows are short), then merge duplicates in place
    for (int64_t p = a + 1; p < e; ++p) {
      int32_t cc = tc[p]; S vv = tv[p];
      int64_t q = p - 1;
      while (q >= a && tc[q] > cc) { tc[q + 1] = tc[q]; tv[q + 1] = tv[q]; --q; }
      tc[q + 1] = cc; tv[q + 1] = vv;
    }
    int64_t w = a;
    for (int64_t p = a; p < e; ++p) {
      if (w > a && tc[w - 1] == tc[p]) tv[w - 1] += tv[p];
      else { tc[w] = tc[p]; tv[w] = tv[p]; ++w; }
    }
    newcnt[i] = w - a;
  }
  for (int64_t i = 0; i < nrows; ++i) m.rowptr[i + 1] = m.rowptr[i] + newcnt[i];
  m.col.resize(m.rowptr[nrows]);
  m.val.resize(m.rowptr[nrows]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nrows; ++i) {
    std::copy(tc.begin() + cnt[i], tc.begin() + cnt[i] + newcnt[i], m.col.begin() + m.rowptr[i]);
    std::copy(tv.begin() + cnt[i], tv.begin() + cnt[i] + newcnt[i], m.val.begin() + m.rowptr[i]);
  }
  return m;
}

template <typename S, typename V>
Csr<S> diagonal(const V& d) {
  Csr<S> m;
  const int64_t n = (int64_t)d.size();
  m.nrows = m.ncols = n;
  m.rowptr.resize(n + 1);
  m.col.resize(n);
  m.val.resize(n);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) { m.rowptr[i] = i; m.col[i] = (int32_t)i; m.val[i] = d[i]; }
  m.rowptr[n] = n;
  return m;
}

inline double conjScalar(double v) { return v; }
inline std::complex<double> conjScalar(std::complex<double> v) { return std::conj(v); }

template <typename S>
Csr<S> transpose(const Csr<S>& a, bool conj = false) {
  Csr<S> t;
  t.nrows = a.ncols; t.ncols = a.nrows;
  t.rowptr.assign(t.nrows + 1, 0);
  for (int64_t q = 0; q < a.nnz(); ++q) t.rowptr[a.col[q] + 1]++;
  for (int64_t i = 0; i < t.nrows; ++i) t.rowptr[i + 1] += t.rowptr[i];
  t.col.resize(a.nnz()); t.val.resize(a.nnz());
  std::vector<int64_t> pos(t.rowptr.begin(), t.rowptr.end() - 1);
  for (int64_t i = 0; i < a.nrows; ++i)
    for (int64_t q = a.rowptr[i]; q < a.rowptr[i + 1]; ++q) {
      int64_t p = pos[a.col[q]]++;
      t.col[p] = (int32_t)i;
      t.val[p] = conj ? conjScalar(a.val[q]) : a.val[q];
    }
  return t;
}

template <typename S>
Csr<S> cloneParallel(const Csr<S>& A) {
  Csr<S> C;
  C.nrows = A.nrows; C.ncols = A.ncols;
  C.rowptr.resize(A.rowptr.size());
  C.col.resize(A.col.size());
  C.val.resize(A.val.size());
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i <= A.nrows; ++i) C.rowptr[i] = A.rowptr[i];
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < (int64_t)A.col.size(); ++p) { C.col[p] = A.col[p]; C.val[p] = A.val[p]; }
  return C;
}

template <typename S>
bool isDiagonal(const Csr<S>& A) {
  if (A.nrows != A.ncols || A.nnz() != A.nrows) return false;
  for (int64_t i = 0; i < A.nrows; ++i)
    if (A.rowptr[i] != i || A.col[i] != (int32_t)i) return false;
  return true;
}

// C = A * B.  Diagonal factors (the Dey-Mittra fraction matrices) are applied as row / column
// scalings -- the same single products a general SpGEMM would form, on the same pattern.
// General case: Gustavson per row; the partial products of one output entry are accumulated in
// the order they are met (the order scipy's csr_matmat uses), entries emitted by ascending column.
template <typename S>
Csr<S> multiply(const Csr<S>& A, const Csr<S>& B) {
  if (isDiagonal(A)) {
    Csr<S> C = cloneParallel(B);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < C.nrows; ++i)
      for (int64_t p = C.rowptr[i]; p < C.rowptr[i + 1]; ++p) C.val[p] = A.val[i] * B.val[p];
    return C;
  }
  if (isDiagonal(B)) {
    Csr<S> C = cloneParallel(A);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < C.nrows; ++i)
      for (int64_t p = C.rowptr[i]; p < C.rowptr[i + 1]; ++p) C.val[p] = A.val[p] * B.val[A.col[p]];
    return C;
  }
  Csr<S> C;
  C.nrows = A.nrows; C.ncols = B.ncols;
  // upper bound of the entries of every row -> private slice per row, no allocation in the loop
  UVec<int64_t> ub(A.nrows + 1); ub[0] = 0;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.nrows; ++i) {
    int64_t n = 0;
    for (int64_t p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) n += B.rowptr[A.col[p] + 1] - B.rowptr[A.col[p]];
    ub[i + 1] = n;
  }
  for (int64_t i = 0; i < A.nrows; ++i) ub[i + 1] += ub[i];
  UVec<int32_t> tc(ub[A.nrows]);
  UVec<S> tv(ub[A.nrows]);
  UVec<int64_t> cnt(A.nrows);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.nrows; ++i) {
    int32_t* c = tc.data() + ub[i];
    S* v = tv.data() + ub[i];
    int64_t n = 0;
    for (int64_t p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) {
      const int32_t k = A.col[p];
      const S a = A.val[p];
      for (int64_t q = B.rowptr[k]; q < B.rowptr[k + 1]; ++q) {
        const int32_t j = B.col[q];
        const S prod = a * B.val[q];
        int64_t f = 0;
        while (f < n && c[f] != j) ++f;
        if (f < n) v[f] += prod; else { c[n] = j; v[n] = prod; ++n; }
      }
    }
    for (int64_t x = 1; x < n; ++x) {   // insertion sort by column
      const int32_t cc = c[x]; const S vv = v[x];
      int64_t y = x - 1;
      while (y >= 0 && c[y] > cc) { c[y + 1] = c[y]; v[y + 1] = v[y]; --y; }
      c[y + 1] = cc; v[y + 1] = vv;
    }
    cnt[i] = n;
  }
  C.rowptr.resize(A.nrows + 1); C.rowptr[0] = 0;
  for (int64_t i = 0; i < A.nrows; ++i) C.rowptr[i + 1] = C.rowptr[i] + cnt[i];
  C.col.resize(C.rowptr[A.nrows]);
  C.val.resize(C.rowptr[A.nrows]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.nrows; ++i) {
    std::copy(tc.begin() + ub[i], tc.begin() + ub[i] + cnt[i], C.col.begin() + C.rowptr[i]);
    std::copy(tv.begin() + ub[i], tv.begin() + ub[i] + cnt[i], C.val.begin() + C.rowptr[i]);
  }
  return C;
}

// C = sa * A + sb * B (union pattern; matrixMatrixAdd, src/MxCrsMatrix.cpp:404-457)
template <typename S>
Csr<S> add(const Csr<S>& A, S sa, const Csr<S>& B, S sb) {
  Csr<S> C;
  C.nrows = A.nrows; C.ncols = A.ncols;
  C.rowptr.assign(A.nrows + 1, 0);
  UVec<int64_t> cnt(A.nrows);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.nrows; ++i) {
    int64_t p = A.rowptr[i], pe = A.rowptr[i + 1], q = B.rowptr[i], qe = B.rowptr[i + 1], n = 0;
    while (p < pe || q < qe) {
      if (q >= qe || (p < pe && A.col[p] < B.col[q])) ++p;
      else if (p >= pe || B.col[q] < A.col[p]) ++q;
      else { ++p; ++q; }
      ++n;
    }
    cnt[i] = n;
  }
  for (int64_t i = 0; i < A.nrows; ++i) C.rowptr[i + 1] = C.rowptr[i] + cnt[i];
  C.col.resize(C.rowptr[A.nrows]);
  C.val.resize(C.rowptr[A.nrows]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.nrows; ++i) {
    int64_t p = A.rowptr[i], pe = A.rowptr[i + 1], q = B.rowptr[i], qe = B.rowptr[i + 1];
    int64_t w = C.rowptr[i];
    while (p < pe || q < qe) {
      if (q >= qe || (p < pe && A.col[p] < B.col[q])) { C.col[w] = A.col[p]; C.val[w] = sa * A.val[p]; ++p; }
      else if (p >= pe || B.col[q] < A.col[p]) { C.col[w] = B.col[q]; C.val[w] = sb * B.val[q]; ++q; }
      else { C.col[w] = A.col[p]; C.val[w] = sa * A.val[p] + sb * B.val[q]; ++p; ++q; }
      ++w;
    }
  }
  return C;
}

// keep |a| > tol (purgeZeros, src/MxCrsMatrix.cpp:84-116, tol 1e-12)
template <typename S>
Csr<S> purge(const Csr<S>& A, double tol = 1.e-12) {
  Csr<S> C;
  C.nrows = A.nrows; C.ncols = A.ncols;
  C.rowptr.assign(A.nrows + 1, 0);
  for (int64_t i = 0; i < A.nrows; ++i) {
    int64_t n = 0;
    for (int64_t p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) if (std::abs(A.val[p]) > tol) ++n;
    C.rowptr[i + 1] = C.rowptr[i] + n;
  }
  C.col.resize(C.rowptr[A.nrows]);
  C.val.resize(C.rowptr[A.nrows]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.nrows; ++i) {
    int64_t w = C.rowptr[i];
    for (int64_t p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p)
      if (std::abs(A.val[p]) > tol) { C.col[w] = A.col[p]; C.val[w] = A.val[p]; ++w; }
  }
  return C;
}

template <typename S>
void scaleInPlace(Csr<S>& A, S s) {
  for (auto& v : A.val) v *= s;
}

// row i *= d[i]  (leftScale, src/MxCrsMatrix.cpp:272-289)
template <typename S>
void leftScaleInPlace(Csr<S>& A, const std::vector<S>& d) {
  for (int64_t i = 0; i < A.nrows; ++i)
    for (int64_t p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) A.val[p] *= d[i];
}

}  // namespace mx
#endif
