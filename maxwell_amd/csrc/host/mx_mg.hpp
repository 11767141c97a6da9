// Geometric-aggregation multigrid hierarchy for the Yee-lattice operators (host, setup).
// Stands in for ML's smoothed-aggregation setup (reference src/MxMagWaveOp.cpp:425-449,524-531):
// aggregates are the 2x2x2 cell blocks of the lattice, per field component; P is piecewise
// constant, R = P^T, A_c = R A P (Galerkin).  The V-cycle itself runs on the GPU (mxg_mg_*).
#ifndef MX_MG_HPP
#define MX_MG_HPP

#include <algorithm>
#include <vector>

#include "mx/mx_sparse.hpp"

namespace mx {

struct MgHierarchy {
  std::vector<Csr<double>> A;   // level operators, fine first
  std::vector<Csr<double>> P;   // P[l]: n_l x n_{l+1}
  std::vector<Csr<double>> R;   // R[l]: n_{l+1} x n_l
};

inline MgHierarchy buildAggregationHierarchy(const Csr<double>& A0, const UVec<int64_t>& gids, int ncomp,
                                             const int n[3], int64_t minCoarse, int maxLevels) {
  MgHierarchy h;
  h.A.push_back(A0);
  // lattice coordinates of every fine unknown
  int64_t dims[3] = {n[0] + 1, n[1] + 1, n[2] + 1};
  const int64_t nf0 = (int64_t)gids.size();
  std::vector<int> comp(nf0);
  std::vector<int64_t> ci(nf0), cj(nf0), ck(nf0);
  for (int64_t r = 0; r < nf0; ++r) {
    const int64_t cell = gids[r] / ncomp;
    comp[r] = (int)(gids[r] % ncomp);
    ck[r] = cell % dims[2];
    cj[r] = (cell / dims[2]) % dims[1];
    ci[r] = cell / (dims[2] * dims[1]);
  }
  while ((int)h.A.size() < maxLevels && h.A.back().nrows > minCoarse) {
    const Csr<double>& A = h.A.back();
    const int64_t nf = A.nrows;
    int64_t cd[3] = {(dims[0] + 1) / 2, (dims[1] + 1) / 2, (dims[2] + 1) / 2};
    // unknowns with a zero diagonal (structurally empty rows: unused components that stay in the
    // map, reference src/MxYeeDeyMittraCurlE.cpp:148-150) take no part: they get no correction,
    // so they stay exactly zero like in the reference
    std::vector<uint8_t> active(nf, 0);
    for (int64_t i = 0; i < nf; ++i)
      for (int64_t q = A.rowptr[i]; q < A.rowptr[i + 1]; ++q)
        if (A.col[q] == i && A.val[q] != 0.0) active[i] = 1;
    std::vector<int64_t> key(nf);
    std::vector<int64_t> uniq;
    for (int64_t r = 0; r < nf; ++r) {
      key[r] = comp[r] + (int64_t)ncomp * (((ci[r] >> 1) * cd[1] + (cj[r] >> 1)) * cd[2] + (ck[r] >> 1));
      if (active[r]) uniq.push_back(key[r]);
    }
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    const int64_t nc = (int64_t)uniq.size();
    if (nc == 0 || nc >= nf) break;
    std::vector<int32_t> agg(nf, -1);
    for (int64_t r = 0; r < nf; ++r)
      if (active[r]) agg[r] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), key[r]) - uniq.begin());
    // P: one unit entry per active fine row
    Csr<double> P;
    P.nrows = nf; P.ncols = nc;
    P.rowptr.resize(nf + 1);
    P.rowptr[0] = 0;
    for (int64_t r = 0; r < nf; ++r) P.rowptr[r + 1] = P.rowptr[r] + (active[r] ? 1 : 0);
    P.col.resize(P.rowptr[nf]); P.val.resize(P.rowptr[nf]);
    for (int64_t r = 0; r < nf; ++r) if (active[r]) { P.col[P.rowptr[r]] = agg[r]; P.val[P.rowptr[r]] = 1.0; }
    Csr<double> R = transpose(P);
    // Galerkin coarse operator by relabelling the fine entries
    std::vector<int64_t> tr;
    std::vector<int32_t> tc;
    std::vector<double> tv;
    tr.reserve(A.nnz()); tc.reserve(A.nnz()); tv.reserve(A.nnz());
    for (int64_t i = 0; i < nf; ++i) {
      if (!active[i]) continue;
      for (int64_t q = A.rowptr[i]; q < A.rowptr[i + 1]; ++q) {
        if (!active[A.col[q]]) continue;
        tr.push_back(agg[i]); tc.push_back(agg[A.col[q]]); tv.push_back(A.val[q]);
      }
    }
    Csr<double> Ac = fromTriplets<double>(nc, nc, tr, tc, tv);
    // coordinates of the coarse unknowns
    std::vector<int> ncomp_c(nc);
    std::vector<int64_t> ni(nc), nj(nc), nk(nc);
    for (int64_t r = 0; r < nf; ++r) {
      const int32_t a = agg[r];
      if (a < 0) continue;
      ncomp_c[a] = comp[r]; ni[a] = ci[r] >> 1; nj[a] = cj[r] >> 1; nk[a] = ck[r] >> 1;
    }
    comp.swap(ncomp_c); ci.swap(ni); cj.swap(nj); ck.swap(nk);
    dims[0] = cd[0]; dims[1] = cd[1]; dims[2] = cd[2];
    h.P.push_back(P); h.R.push_back(R); h.A.push_back(Ac);
  }
  return h;
}

}  // namespace mx

// host + device hierarchy in one object (defined in mxh_mg.cpp; C handle type of mxh.h)
struct mxh_mg;
struct mxg_ctx;
namespace mx { struct YeeField; }
mxh_mg* mxh_mg_build(mxg_ctx* ctx, const mx::Csr<double>& A0, const mx::YeeField& F, const int n[3],
                     int64_t min_coarse, int max_levels, int sweeps, int coarse_sweeps, double omega,
                     double coarse_scale);
#endif
