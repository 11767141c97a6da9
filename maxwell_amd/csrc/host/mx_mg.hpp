// Geometric-aggregation multigrid hierarchy for the Yee-lattice operators (host, setup).
// Stands in for ML's smoothed-aggregation setup (reference src/MxMagWaveOp.cpp:425-449,524-531):
// aggregates are the 2x2x2 cell blocks of the lattice, per field component; P is piecewise
// constant, R = P^T, A_c = R A P (Galerkin).  The V-cycle itself runs on the GPU (mxg_mg_*).
#ifndef MX_MG_HPP
#define MX_MG_HPP

#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

#include "mx/mx_sparse.hpp"

namespace mx {

struct MgHierarchy {
  const Csr<double>* fine = nullptr;   // level 0 is the caller's matrix (not copied: 2.2 GB at 256^3); A[0] stays empty
  std::vector<Csr<double>> A;   // level operators, fine first
  const Csr<double>& level(size_t l) const { return l == 0 ? *fine : A[l]; }
  std::vector<Csr<double>> P;   // P[l]: n_l x n_{l+1}
  std::vector<Csr<double>> R;   // R[l]: n_{l+1} x n_l
};

inline MgHierarchy buildAggregationHierarchy(const Csr<double>& A0, const UVec<int64_t>& gids, int ncomp,
                                             const int n[3], int64_t minCoarse, int maxLevels) {
  MgHierarchy h;
  h.fine = &A0;
  h.A.emplace_back();
  // lattice coordinates of every fine unknown
  int64_t dims[3] = {n[0] + 1, n[1] + 1, n[2] + 1};
  const int64_t nf0 = (int64_t)gids.size();
  UVec<int32_t> comp(nf0), ci(nf0), cj(nf0), ck(nf0);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nf0; ++r) {
    const int64_t cell = gids[r] / ncomp;
    comp[r] = (int32_t)(gids[r] % ncomp);
    ck[r] = (int32_t)(cell % dims[2]);
    cj[r] = (int32_t)((cell / dims[2]) % dims[1]);
    ci[r] = (int32_t)(cell / (dims[2] * dims[1]));
  }
  while ((int)h.A.size() < maxLevels && h.level(h.A.size() - 1).nrows > minCoarse) {
    const Csr<double>& A = h.level(h.A.size() - 1);
    const int64_t nf = A.nrows;
    const int64_t cd[3] = {(dims[0] + 1) / 2, (dims[1] + 1) / 2, (dims[2] + 1) / 2};
    // unknowns with a zero diagonal (structurally empty rows: unused components that stay in the
    // map, reference src/MxYeeDeyMittraCurlE.cpp:148-150) take no part: they get no correction,
    // so they stay exactly zero like in the reference
    UVec<uint8_t> active(nf);
    UVec<int64_t> key(nf);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nf; ++i) {
      uint8_t act = 0;
      for (int64_t q = A.rowptr[i]; q < A.rowptr[i + 1]; ++q)
        if (A.col[q] == i && A.val[q] != 0.0) act = 1;
      active[i] = act;
      key[i] = comp[i] + (int64_t)ncomp * ((((int64_t)ci[i] >> 1) * cd[1] + (cj[i] >> 1)) * cd[2] + (ck[i] >> 1));
    }
    // aggregates numbered in ascending key order: flag the keys in use over the dense coarse lattice, prefix sum
    const int64_t nkeys = (int64_t)ncomp * cd[0] * cd[1] * cd[2];
    UVec<int32_t> number(nkeys + 1);
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k <= nkeys; ++k) number[k] = 0;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nf; ++r) if (active[r]) number[key[r]] = 1;   // benign race: every writer stores 1
    int64_t nc = 0;
    for (int64_t k = 0; k < nkeys; ++k) { const int32_t f = number[k]; number[k] = (int32_t)nc; nc += f; }
    if (nc == 0 || nc >= nf) break;
    UVec<int32_t> agg(nf);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nf; ++r) agg[r] = active[r] ? number[key[r]] : -1;
    // P: one unit entry per active fine row
    Csr<double> P;
    P.nrows = nf; P.ncols = nc;
    P.rowptr.resize(nf + 1);
    P.rowptr[0] = 0;
    for (int64_t r = 0; r < nf; ++r) P.rowptr[r + 1] = P.rowptr[r] + (active[r] ? 1 : 0);
    P.col.resize(P.rowptr[nf]); P.val.resize(P.rowptr[nf]);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nf; ++r) if (active[r]) { P.col[P.rowptr[r]] = agg[r]; P.val[P.rowptr[r]] = 1.0; }
    // R = P^T: the fine rows of every aggregate in ascending order (counting sort)
    Csr<double> R;
    R.nrows = nc; R.ncols = nf;
    R.rowptr.resize(nc + 1);
    for (int64_t a = 0; a <= nc; ++a) R.rowptr[a] = 0;
    for (int64_t r = 0; r < nf; ++r) if (agg[r] >= 0) R.rowptr[agg[r] + 1]++;
    for (int64_t a = 0; a < nc; ++a) R.rowptr[a + 1] += R.rowptr[a];
    R.col.resize(R.rowptr[nc]); R.val.resize(R.rowptr[nc]);
    {
      UVec<int64_t> pos(R.rowptr.begin(), R.rowptr.end() - 1);
      for (int64_t r = 0; r < nf; ++r)
        if (agg[r] >= 0) { const int64_t q = pos[agg[r]]++; R.col[q] = (int32_t)r; R.val[q] = 1.0; }
    }
    // Galerkin coarse operator R A P, one coarse row at a time: the entries of the aggregate's fine rows (fine-row
    // order, then storage order) are relabelled and merged into a list sorted by coarse column, repeated columns
    // summed in arrival order (= a stable sort by column followed by a left-to-right sum).  One pass: contiguous
    // chunks of coarse rows go to per-chunk buffers, the row pointer is their prefix sum, the buffers are copied out.
    Csr<double> Ac;
    Ac.nrows = nc; Ac.ncols = nc;
    Ac.rowptr.resize(nc + 1);
    UVec<int64_t> cnt(nc);
    const int64_t nchunk = std::max<int64_t>(1, std::min<int64_t>(nc, 256));
    std::vector<std::vector<int32_t> > ccol((size_t)nchunk);
    std::vector<std::vector<double> > cval((size_t)nchunk);
    auto chunkStart = [&](int64_t ch) { return ch * nc / nchunk; };
#pragma omp parallel
    {
      std::vector<int32_t> kc;
      std::vector<double> kv;
#pragma omp for schedule(dynamic, 1)
      for (int64_t ch = 0; ch < nchunk; ++ch) {
        std::vector<int32_t>& oc = ccol[(size_t)ch];
        std::vector<double>& ov = cval[(size_t)ch];
        const int64_t a0 = chunkStart(ch), a1 = chunkStart(ch + 1);
        oc.reserve((size_t)(a1 - a0) * 8); ov.reserve((size_t)(a1 - a0) * 8);
        for (int64_t a = a0; a < a1; ++a) {
          kc.clear(); kv.clear();
          for (int64_t t = R.rowptr[a]; t < R.rowptr[a + 1]; ++t) {
            const int64_t i = R.col[t];
            for (int64_t q = A.rowptr[i]; q < A.rowptr[i + 1]; ++q) {
              const int32_t c = agg[A.col[q]];
              if (c < 0) continue;
              size_t pos = kc.size();                            // rows are short: linear search from the back
              while (pos > 0 && kc[pos - 1] > c) --pos;
              if (pos > 0 && kc[pos - 1] == c) kv[pos - 1] += A.val[q];
              else { kc.insert(kc.begin() + (long)pos, c); kv.insert(kv.begin() + (long)pos, A.val[q]); }
            }
          }
          cnt[a] = (int64_t)kc.size();
          oc.insert(oc.end(), kc.begin(), kc.end());
          ov.insert(ov.end(), kv.begin(), kv.end());
        }
      }
    }
    Ac.rowptr[0] = 0;
    for (int64_t a = 0; a < nc; ++a) Ac.rowptr[a + 1] = Ac.rowptr[a] + cnt[a];
    Ac.col.resize(Ac.rowptr[nc]); Ac.val.resize(Ac.rowptr[nc]);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t ch = 0; ch < nchunk; ++ch) {
      const int64_t o = Ac.rowptr[chunkStart(ch)];
      std::copy(ccol[(size_t)ch].begin(), ccol[(size_t)ch].end(), Ac.col.begin() + o);
      std::copy(cval[(size_t)ch].begin(), cval[(size_t)ch].end(), Ac.val.begin() + o);
      std::vector<int32_t>().swap(ccol[(size_t)ch]);
      std::vector<double>().swap(cval[(size_t)ch]);
    }
    // coordinates of the coarse unknowns
    UVec<int32_t> ncomp_c(nc), ni(nc), nj(nc), nk(nc);
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < nc; ++a) {
      const int64_t r = R.col[R.rowptr[a]];
      ncomp_c[a] = comp[r]; ni[a] = ci[r] >> 1; nj[a] = cj[r] >> 1; nk[a] = ck[r] >> 1;
    }
    comp.swap(ncomp_c); ci.swap(ni); cj.swap(nj); ck.swap(nk);
    dims[0] = cd[0]; dims[1] = cd[1]; dims[2] = cd[2];
    h.P.push_back(std::move(P)); h.R.push_back(std::move(R)); h.A.push_back(std::move(Ac));
  }
  return h;
}

}  // namespace mx

// host + device hierarchy in one object (defined in mxh_mg.cpp; C handle type of mxh.h)
struct mxh_mg;
struct mxg_ctx;
struct mxg_csr;
namespace mx { struct YeeField; }
mxh_mg* mxh_mg_build(mxg_ctx* ctx, const mx::Csr<double>& A0, const mx::YeeField& F, const int n[3],
                     int64_t min_coarse, int max_levels, int sweeps, int coarse_sweeps, double omega,
                     double coarse_scale, mxg_csr* A0dev = nullptr);
// same for any set of lattice unknowns given by their global ids (the owned block of one rank of a decomposed grid)
mxh_mg* mxh_mg_build_gids(mxg_ctx* ctx, const mx::Csr<double>& A0, const mx::UVec<int64_t>& gids, int ncomp, const int n[3],
                          int64_t min_coarse, int max_levels, int sweeps, int coarse_sweeps, double omega,
                          double coarse_scale, mxg_csr* A0dev = nullptr);
#endif
