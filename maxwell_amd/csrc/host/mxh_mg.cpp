// C ABI over the multigrid setup (include/mxh.h, mxh_mg_*).
#include <chrono>
#include <cstdio>
#include <memory>
#include <stdexcept>

#include "mx_mg.hpp"
#include "mxh.h"
#include "mxh_internal.hpp"

struct mxh_mg {
  mxg_ctx* ctx = nullptr;
  std::vector<mxg_csr*> A, P, R;
  std::vector<int64_t> rows, nnz;
  mxg_mg* mg = nullptr;
  bool ownsA0 = true;                  // level 0 may be the caller's device matrix (not copied, not destroyed here)
  ~mxh_mg() {
    if (mg) mxg_mg_destroy(mg);
    for (size_t l = ownsA0 ? 0 : 1; l < A.size(); ++l) mxg_csr_destroy(A[l]);
    for (auto* m : P) mxg_csr_destroy(m);
    for (auto* m : R) mxg_csr_destroy(m);
  }
};

// reencode: level operators take the tile form / compressed stream (their products carry the fused smoother
// epilogues); the transfer operators never use either - the piecewise-constant prolongation runs from its injection
// form, the restriction from its ELL form (mxg_csr.hip) - and planning them cost 0.9 s of the 256^3 setup
static mxg_csr* upload(mxg_ctx* ctx, const mx::Csr<double>& m, bool reencode) {
  mxg_csr* d = nullptr;
  if (mxg_csr_create(ctx, m.nrows, m.ncols, 0, m.rowptr.data(), m.col.data(), m.val.data(), 0, nullptr, &d) != 0)
    throw std::runtime_error(std::string("mxg_csr_create: ") + mxg_last_error());
  static const bool compress = !(getenv("MXG_COMPRESS") && atoi(getenv("MXG_COMPRESS")) == 0);
  if (reencode && compress && m.nrows >= 256 &&    // small levels too: their sweeps are launch-bound, the fused epilogues halve the launches
      mxg_csr_compress_from(d, m.rowptr.data(), m.col.data(), m.val.data(), nullptr, nullptr, nullptr) != 0)
    throw std::runtime_error(std::string("mxg_csr_compress: ") + mxg_last_error());
  return d;
}

mxh_mg* mxh_mg_build(mxg_ctx* ctx, const mx::Csr<double>& A0, const mx::YeeField& F, const int n[3],
                     int64_t min_coarse, int max_levels, int sweeps, int coarse_sweeps, double omega,
                     double coarse_scale, mxg_csr* A0dev) {
  return mxh_mg_build_gids(ctx, A0, F.gids, F.ncomp, n, min_coarse, max_levels, sweeps, coarse_sweeps, omega, coarse_scale, A0dev);
}

// A0dev: the caller's device copy of A0 (already re-encoded) serves as level 0 and stays the caller's: the 256^3
// operator is 2.2 GB and its upload + tile plan 0.3 s
mxh_mg* mxh_mg_build_gids(mxg_ctx* ctx, const mx::Csr<double>& A0, const mx::UVec<int64_t>& gids, int ncomp, const int n[3],
                          int64_t min_coarse, int max_levels, int sweeps, int coarse_sweeps, double omega,
                          double coarse_scale, mxg_csr* A0dev) {
  const bool timing = getenv("MXH_VERBOSE") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const auto t0 = now();
  mx::MgHierarchy h = mx::buildAggregationHierarchy(A0, gids, ncomp, n, min_coarse, max_levels);
  const auto t1 = now();
  std::unique_ptr<mxh_mg> g(new mxh_mg());
  g->ctx = ctx;
  g->ownsA0 = A0dev == nullptr;
  for (size_t l = 0; l < h.A.size(); ++l) {
    const mx::Csr<double>& m = h.level(l);
    g->A.push_back(l == 0 && A0dev ? A0dev : upload(ctx, m, true));
    g->rows.push_back(m.nrows); g->nnz.push_back(m.nnz());
  }
  for (auto& m : h.P) g->P.push_back(upload(ctx, m, false));
  for (auto& m : h.R) g->R.push_back(upload(ctx, m, false));
  const auto t2 = now();
  if (mxg_mg_create(ctx, (int)g->A.size(), g->A.data(), g->P.data(), g->R.data(), sweeps, sweeps, coarse_sweeps,
                    omega, coarse_scale, &g->mg) != 0)
    throw std::runtime_error(std::string("mxg_mg_create: ") + mxg_last_error());
  if (timing)
    fprintf(stderr, "[mxh_mg] %lld rows, %d levels: hierarchy (host) %.3f s, upload + product forms %.3f s, cycle workspace %.3f s\n",
            (long long)A0.nrows, (int)g->A.size(), secs(t0, t1), secs(t1, t2), secs(t2, now()));
  return g.release();
}

extern "C" int mxh_mg_create(mxg_ctx* ctx, mxh_sim* sim, const char* op, double shift, int64_t min_coarse,
                             int max_levels, int sweeps, int coarse_sweeps, double omega, double coarse_scale,
                             mxh_mg** out) {
  try {
    if (!ctx || !sim || !op || !out) throw std::runtime_error("mxh_mg_create: NULL argument");
    if (sim->asmb->isWindowed()) throw std::runtime_error("mxh_mg_create: single-rank simulations only");
    const std::string nm(op);
    const mx::Csr<double>* A0;
    const mx::YeeField* F;
    if (nm == "shifted") { A0 = &mxh_sim_shifted_matrix(sim, shift); F = &sim->asmb->bfield(); }
    else if (nm == "scaLapl") { mxh_sim_ensure_chain(sim); A0 = &sim->ops.at("scaLapl"); F = &sim->asmb->psifield(); }
    else throw std::runtime_error("mxh_mg_create: operator must be 'shifted' or 'scaLapl'");
    *out = mxh_mg_build(ctx, *A0, *F, sim->cfg.n, min_coarse, max_levels, sweeps, coarse_sweeps, omega, coarse_scale);
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}

extern "C" int mxh_mg_destroy(mxh_mg* mg) { delete mg; return 0; }
extern "C" mxg_mg* mxh_mg_handle(mxh_mg* mg) { return mg->mg; }
extern "C" mxg_csr* mxh_mg_fine_matrix(mxh_mg* mg) { return mg->A[0]; }
extern "C" mxg_csr* mxh_mg_level_matrix(mxh_mg* mg, int level, char which) {
  if (!mg || level < 0) return nullptr;
  const std::vector<mxg_csr*>& v = which == 'A' ? mg->A : (which == 'P' ? mg->P : mg->R);
  if (which != 'A' && which != 'P' && which != 'R') return nullptr;
  return (size_t)level < v.size() ? v[(size_t)level] : nullptr;
}
extern "C" int mxh_mg_level_sizes(mxh_mg* mg, int64_t* rows, int64_t* nnz, int* nlevels) {
  const int n = (int)std::min<size_t>(mg->rows.size(), 16);
  for (int i = 0; i < n; ++i) { if (rows) rows[i] = mg->rows[i]; if (nnz) nnz[i] = mg->nnz[i]; }
  if (nlevels) *nlevels = n;
  return 0;
}
