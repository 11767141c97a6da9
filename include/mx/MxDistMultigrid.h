// Aggregation multigrid on a decomposed grid: the same hierarchy mx::buildAggregationHierarchy builds on one rank
// (2x2x2 lattice aggregates per component, piecewise-constant P, R = P^T, Galerkin R A P; the stand-in for ML's
// smoothed aggregation, reference src/MxMagWaveOp.cpp:425-449,524-531), with every level row-partitioned like the
// operators themselves (one rank per MxGridDomain box, src/MxGrid.cpp:171-193):
//
//   * unknowns of level l are named by their lattice id on the level's lattice, comp + ncomp * cell;
//   * aggregates never straddle two boxes (coarsening stops at the level where a box edge stops being a multiple of
//     the aggregate size), so P and R are rank-local and a coarse row is formed from this rank's finer rows alone;
//   * coarse columns are lattice ids of the coarse lattice, whoever owns them: the MxCrsMatrix of the level agrees
//     its ghost import with the neighbours exactly like a fine operator;
//   * the V-cycle (mxg_mg_*, Jacobi sweeps) then imports ghosts inside each level product, so the preconditioner
//     is the single-rank one up to summation order - inner iteration counts do not grow with the number of boxes.
#ifndef MX_DIST_MULTIGRID_H
#define MX_DIST_MULTIGRID_H

#include <array>

#include "MxCrsMatrix.hpp"
#include "mx_mg.hpp"   // maxwell_amd/csrc/host
#include "mxh.h"

class MxDistMultigrid {
 public:
  // A0: row-partitioned fine operator (square on its map); n: global cell counts; COLLECTIVE
  MxDistMultigrid(RCP<MxCrsMatrix<double> > A0, int ncomp, const int n[3], int64_t minCoarse, int maxLevels, int sweeps,
                  int coarseSweeps, double omega, double coarseScale)
      : mNcomp(ncomp) {
    RCP<MxComm> comm = A0->getRangeMap()->getComm();
    const int np = comm->numProc();
    for (int i = 0; i < 3; ++i) mN[i] = n[i];
    // every rank knows every box (the partition is a pure function of n and the number of ranks)
    std::vector<std::array<int, 3> > boxLo(np);
    for (int r = 0; r < np; ++r) {
      int lo[3], hi[3];
      if (mxh_partition(n, np, r, mProcGrid, lo, hi) != 0) throw 1;
      for (int i = 0; i < 3; ++i) boxLo[r][i] = lo[i];
    }
    mA.push_back(A0);
    int64_t dims[3] = {n[0] + 1, n[1] + 1, n[2] + 1};
    int level = 0;
    // levels above this many global rows are row-partitioned; below, the hierarchy continues replicated on every
    // rank (one all-gather per V-cycle instead of a latency-bound ghost import per sweep and level)
    // (4e5: at 256^3 .. 512^3 cells the levels of 2e7 and 2.6e6 rows stay partitioned; the 3.3e5-row level costs a rank
    // ~15 us per product when replicated, less than one latency-bound ghost import)
    int64_t replicateBelow = 400000;
    if (const char* e = getenv("MXG_DIST_MG_REPLICATE")) replicateBelow = atoll(e);
    while ((int)mA.size() < maxLevels &&
           (int64_t)mA.back()->getRangeMap()->getGlobalNumIndices() > std::max(minCoarse, replicateBelow)) {
      bool aligned = true;
      for (int r = 0; r < np; ++r)
        for (int i = 0; i < 3; ++i) aligned = aligned && (boxLo[r][i] % (1 << (level + 1)) == 0);
      if (!aligned) break;
      if (!coarsen(level, dims)) break;
      for (int i = 0; i < 3; ++i) dims[i] = (dims[i] + 1) / 2;
      ++level;
    }
    std::vector<mxg_csr*> a, p, r;
    for (auto& m : mA) a.push_back(m->getRawMatrix());
    for (auto& m : mP) p.push_back(m->getRawMatrix());
    for (auto& m : mR) r.push_back(m->getRawMatrix());
    MX_ABI(mxg_mg_create(comm->getCtx(), (int)a.size(), a.data(), p.data(), r.data(), sweeps, sweeps, coarseSweeps, omega,
                         coarseScale, &mMg), "MxDistMultigrid");
    // (one rank: only when asked for - MXG_DIST_MG=force runs this class on a single rank, which is how the gathered
    // levels and their all-gather are exercised over a one-rank RCCL communicator in the tests)
    if (np > 1 || (getenv("MXG_DIST_MG") && std::string(getenv("MXG_DIST_MG")) == "force"))
      replicateTail(dims, minCoarse, maxLevels - (int)mA.size() + 1, sweeps, coarseSweeps, omega, coarseScale);
  }
  ~MxDistMultigrid() {
    if (mMg) mxg_mg_destroy(mMg);
    if (mTail) mxh_mg_destroy(mTail);
  }
  mxg_mg* handle() const { return mMg; }
  // Chebyshev smoothing on the row-partitioned levels and on the replicated ones (COLLECTIVE: all-reduced norms)
  void setChebyshev(double ratio) {
    MX_ABI(mxg_mg_set_chebyshev(mMg, ratio), "MxDistMultigrid::setChebyshev");
    if (mTail) MX_ABI(mxg_mg_set_chebyshev(mxh_mg_handle(mTail), ratio), "MxDistMultigrid::setChebyshev (replicated levels)");
  }
  int numLevels() const { return (int)mA.size(); }
  int numReplicatedLevels() const { return mTailLevels; }
  int64_t globalRows(int l) const { return (int64_t)mA[l]->getRangeMap()->getGlobalNumIndices(); }

 private:
  MxDistMultigrid(const MxDistMultigrid&);
  MxDistMultigrid& operator=(const MxDistMultigrid&);

  // lattice id on the next coarser lattice
  int64_t coarseId(int64_t gid, const int64_t dims[3], const int64_t cd[3]) const {
    const int64_t cell = gid / mNcomp, comp = gid % mNcomp;
    const int64_t k = cell % dims[2], j = (cell / dims[2]) % dims[1], i = cell / (dims[2] * dims[1]);
    return comp + (int64_t)mNcomp * (((i >> 1) * cd[1] + (j >> 1)) * cd[2] + (k >> 1));
  }

  bool coarsen(int level, const int64_t dims[3]) {
    RCP<MxCrsMatrix<double> > Af = mA.back();
    RCP<MxMap> fmap = Af->getRangeMap();
    RCP<MxComm> comm = fmap->getComm();
    const mx::Csr<double>& A = Af->getHostCsr();
    const int64_t nf = A.nrows, nown = nf, ncolsLocal = A.ncols;
    const int64_t cd[3] = {(dims[0] + 1) / 2, (dims[1] + 1) / 2, (dims[2] + 1) / 2};
    // unknowns with a zero diagonal take no part (they stay exactly zero, as in the serial hierarchy)
    std::vector<uint8_t> active((size_t)nf);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nf; ++i) {
      uint8_t act = 0;
      for (int64_t q = A.rowptr[i]; q < A.rowptr[i + 1]; ++q)
        if (A.col[q] == i && A.val[q] != 0.0) act = 1;
      active[i] = act;
    }
    std::vector<uint8_t> gactive = Af->importGhostsHost(active);          // collective
    // coarse lattice id of every local column (owned, then ghosts); -1 = inactive
    mx::UVec<int64_t> ckey((size_t)ncolsLocal);
    const std::vector<int64_t>& ghosts = Af->ghostGids();
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < ncolsLocal; ++c) {
      const bool act = c < nown ? active[c] : gactive[c - nown];
      const int64_t gid = c < nown ? (int64_t)fmap->getGlobalIndex((MxIndex)c) : ghosts[c - nown];
      ckey[c] = act ? coarseId(gid, dims, cd) : -1;
    }
    // owned coarse unknowns: the keys of the active owned rows, ascending
    std::vector<int64_t> ckeys;
    for (int64_t i = 0; i < nf; ++i) if (ckey[i] >= 0) ckeys.push_back(ckey[i]);
    std::sort(ckeys.begin(), ckeys.end());
    ckeys.erase(std::unique(ckeys.begin(), ckeys.end()), ckeys.end());
    const int64_t nc = (int64_t)ckeys.size();
    double tot[2] = {(double)nc, (double)nf};
    MX_ABI(mxg_comm_allreduce_host(comm->getCtx(), tot, 2), "MxDistMultigrid (level sizes)");
    if (tot[0] < 0.5 || tot[0] >= tot[1]) return false;
    mx::UVec<int32_t> agg((size_t)nf);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nf; ++i)
      agg[i] = ckey[i] >= 0 ? (int32_t)(std::lower_bound(ckeys.begin(), ckeys.end(), ckey[i]) - ckeys.begin()) : -1;
    // P: one unit entry per active fine row;  R = P^T with the fine rows of an aggregate ascending
    mx::Csr<double> P, R;
    P.nrows = nf; P.ncols = nc;
    P.rowptr.resize(nf + 1);
    P.rowptr[0] = 0;
    for (int64_t i = 0; i < nf; ++i) P.rowptr[i + 1] = P.rowptr[i] + (agg[i] >= 0 ? 1 : 0);
    P.col.resize(P.rowptr[nf]); P.val.resize(P.rowptr[nf]);
    for (int64_t i = 0; i < nf; ++i) if (agg[i] >= 0) { P.col[P.rowptr[i]] = agg[i]; P.val[P.rowptr[i]] = 1.0; }
    R.nrows = nc; R.ncols = nf;
    R.rowptr.resize(nc + 1);
    for (int64_t a = 0; a <= nc; ++a) R.rowptr[a] = 0;
    for (int64_t i = 0; i < nf; ++i) if (agg[i] >= 0) R.rowptr[agg[i] + 1]++;
    for (int64_t a = 0; a < nc; ++a) R.rowptr[a + 1] += R.rowptr[a];
    R.col.resize(R.rowptr[nc]); R.val.resize(R.rowptr[nc]);
    {
      mx::UVec<int64_t> pos(R.rowptr.begin(), R.rowptr.end() - 1);
      for (int64_t i = 0; i < nf; ++i)
        if (agg[i] >= 0) { const int64_t q = pos[agg[i]]++; R.col[q] = (int32_t)i; R.val[q] = 1.0; }
    }
    // Galerkin rows R A P of the owned coarse unknowns, columns as coarse lattice ids: entries of the aggregate's fine
    // rows (fine-row order, then storage order), sorted by coarse column, repeated columns summed in that order
    mx::UVec<int64_t> cnt((size_t)nc), crowptr((size_t)nc + 1);
    auto gatherRow = [&](int64_t a, std::vector<std::pair<int64_t, double> >& buf) {
      buf.clear();
      for (int64_t t = R.rowptr[a]; t < R.rowptr[a + 1]; ++t) {
        const int64_t i = R.col[t];
        for (int64_t q = A.rowptr[i]; q < A.rowptr[i + 1]; ++q) {
          const int64_t c = ckey[A.col[q]];
          if (c >= 0) buf.emplace_back(c, A.val[q]);
        }
      }
      std::stable_sort(buf.begin(), buf.end(), [](const std::pair<int64_t, double>& x, const std::pair<int64_t, double>& y) { return x.first < y.first; });
      size_t w = 0;
      for (size_t x = 0; x < buf.size(); ++x) {
        if (w > 0 && buf[w - 1].first == buf[x].first) buf[w - 1].second += buf[x].second;
        else buf[w++] = buf[x];
      }
      buf.resize(w);
    };
#pragma omp parallel
    {
      std::vector<std::pair<int64_t, double> > buf;
#pragma omp for schedule(dynamic, 1024)
      for (int64_t a = 0; a < nc; ++a) { gatherRow(a, buf); cnt[a] = (int64_t)buf.size(); }
    }
    crowptr[0] = 0;
    for (int64_t a = 0; a < nc; ++a) crowptr[a + 1] = crowptr[a] + cnt[a];
    mx::UVec<int64_t> ccol((size_t)crowptr[nc]);
    mx::UVec<double> cval((size_t)crowptr[nc]);
#pragma omp parallel
    {
      std::vector<std::pair<int64_t, double> > buf;
#pragma omp for schedule(dynamic, 1024)
      for (int64_t a = 0; a < nc; ++a) {
        gatherRow(a, buf);
        for (size_t x = 0; x < buf.size(); ++x) { ccol[crowptr[a] + x] = buf[x].first; cval[crowptr[a] + x] = buf[x].second; }
      }
    }
    // coarse map: owner of a coarse cell = owner of its lowest fine cell (aggregates do not straddle boxes)
    std::vector<MxIndex> cg(ckeys.begin(), ckeys.end());
    RCP<MxMap> cmap = rcp(new MxMap((size_t)(tot[0] + 0.5), cg, comm));
    const int ncomp = mNcomp, shiftBits = level + 1;
    const int nn[3] = {mN[0], mN[1], mN[2]}, pg[3] = {mProcGrid[0], mProcGrid[1], mProcGrid[2]};
    const int64_t d1 = cd[1], d2 = cd[2];
    cmap->setOwnerFunction([ncomp, shiftBits, nn, pg, d1, d2](MxIndex gid) {
      const int64_t cell = (int64_t)gid / ncomp;
      int c[3] = {(int)(cell / (d2 * d1)), (int)((cell / d2) % d1), (int)(cell % d2)};
      for (int i = 0; i < 3; ++i) { c[i] = (int)std::min<int64_t>((int64_t)c[i] << shiftBits, nn[i]); }
      return mxh_owner_of_cell(nn, pg, c);
    });
    mP.push_back(rcp(new MxCrsMatrix<double>(fmap, cmap, P)));
    mR.push_back(rcp(new MxCrsMatrix<double>(cmap, fmap, R)));
    mA.push_back(rcp(new MxCrsMatrix<double>(cmap, cmap, crowptr, ccol, cval)));   // collective: ghost import of the level
    return true;
  }

  // The last row-partitioned level gathered on every rank (rows in rank order, so a rank's own rows are one
  // segment), continued by the single-rank builder; the device cycle gathers that level's right-hand side once.
  void replicateTail(const int64_t dims[3], int64_t minCoarse, int maxLevels, int sweeps, int coarseSweeps, double omega,
                     double coarseScale) {
    RCP<MxCrsMatrix<double> > Al = mA.back();
    RCP<MxMap> map = Al->getRangeMap();
    RCP<MxComm> comm = map->getComm();
    const int np = comm->numProc();
    const mx::Csr<double>& A = Al->getHostCsr();
    const int64_t nr = A.nrows, nnz = A.nnz(), nown = nr;
    const std::vector<int64_t>& ghosts = Al->ghostGids();
    // message (int64 words): rows, entries, row ids, row pointers, column ids, value bit patterns
    std::vector<int64_t> msg((size_t)(2 + nr + (nr + 1) + 2 * nnz));
    size_t w = 0;
    msg[w++] = nr; msg[w++] = nnz;
    for (int64_t i = 0; i < nr; ++i) msg[w++] = (int64_t)map->getGlobalIndex((MxIndex)i);
    for (int64_t i = 0; i <= nr; ++i) msg[w++] = A.rowptr[i];
    for (int64_t q = 0; q < nnz; ++q) msg[w++] = A.col[q] < nown ? (int64_t)map->getGlobalIndex((MxIndex)A.col[q]) : ghosts[A.col[q] - nown];
    for (int64_t q = 0; q < nnz; ++q) { int64_t bits; std::memcpy(&bits, &A.val[q], sizeof bits); msg[w++] = bits; }
    std::vector<int64_t> counts(np, 0);
    MX_ABI(mxg_comm_allgather_i64(comm->getCtx(), (int64_t)(msg.size() * sizeof(int64_t)), counts.data()), "MxDistMultigrid (tail sizes)");
    int64_t total = 0;
    for (int r = 0; r < np; ++r) total += counts[r] / (int64_t)sizeof(int64_t);
    std::vector<int64_t> all((size_t)total + 1);
    MX_ABI(mxg_comm_allgatherv(comm->getCtx(), msg.data(), (int64_t)(msg.size() * sizeof(int64_t)), counts.data(), all.data()),
           "MxDistMultigrid (tail gather)");
    // pass 1: row ids in gathered order
    std::vector<int64_t> rowsOf(np), off(np + 1, 0);
    mx::UVec<int64_t> gids;
    int64_t nnzAll = 0;
    {
      size_t p = 0;
      for (int r = 0; r < np; ++r) {
        const int64_t n_r = all[p], z_r = all[p + 1];
        rowsOf[r] = n_r; off[r + 1] = off[r] + n_r; nnzAll += z_r;
        for (int64_t i = 0; i < n_r; ++i) gids.push_back(all[p + 2 + i]);
        p += (size_t)(2 + n_r + (n_r + 1) + 2 * z_r);
      }
    }
    const int64_t nAll = off[np];
    std::vector<std::pair<int64_t, int64_t> > lookup((size_t)nAll);
    for (int64_t i = 0; i < nAll; ++i) lookup[i] = std::make_pair(gids[i], i);
    std::sort(lookup.begin(), lookup.end());
    mx::Csr<double> G;
    G.nrows = nAll; G.ncols = nAll;
    G.rowptr.resize(nAll + 1);
    G.col.resize(nnzAll); G.val.resize(nnzAll);
    G.rowptr[0] = 0;
    {
      size_t p = 0;
      int64_t row = 0, q0 = 0;
      for (int r = 0; r < np; ++r) {
        const int64_t n_r = all[p], z_r = all[p + 1];
        const int64_t* rp = &all[p + 2 + n_r];
        const int64_t* ck = rp + (n_r + 1);
        const int64_t* vb = ck + z_r;
        for (int64_t i = 0; i < n_r; ++i) G.rowptr[row + i + 1] = q0 + rp[i + 1];
        for (int64_t q = 0; q < z_r; ++q) {
          auto it = std::lower_bound(lookup.begin(), lookup.end(), std::make_pair(ck[q], (int64_t)-1));
          if (it == lookup.end() || it->first != ck[q]) { std::cout << "MxDistMultigrid: column " << ck[q] << " of the gathered level has no row\n"; throw 1; }
          G.col[q0 + q] = (int32_t)it->second;
          std::memcpy(&G.val[q0 + q], &vb[q], sizeof(double));
        }
        row += n_r; q0 += z_r;
        p += (size_t)(2 + n_r + (n_r + 1) + 2 * z_r);
      }
    }
    const int nl[3] = {(int)dims[0] - 1, (int)dims[1] - 1, (int)dims[2] - 1};
    mTail = mxh_mg_build_gids(comm->getCtx(), G, gids, mNcomp, nl, minCoarse, std::max(maxLevels, 1), sweeps, coarseSweeps,
                              omega, coarseScale);
    int nlev = 0;
    mxh_mg_level_sizes(mTail, nullptr, nullptr, &nlev);
    mTailLevels = nlev;
    MX_ABI(mxg_mg_set_tail(mMg, mxh_mg_handle(mTail), rowsOf.data()), "MxDistMultigrid (tail)");
  }

  mxh_mg* mTail = nullptr;
  int mTailLevels = 0;
  int mNcomp;
  int mN[3];
  int mProcGrid[3] = {1, 1, 1};
  std::vector<RCP<MxCrsMatrix<double> > > mA, mP, mR;
  mxg_mg* mMg = nullptr;
};

#endif
