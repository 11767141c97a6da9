// See mx_yee.hpp for the reference lines each routine follows.
#include "mx_yee.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace mx {

namespace {
struct Tick {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  bool on = std::getenv("MXH_VERBOSE") != nullptr;
  void operator()(const char* what) {
    auto now = std::chrono::steady_clock::now();
    if (on) std::fprintf(stderr, "[mx_yee] %-12s %.3f s\n", what, std::chrono::duration<double>(now - t).count());
    t = now;
  }
};
}  // namespace

BcKind parseBc(const std::string& s) {
  if (s == "periodic") return BC_PERIODIC;
  if (s == "pec") return BC_PEC;
  if (s == "pmc") return BC_PMC;
  throw std::runtime_error("unknown boundary condition '" + s + "'");
}

template <> double YeeAssembler<double>::toScalar(const std::complex<double>& z) { return z.real(); }
template <> std::complex<double> YeeAssembler<std::complex<double>>::toScalar(const std::complex<double>& z) { return z; }

template <typename S>
YeeAssembler<S>::YeeAssembler(const YeeConfig& cfg) : cfg_(cfg) {
  complexProblem_ = false;
  for (int i = 0; i < 3; ++i) {
    n_[i] = cfg.n[i];
    d_[i] = cfg.size[i] / double(cfg.n[i]);                    // src/MxGrid.cpp:18
    phase_[i] = std::exp(std::complex<double>(0.0, 1.0) * cfg.phaseShifts[i]);  // src/MxGridField.cpp:34-38
    if (cfg.phaseShifts[i] != 0.0) complexProblem_ = true;
  }
  for (const YeeDielectric& dl : cfg.dielectrics) if (dl.lossy()) lossy_ = true;
  for (const YeeMu& m : cfg.mus) if (m.lossy()) { muLossy_ = true; complexProblem_ = true; }
  for (const YeePml& pm : cfg.pmls) {                             // MxPML::setProfileEndpoints, src/MxPML.h:107-115
    if (!pm.shape) throw std::runtime_error("YeeAssembler: PML '" + pm.name + "' has no shape");
    PmlGeom g;
    double dd[3];
    for (int i = 0; i < 3; ++i) dd[i] = pm.end[i] - pm.begin[i];
    g.width = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
    for (int i = 0; i < 3; ++i) g.dir[i] = dd[i] / g.width;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        g.outer[i][j] = std::complex<double>(g.dir[i] * g.dir[j], 0.0);      // (a, 0)(b, 0): setOuterProds, :85-92
        g.compl_[i][j] = std::complex<double>(i == j ? 1.0 : 0.0, 0.0) - g.outer[i][j];
      }
    pmlGeom_.push_back(g);
    lossy_ = true;
  }
  if (lossy_) complexProblem_ = true;
  if (complexProblem_ && sizeof(S) == sizeof(double))
    throw std::runtime_error("phase shifts / lossy dielectrics / PMLs need the complex assembler");
  B_.name = "bfield"; E_.name = "efield"; P_.name = "psifield"; D_.name = "dfield";
  B_.ncomp = 3; E_.ncomp = 3; P_.ncomp = 1; D_.ncomp = 3;
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 3; ++i) {
      B_.xi0[c][i] = (i == c);   // face centre: 0 along the normal, d/2 elsewhere
      E_.xi0[c][i] = (i != c);   // edge centre: d/2 along the edge, 0 elsewhere
      P_.xi0[c][i] = false;      // cell centre
    }
  auto compBc = [](BcKind b, bool normal, bool isB) -> int {
    if (b == BC_PERIODIC) return CBC_PERIODIC;
    // B: PEC -> normal ZERO / tangential CONSTANT; E the opposite; PMC swaps
    const bool zero = (b == BC_PEC) ? (isB ? normal : !normal) : (isB ? !normal : normal);
    return zero ? CBC_ZERO : CBC_CONSTANT;
  };
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 3; ++i) {
      B_.lbc[c][i] = compBc(cfg.lower[i], i == c, true);
      B_.ubc[c][i] = compBc(cfg.upper[i], i == c, true);
      E_.lbc[c][i] = compBc(cfg.lower[i], i == c, false);
      E_.ubc[c][i] = compBc(cfg.upper[i], i == c, false);
    }
  for (int i = 0; i < 3; ++i) {
    auto psiBc = [](BcKind b) { return b == BC_PERIODIC ? CBC_PERIODIC : (b == BC_PEC ? CBC_CONSTANT : CBC_ZERO); };
    P_.lbc[0][i] = psiBc(cfg.lower[i]);
    P_.ubc[0][i] = psiBc(cfg.upper[i]);
  }
  windowed_ = false;
  for (int i = 0; i < 3; ++i) {
    periodicAxis_[i] = cfg.lower[i] == BC_PERIODIC && cfg.upper[i] == BC_PERIODIC;
    const int lo = cfg.boxLo[i], hi = cfg.boxHi[i] < 0 ? n_[i] + 1 : cfg.boxHi[i];
    if (lo < 0 || hi > n_[i] + 1 || lo >= hi) throw std::runtime_error("YeeAssembler: bad owned box");
    cfg_.boxLo[i] = lo; cfg_.boxHi[i] = hi;
    splitAxis_[i] = !(lo == 0 && hi == n_[i] + 1);
    if (!splitAxis_[i]) { wlo_[i] = 0; whi_[i] = n_[i] + 1; continue; }   // whole axis
    windowed_ = true;
    wlo_[i] = lo - cfg.halo; whi_[i] = hi + cfg.halo;
    if (!periodicAxis_[i]) { wlo_[i] = std::max(wlo_[i], 0); whi_[i] = std::min(whi_[i], n_[i] + 1); }
    else if (whi_[i] - wlo_[i] > n_[i]) throw std::runtime_error("YeeAssembler: box + halo wraps onto itself");
  }
  Tick tick;
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 3; ++i) { D_.xi0[c][i] = E_.xi0[c][i]; D_.lbc[c][i] = E_.lbc[c][i]; D_.ubc[c][i] = E_.ubc[c][i]; }
  H_.name = "hfield"; H_.ncomp = 3;
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 3; ++i) { H_.xi0[c][i] = B_.xi0[c][i]; H_.lbc[c][i] = B_.lbc[c][i]; H_.ubc[c][i] = B_.ubc[c][i]; }
  const bool diel = !cfg_.dielectrics.empty() || !cfg_.pmls.empty();   // src/MxEMSim.cpp:89
  if (cfg_.pec) {                                              // src/MxEMSim.cpp:121-129
    addShapeRep(B_, *cfg_.pec, B_.frac);
    addShapeRep(E_, *cfg_.pec, E_.frac);
    addShapeRep(P_, *cfg_.pec, P_.frac);
    if (diel) addShapeRep(D_, *cfg_.pec, D_.frac);
  }
  for (const YeeDielectric& dl : cfg_.dielectrics) {           // src/MxEMSim.cpp:131-146 (not on B)
    if (!dl.shape) throw std::runtime_error("YeeAssembler: dielectric '" + dl.name + "' has no shape");
    for (YeeField* F : {&E_, &D_, &P_}) {
      F->dielFrac.emplace_back();
      addShapeRep(*F, *dl.shape, F->dielFrac.back());
    }
  }
  for (const YeeMu& m : cfg_.mus) {                            // src/MxEMSim.cpp:148-163 (not on E, D)
    if (!m.shape) throw std::runtime_error("YeeAssembler: mu object '" + m.name + "' has no shape");
    for (YeeField* F : {&B_, &H_, &P_}) {
      F->muFrac.emplace_back();
      addShapeRep(*F, *m.shape, F->muFrac.back());
    }
  }
  tick("fractions");
  // "any used component keeps the whole B cell" lookup over the extended lattice
  bcellUsed_.resize(extSize());
#pragma omp parallel for schedule(static)
  for (int i = wlo_[0] - 1; i <= whi_[0]; ++i)
    for (int j = wlo_[1] - 1; j <= whi_[1]; ++j)
      for (int k = wlo_[2] - 1; k <= whi_[2]; ++k) {
        const int u[3] = {i, j, k};
        int c[3];
        canon(u, c);
        bcellUsed_[extIndex(u)] = baseUse(B_, 0, c) || baseUse(B_, 1, c) || baseUse(B_, 2, c);
      }
  tick("bcellUsed");
  setMap(B_);
  setMap(E_);
  setMap(P_);
  if (diel) {                                                  // dfield->setMap(efield), src/MxEMSim.cpp:196-197
    D_.gids = E_.gids; D_.wcell = E_.wcell; D_.owned = E_.owned; D_.lidOfWin = E_.lidOfWin;
  }
  if (!cfg_.mus.empty() || !cfg_.pmls.empty()) {               // hfield->setMap(bfield), src/MxEMSim.cpp:96,198-199
    H_.gids = B_.gids; H_.wcell = B_.wcell; H_.owned = B_.owned; H_.lidOfWin = B_.lidOfWin;
  }
  tick("maps");
}

template <typename S>
int64_t YeeAssembler<S>::cellToGlobal(const int c[3]) const {   // src/MxGrid.h:98-113
  int64_t res = 0, f = 1;
  for (int i = 2; i >= 0; --i) {
    const int ni = n_[i] + 1;
    int g = c[i];
    if (g >= ni) g -= ni; else if (g < 0) g += ni;
    res += (int64_t)g * f;
    f *= ni;
  }
  return res;
}

template <typename S>
void YeeAssembler<S>::globalToCell(int64_t idx, int c[3]) const {  // src/MxGrid.h:116-127
  int64_t f = 1;
  for (int i = 2; i >= 0; --i) {
    const int ni = n_[i] + 1;
    c[i] = (int)((idx / f) % ni);
    f *= ni;
  }
}

template <typename S>
void YeeAssembler<S>::interior(const YeeField& F, int comp, const int c[3], int out[3]) const {
  // getInteriorComp, src/MxGridField.cpp:80-142; case 1 is a stand-alone `if`, so for cell < 0
  // case 2 always has the last word
  for (int i = 0; i < 3; ++i) {
    const int lbc = F.lbc[comp][i], ubc = F.ubc[comp][i];
    const bool xi0 = F.xi0[comp][i];
    const int ci = c[i], ni = n_[i];
    int o = ci;
    if (ci < 0) {
      o = (lbc == CBC_PERIODIC) ? ni + ci : -ci - 1;
    } else if (ci == ni && xi0) {
      if (ubc == CBC_PERIODIC) o = 0;
    } else if (ci >= ni && xi0) {
      o = (ubc == CBC_PERIODIC) ? ci - ni : ni - (ci - ni);
    } else if (ci >= ni) {
      o = (ubc == CBC_PERIODIC) ? ci - ni : ni - (ci - ni + 1);
    }
    out[i] = o;
  }
}

template <typename S>
double YeeAssembler<S>::fracAt(const YeeField& F, int comp, const int c[3]) const {
  if (F.frac.empty()) return 1.0;                             // src/MxGridField.hpp:274-277
  const int64_t e = extIndex(c);
  return e < 0 ? 0.0 : F.frac[comp][e];                        // outside the window: unknown -> unused
}

template <typename S>
void YeeAssembler<S>::addShapeRep(const YeeField& F, const Shape& shape,
                                  std::vector<UVec<double>>& out) const {   // src/MxGridField.cpp:193-226
  const int64_t next = extSize();
  out.assign(F.ncomp, UVec<double>());
  for (int comp = 0; comp < F.ncomp; ++comp) {
    UVec<double>& fr = out[comp];
    fr.resize(next);
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = wlo_[0] - 1; i <= whi_[0]; ++i)
      for (int j = wlo_[1] - 1; j <= whi_[1]; ++j)
        for (int k = wlo_[2] - 1; k <= whi_[2]; ++k) {
          const int u[3] = {i, j, k};
          int c[3], ic[3];
          canon(u, c);
          interior(F, comp, c, ic);
          Vec3 p;                                              // calcCompFrac, src/MxGridField.hpp:177-183
          for (int a = 0; a < 3; ++a)
            p[a] = (cfg_.origin[a] + double(ic[a]) * d_[a]) + (F.xi0[comp][a] ? 0.0 : 0.5 * d_[a]);
          double f;
          if (&F == &B_ || &F == &D_) f = rectFraction(shape, p, comp, d_[(comp + 1) % 3], d_[(comp + 2) % 3]);
          else if (&F == &E_ || &F == &H_) f = segmentFraction(shape, p, comp, d_[comp]);
          else f = boxFraction(shape, p, d_[0], d_[1], d_[2]);
          fr[extIndex(u)] = f;
        }
  }
}

template <typename S>
bool YeeAssembler<S>::baseUse(const YeeField& F, int comp, const int c[3]) const {
  // MxGridField::useCompInMap, src/MxGridField.cpp:41-77
  if (!F.frac.empty() && fracAt(F, comp, c) == 0.0) return false;
  for (int i = 0; i < 3; ++i) {
    const bool xi0 = F.xi0[comp][i];
    if (c[i] == 0 && xi0) {
      if (F.lbc[comp][i] == CBC_ZERO) return false;
    } else if (c[i] == n_[i] && !xi0) {
      return false;
    } else if (c[i] == n_[i] && xi0) {
      if (F.ubc[comp][i] == CBC_ZERO || F.ubc[comp][i] == CBC_PERIODIC) return false;
    }
  }
  return true;
}

template <typename S>
bool YeeAssembler<S>::bUse(const int c[3]) const {             // src/MxYeeFitBField.cpp:63-80
  const int64_t e = extIndex(c);
  return e >= 0 && bcellUsed_[e] != 0;
}

template <typename S>
bool YeeAssembler<S>::eUse(int comp, const int c[3]) const {   // src/MxYeeFitEField.cpp:53-88
  if (!baseUse(E_, comp, c)) return false;
  const int c2 = (comp + 1) % 3, c3 = (comp + 2) % 3;
  int t[3] = {c[0], c[1], c[2]};
  if (!bUse(t)) return false;
  t[c3]--; if (!bUse(t)) return false; t[c3]++;
  t[c2]--; if (!bUse(t)) return false;
  return true;
}

template <typename S>
bool YeeAssembler<S>::psiUse(const int c[3]) const {           // src/MxYeePsiField.cpp:100-114
  if (!baseUse(P_, 0, c)) return false;
  int t[3] = {c[0], c[1], c[2]};
  for (int i = 0; i < 3; ++i) {
    if (bUse(t)) return true;
    t[i]++; if (bUse(t)) return true; t[i]--;
  }
  return false;
}

template <typename S>
bool YeeAssembler<S>::use(const YeeField& F, int comp, const int c[3]) const {
  if (&F == &B_) return bUse(c);
  if (&F == &E_) return eUse(comp, c);
  return psiUse(c);
}

template <typename S>
std::complex<double> YeeAssembler<S>::baseFactor(const YeeField& F, int comp, const int c[3]) const {
  // MxGridField::getCompFactor, src/MxGridField.cpp:145-190
  if (!F.frac.empty() && fracAt(F, comp, c) == 0.0) return 0.0;
  std::complex<double> res(1.0, 0.0);
  for (int i = 0; i < 3; ++i) {
    const int lbc = F.lbc[comp][i], ubc = F.ubc[comp][i];
    const bool xi0 = F.xi0[comp][i];
    if (c[i] < 0) {
      if (lbc == CBC_PERIODIC) res /= phase_[i];
      else if (lbc == CBC_ZERO) res *= -1.0;
    } else if (c[i] == 0 && xi0) {
      if (lbc == CBC_ZERO) res *= 0.0;
    } else if (c[i] == n_[i] && xi0) {
      if (ubc == CBC_PERIODIC) res *= phase_[i];
      else if (ubc == CBC_ZERO) res *= 0.0;
    } else if (c[i] >= n_[i]) {
      if (ubc == CBC_PERIODIC) res *= phase_[i];
      else if (ubc == CBC_ZERO) res *= -1.0;
    }
  }
  return res;
}

template <typename S>
std::complex<double> YeeAssembler<S>::factor(const YeeField& F, int comp, const int c[3]) const {
  const std::complex<double> res = baseFactor(F, comp, c);
  if (&F == &B_) {                                             // src/MxYeeFitBField.cpp:83-90
    if (cfg_.dmFrac > 0.0 && !F.frac.empty() && fracAt(F, comp, c) < cfg_.dmFrac) return 0.0;
    return res;
  }
  if (&F == &D_) return res;                                   // no override: MxGridField::getCompFactor
  int probe[3] = {c[0], c[1], c[2]};
  if (!cfg_.strict) interior(F, comp, c, probe);
  return use(F, comp, probe) ? res : std::complex<double>(0.0);  // src/MxYeeFitEField.cpp:91-98, MxYeePsiField.cpp:117-124
}

template <typename S>
int64_t YeeAssembler<S>::gid(const YeeField& F, int comp, const int c[3]) const {
  int ic[3];
  interior(F, comp, c, ic);
  return comp + (int64_t)F.ncomp * cellToGlobal(ic);           // src/MxGridField.hpp:142-145
}

template <typename S>
void YeeAssembler<S>::setMap(YeeField& F) {                     // src/MxGridField.cpp:256-297
  // every window cell that is a lattice node of the global grid, in window C order (== ascending
  // global id when the window is the whole grid); entries of the owned box are flagged
  const int64_t nwin = (int64_t)(whi_[0] - wlo_[0]) * (whi_[1] - wlo_[1]) * (whi_[2] - wlo_[2]);
  F.lidOfWin.resize(nwin * F.ncomp);
  UVec<uint8_t> keep(nwin * F.ncomp);
#pragma omp parallel for schedule(static)
  for (int64_t w = 0; w < nwin; ++w) {
    int u[3], c[3];
    winCell(w, u);
    canon(u, c);
    for (int comp = 0; comp < F.ncomp; ++comp) keep[w * F.ncomp + comp] = use(F, comp, c);
  }
  // local ids ascend with the window cell: count per block of cells, prefix sum, fill the blocks in parallel
  const int64_t nblk = std::max<int64_t>(1, std::min<int64_t>(nwin, 1024));
  auto blkStart = [&](int64_t b) { return b * nwin / nblk; };
  UVec<int64_t> first(nblk + 1);
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < nblk; ++b) {
    int64_t n = 0;
    for (int64_t q = blkStart(b) * F.ncomp; q < blkStart(b + 1) * F.ncomp; ++q) n += keep[q];
    first[b + 1] = n;
  }
  first[0] = 0;
  for (int64_t b = 0; b < nblk; ++b) first[b + 1] += first[b];
  const int64_t total = first[nblk];
  F.gids.resize(total); F.wcell.resize(total); F.owned.resize(total);
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < nblk; ++b) {
    int64_t count = first[b];
    for (int64_t w = blkStart(b); w < blkStart(b + 1); ++w) {
      int u[3], c[3];
      winCell(w, u);
      canon(u, c);
      bool own = true;   // ownership follows the canonical cell (node n of a split periodic axis is node 0)
      for (int i = 0; i < 3; ++i) own = own && c[i] >= cfg_.boxLo[i] && c[i] < cfg_.boxHi[i];
      for (int comp = 0; comp < F.ncomp; ++comp) {
        const int64_t q = w * F.ncomp + comp;
        if (keep[q]) {
          F.lidOfWin[q] = (int32_t)count;
          F.gids[count] = gid(F, comp, c);
          F.wcell[count] = (int32_t)w;
          F.owned[count] = own;
          ++count;
        } else {
          F.lidOfWin[q] = -1;
        }
      }
    }
  }
}

template <typename S>
Csr<S> YeeAssembler<S>::assemble(const YeeField& rowF, const YeeField& colF, bool skipDeadRows,
                                 const std::vector<std::vector<Entry>>& stencil) const {
  const int64_t nrows = (int64_t)rowF.gids.size();
  size_t maxe = 0;
  for (auto& s : stencil) maxe = std::max(maxe, s.size());
  UVec<int32_t> cols(nrows * maxe);
  UVec<S> vals(nrows * maxe);
  UVec<int32_t> cnt(nrows);
  bool bad = false;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrows; ++r) {
    const int comp = (int)(rowF.gids[r] % rowF.ncomp);
    int ucell[3], cell[3];
    winCell(rowF.wcell[r], ucell);
    canon(ucell, cell);
    if (skipDeadRows && factor(rowF, comp, cell) == std::complex<double>(0.0)) { cnt[r] = 0; continue; }
    int n = 0;
    for (const Entry& e : stencil[comp]) {
      const int c[3] = {cell[0] + e.off[0], cell[1] + e.off[1], cell[2] + e.off[2]};
      const std::complex<double> f = factor(colF, e.comp, c);
      if (f == std::complex<double>(0.0)) continue;           // `if (factors[i] != 0)` in every assembler
      int ic[3];
      interior(colF, e.comp, c, ic);
      const int32_t lid = lidOf(colF, e.comp, ic);
      if (lid < 0) { if (!windowed_) bad = true; continue; }   // windowed: beyond the halo, row is discarded later
      cols[r * maxe + n] = lid;
      vals[r * maxe + n] = toScalar(f) * S(e.val);
      ++n;
    }
    cnt[r] = n;
  }
  if (bad) throw std::runtime_error("assemble: column outside the domain map (" + rowF.name + " x " + colF.name + ")");
  // rows are short: sort each by column and sum repeated (row, col) pairs in insertion order,
  // as Epetra's FillComplete does (only reachable on 2-cell periodic directions)
  Csr<S> m;
  m.nrows = nrows; m.ncols = (int64_t)colF.gids.size();
  m.rowptr.resize(nrows + 1); m.rowptr[0] = 0;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrows; ++r) {
    int32_t* c = cols.data() + r * maxe;
    S* v = vals.data() + r * maxe;
    int n = cnt[r];
    for (int x = 1; x < n; ++x) {
      const int32_t cc = c[x]; const S vv = v[x];
      int y = x - 1;
      while (y >= 0 && c[y] > cc) { c[y + 1] = c[y]; v[y + 1] = v[y]; --y; }
      c[y + 1] = cc; v[y + 1] = vv;
    }
    int w = 0;
    for (int x = 0; x < n; ++x) {
      if (w > 0 && c[w - 1] == c[x]) v[w - 1] += v[x];
      else { c[w] = c[x]; v[w] = v[x]; ++w; }
    }
    cnt[r] = w;
  }
  for (int64_t r = 0; r < nrows; ++r) m.rowptr[r + 1] = m.rowptr[r] + cnt[r];
  m.col.resize(m.rowptr[nrows]);
  m.val.resize(m.rowptr[nrows]);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrows; ++r) {
    std::copy(cols.begin() + r * maxe, cols.begin() + r * maxe + cnt[r], m.col.begin() + m.rowptr[r]);
    std::copy(vals.begin() + r * maxe, vals.begin() + r * maxe + cnt[r], m.val.begin() + m.rowptr[r]);
  }
  return m;
}

template <typename S>
Csr<S> YeeAssembler<S>::curlE() const {                        // src/MxYeeDeyMittraCurlE.cpp:117-179
  std::vector<std::vector<Entry>> st(3);
  for (int c0 = 0; c0 < 3; ++c0) {
    const int c1 = (c0 + 1) % 3, c2 = (c0 + 2) % 3;
    Entry e0{c2, {0, 0, 0}, 1.0 / d_[c1]}; e0.off[c1] = 1;
    Entry e1{c2, {0, 0, 0}, -(1.0 / d_[c1])};
    Entry e2{c1, {0, 0, 0}, -(1.0 / d_[c2])}; e2.off[c2] = 1;
    Entry e3{c1, {0, 0, 0}, 1.0 / d_[c2]};
    st[c0] = {e0, e1, e2, e3};
  }
  return assemble(B_, E_, true, st);
}

template <typename S>
Csr<S> YeeAssembler<S>::curlB() const {                        // src/MxYeeDeyMittraCurlB.cpp:113-169
  std::vector<std::vector<Entry>> st(3);
  for (int c0 = 0; c0 < 3; ++c0) {
    const int c1 = (c0 + 1) % 3, c2 = (c0 + 2) % 3;
    Entry e0{c2, {0, 0, 0}, 1.0 / d_[c1]};
    Entry e1{c2, {0, 0, 0}, -(1.0 / d_[c1])}; e1.off[c1] = -1;
    Entry e2{c1, {0, 0, 0}, -(1.0 / d_[c2])};
    Entry e3{c1, {0, 0, 0}, 1.0 / d_[c2]}; e3.off[c2] = -1;
    st[c0] = {e0, e1, e2, e3};
  }
  return assemble(E_, B_, false, st);
}

template <typename S>
Csr<S> YeeAssembler<S>::divB() const {                         // src/MxYeeDeyMittraDivB.cpp:157-200
  std::vector<std::vector<Entry>> st(1);
  for (int c = 0; c < 3; ++c) {
    Entry a{c, {0, 0, 0}, 1.0 / d_[c]}; a.off[c] = 1;
    Entry b{c, {0, 0, 0}, -1.0 / d_[c]};
    st[0].push_back(a);
    st[0].push_back(b);
  }
  return assemble(P_, B_, false, st);
}

template <typename S>
Csr<S> YeeAssembler<S>::gradPsi() const {                      // src/MxYeeDeyMittraGradPsi.cpp:23-84
  std::vector<std::vector<Entry>> st(3);
  for (int c = 0; c < 3; ++c) {
    const double inv = 1.0 / d_[c];
    Entry a{0, {0, 0, 0}, inv};
    Entry b{0, {0, 0, 0}, -inv}; b.off[c] = -1;
    st[c] = {a, b};
  }
  return assemble(B_, P_, true, st);
}

template <typename S>
Csr<S> YeeAssembler<S>::fracs(const YeeField& F, bool inverse) const {   // src/MxYeeDeyMittraFracs.cpp:30-131
  UVec<S> d(F.gids.size());
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < (int64_t)F.gids.size(); ++r) {
    int ucell[3], cell[3];
    winCell(F.wcell[r], ucell);
    canon(ucell, cell);
    double v = fracAt(F, (int)(F.gids[r] % F.ncomp), cell);
    if (inverse) v = (v == 0.0) ? 0.0 : 1.0 / v;
    d[r] = S(v);
  }
  return diagonal<S>(d);
}

template <typename S>
double YeeAssembler<S>::dielFracAt(const YeeField& F, size_t diel, int comp, const int c[3]) const {
  const int64_t e = extIndex(c);
  return e < 0 ? 0.0 : F.dielFrac[diel][comp][e];
}

namespace {

void matmul3(const double x[3][3], const double y[3][3], double out[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += x[i][k] * y[k][j];
      out[i][j] = acc;
    }
}

void inverse3(const double a[3][3], double r[3][3]) {   // cofactors, same expression order as mx::inverse
  const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) -
                     a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                     a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
  r[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) / det;
  r[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det;
  r[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
  r[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) / det;
  r[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det;
  r[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
  r[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) / det;
  r[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det;
  r[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
}

// gamma = I + n n^T (I - eps) / (n . eps n), pi = eps gamma   (src/MxYeeFitInvEps.cpp:345-347,365-366)
void gammaPi(const double eps[3][3], const double n[3], double gamma[3][3], double pi[3][3]) {
  double epsn[3], ime[3][3], nn[3][3], prod[3][3];
  for (int i = 0; i < 3; ++i) epsn[i] = eps[i][0] * n[0] + eps[i][1] * n[1] + eps[i][2] * n[2];
  const double denom = n[0] * epsn[0] + n[1] * epsn[1] + n[2] * epsn[2];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { ime[i][j] = (i == j ? 1.0 : 0.0) - eps[i][j]; nn[i][j] = n[i] * n[j]; }
  matmul3(nn, ime, prod);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) gamma[i][j] = (i == j ? 1.0 : 0.0) + prod[i][j] / denom;
  matmul3(eps, gamma, pi);
}

}  // namespace

// tupleUpdate, src/MxYeeFitInvEps.cpp:305-385: length-fraction weighted rows of gamma times the
// inverse of the area-fraction weighted rows of pi, over the dielectrics and the background
template <typename S>
void YeeAssembler<S>::tupleUpdate(const int comps[3], const int cells[3][3], const double n[3],
                                  double out[3][3]) const {
  double aveG[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, aveP[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double lsum[3] = {0, 0, 0}, asum[3] = {0, 0, 0}, gamma[3][3], pi[3][3];
  for (size_t i = 0; i < cfg_.dielectrics.size(); ++i) {
    double lf[3] = {0, 0, 0}, af[3] = {0, 0, 0};
    for (int j = 0; j < 3; ++j) {
      const int c = comps[j];
      lf[c] = dielFracAt(E_, i, c, cells[j]);
      af[c] = dielFracAt(D_, i, c, cells[j]);
      lsum[c] += lf[c];
      asum[c] += af[c];
    }
    gammaPi(cfg_.dielectrics[i].eps, n, gamma, pi);
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) { aveG[j][k] += lf[j] * gamma[j][k]; aveP[j][k] += af[j] * pi[j][k]; }
  }
  const double bg[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  gammaPi(bg, n, gamma, pi);
  for (int j = 0; j < 3; ++j) {
    const double lfj = 1.0 - lsum[j], afj = 1.0 - asum[j];
    for (int k = 0; k < 3; ++k) { aveG[j][k] += lfj * gamma[j][k]; aveP[j][k] += afj * pi[j][k]; }
  }
  double inv[3][3];
  inverse3(aveP, inv);
  matmul3(aveG, inv, out);
}

namespace {

typedef std::complex<double> Cx;
// explicit complex arithmetic (naive product and quotient), mirrored operation by operation in oracle/yee.py
inline Cx cmul(const Cx& a, const Cx& b) { return Cx(a.real() * b.real() - a.imag() * b.imag(), a.real() * b.imag() + a.imag() * b.real()); }
inline Cx cdiv(const Cx& a, const Cx& b) {
  const double den = b.real() * b.real() + b.imag() * b.imag();
  return Cx((a.real() * b.real() + a.imag() * b.imag()) / den, (a.imag() * b.real() - a.real() * b.imag()) / den);
}
void matmul3c(const Cx x[3][3], const Cx y[3][3], Cx out[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Cx acc(0.0, 0.0);
      for (int k = 0; k < 3; ++k) acc = acc + cmul(x[i][k], y[k][j]);
      out[i][j] = acc;
    }
}
void inverse3c(const Cx a[3][3], Cx r[3][3]) {
  const Cx det = cmul(a[0][0], cmul(a[1][1], a[2][2]) - cmul(a[1][2], a[2][1])) -
                 cmul(a[0][1], cmul(a[1][0], a[2][2]) - cmul(a[1][2], a[2][0])) +
                 cmul(a[0][2], cmul(a[1][0], a[2][1]) - cmul(a[1][1], a[2][0]));
  r[0][0] = cdiv(cmul(a[1][1], a[2][2]) - cmul(a[1][2], a[2][1]), det);
  r[0][1] = cdiv(cmul(a[0][2], a[2][1]) - cmul(a[0][1], a[2][2]), det);
  r[0][2] = cdiv(cmul(a[0][1], a[1][2]) - cmul(a[0][2], a[1][1]), det);
  r[1][0] = cdiv(cmul(a[1][2], a[2][0]) - cmul(a[1][0], a[2][2]), det);
  r[1][1] = cdiv(cmul(a[0][0], a[2][2]) - cmul(a[0][2], a[2][0]), det);
  r[1][2] = cdiv(cmul(a[0][2], a[1][0]) - cmul(a[0][0], a[1][2]), det);
  r[2][0] = cdiv(cmul(a[1][0], a[2][1]) - cmul(a[1][1], a[2][0]), det);
  r[2][1] = cdiv(cmul(a[0][1], a[2][0]) - cmul(a[0][0], a[2][1]), det);
  r[2][2] = cdiv(cmul(a[0][0], a[1][1]) - cmul(a[0][1], a[1][0]), det);
}
void gammaPiC(const Cx eps[3][3], const double n[3], Cx gamma[3][3], Cx pi[3][3]) {
  Cx nc[3] = {Cx(n[0], 0.0), Cx(n[1], 0.0), Cx(n[2], 0.0)}, epsn[3], ime[3][3], nn[3][3], prod[3][3];
  for (int i = 0; i < 3; ++i) epsn[i] = cmul(eps[i][0], nc[0]) + cmul(eps[i][1], nc[1]) + cmul(eps[i][2], nc[2]);
  const Cx denom = cmul(nc[0], epsn[0]) + cmul(nc[1], epsn[1]) + cmul(nc[2], epsn[2]);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { ime[i][j] = Cx(i == j ? 1.0 : 0.0, 0.0) - eps[i][j]; nn[i][j] = cmul(nc[i], nc[j]); }
  matmul3c(nn, ime, prod);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) gamma[i][j] = Cx(i == j ? 1.0 : 0.0, 0.0) + cdiv(prod[i][j], denom);
  matmul3c(eps, gamma, pi);
}

Cx det3c(const Cx m[3][3]) {   // det(MxDimMatrix<T, 3>), src/MxDimMatrix.hpp:376-384
  return cmul(cmul(m[0][0], m[1][1]), m[2][2]) + cmul(cmul(m[0][1], m[1][2]), m[2][0]) + cmul(cmul(m[0][2], m[1][0]), m[2][1]) -
         cmul(cmul(m[0][2], m[1][1]), m[2][0]) - cmul(cmul(m[0][1], m[1][0]), m[2][2]) - cmul(cmul(m[0][0], m[1][2]), m[2][1]);
}
// (T m T) / det, the matrix / scalar quotient as m * (1 / det) (src/MxDimMatrix.hpp:152-154)
void pmlSandwich(const Cx t[3][3], const Cx& det, Cx m[3][3]) {
  Cx a[3][3], b[3][3];
  matmul3c(t, m, a);
  matmul3c(a, t, b);
  const Cx inv = cdiv(Cx(1.0, 0.0), det);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = cmul(inv, b[i][j]);
}

}  // namespace

template <typename S>
void YeeAssembler<S>::compCoord(const YeeField& F, int comp, const int cell[3], double p[3]) const {
  for (int a = 0; a < 3; ++a) p[a] = (cfg_.origin[a] + double(cell[a]) * d_[a]) + (F.xi0[comp][a] ? 0.0 : 0.5 * d_[a]);
}

// src/MxYeeFitInvEps.cpp:514-537 / src/MxYeeFitMu.cpp:385-410.  MxPML::isDiag() is false for every layer (setIsDiag
// counts |dir_i| < MxUtil::dEps with dEps = 0.0, src/MxUtil.hpp:26), so a row inside any layer takes the stencil path.
template <typename S>
bool YeeAssembler<S>::pmlTransform(const double p[3], bool inverse, Cx t[3][3], Cx& det) const {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[i][j] = Cx(i == j ? 1.0 : 0.0, 0.0);
  bool diag = true;
  Vec3 q; q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
  for (size_t l = 0; l < cfg_.pmls.size(); ++l) {
    const YeePml& pm = cfg_.pmls[l];
    if (!(pm.shape->func(q) > 0)) continue;
    diag = false;
    const PmlGeom& g = pmlGeom_[l];
    const double tt = ((p[0] - pm.begin[0]) * g.dir[0] + (p[1] - pm.begin[1]) * g.dir[1] + (p[2] - pm.begin[2]) * g.dir[2]) / g.width;
    const Cx sv(1.0, pm.alphaMax * std::pow(tt, pm.exponent));               // getAlpha, src/MxPML.h:168-171
    const Cx f = inverse ? sv : cdiv(Cx(1.0, 0.0), sv);                      // getInvS :145-149 / getS :151-161
    Cx m[3][3], r[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = cmul(f, g.outer[i][j]) + g.compl_[i][j];
    matmul3c(t, m, r);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[i][j] = r[i][j];
  }
  det = det3c(t);
  return diag;
}

// tupleUpdate of MxYeeFitMu in complex arithmetic (used with PMLs)
template <typename S>
void YeeAssembler<S>::muTupleUpdateC(const int comps[3], const int cells[3][3], const double n[3], Cx out[3][3]) const {
  Cx aveG[3][3], aveP[3][3], gamma[3][3], pi[3][3];
  for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) { aveG[j][k] = Cx(0.0, 0.0); aveP[j][k] = Cx(0.0, 0.0); }
  double lsum[3] = {0, 0, 0}, asum[3] = {0, 0, 0};
  for (size_t i = 0; i < cfg_.mus.size(); ++i) {
    double lf[3] = {0, 0, 0}, af[3] = {0, 0, 0};
    for (int j = 0; j < 3; ++j) {
      const int c = comps[j];
      lf[c] = muFracAt(H_, i, c, cells[j]);
      af[c] = muFracAt(B_, i, c, cells[j]);
      lsum[c] += lf[c];
      asum[c] += af[c];
    }
    Cx mu[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) mu[a][b] = Cx(cfg_.mus[i].mu[a][b], cfg_.mus[i].muIm[a][b]);
    gammaPiC(mu, n, gamma, pi);
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) {
        aveG[j][k] = aveG[j][k] + cmul(Cx(lf[j], 0.0), gamma[j][k]);
        aveP[j][k] = aveP[j][k] + cmul(Cx(af[j], 0.0), pi[j][k]);
      }
  }
  Cx bg[3][3];
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) bg[a][b] = Cx(a == b ? 1.0 : 0.0, 0.0);
  gammaPiC(bg, n, gamma, pi);
  for (int j = 0; j < 3; ++j) {
    const double lfj = 1.0 - lsum[j], afj = 1.0 - asum[j];
    for (int k = 0; k < 3; ++k) {
      aveG[j][k] = aveG[j][k] + cmul(Cx(lfj, 0.0), gamma[j][k]);
      aveP[j][k] = aveP[j][k] + cmul(Cx(afj, 0.0), pi[j][k]);
    }
  }
  Cx inv[3][3];
  inverse3c(aveG, inv);
  matmul3c(aveP, inv, out);
}

// tupleUpdate for complex permittivity (loss tangent)
template <typename S>
void YeeAssembler<S>::tupleUpdateC(const int comps[3], const int cells[3][3], const double n[3], Cx out[3][3]) const {
  Cx aveG[3][3], aveP[3][3], gamma[3][3], pi[3][3];
  for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) { aveG[j][k] = Cx(0.0, 0.0); aveP[j][k] = Cx(0.0, 0.0); }
  double lsum[3] = {0, 0, 0}, asum[3] = {0, 0, 0};
  for (size_t i = 0; i < cfg_.dielectrics.size(); ++i) {
    double lf[3] = {0, 0, 0}, af[3] = {0, 0, 0};
    for (int j = 0; j < 3; ++j) {
      const int c = comps[j];
      lf[c] = dielFracAt(E_, i, c, cells[j]);
      af[c] = dielFracAt(D_, i, c, cells[j]);
      lsum[c] += lf[c];
      asum[c] += af[c];
    }
    Cx eps[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) eps[a][b] = Cx(cfg_.dielectrics[i].eps[a][b], cfg_.dielectrics[i].epsIm[a][b]);
    gammaPiC(eps, n, gamma, pi);
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) {
        aveG[j][k] = aveG[j][k] + cmul(Cx(lf[j], 0.0), gamma[j][k]);
        aveP[j][k] = aveP[j][k] + cmul(Cx(af[j], 0.0), pi[j][k]);
      }
  }
  Cx bg[3][3];
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) bg[a][b] = Cx(a == b ? 1.0 : 0.0, 0.0);
  gammaPiC(bg, n, gamma, pi);
  for (int j = 0; j < 3; ++j) {
    const double lfj = 1.0 - lsum[j], afj = 1.0 - asum[j];
    for (int k = 0; k < 3; ++k) {
      aveG[j][k] = aveG[j][k] + cmul(Cx(lfj, 0.0), gamma[j][k]);
      aveP[j][k] = aveP[j][k] + cmul(Cx(afj, 0.0), pi[j][k]);
    }
  }
  Cx inv[3][3];
  inverse3c(aveP, inv);
  matmul3c(aveG, inv, out);
}

// getNormal, src/MxYeeFitInvEps.cpp:254-302
template <typename S>
void YeeAssembler<S>::stencilNormal(int comp0, const int cell[3], const int scomps[9], const int scells[9][3],
                                    double n[3]) const {
  Vec3 p;
  for (int a = 0; a < 3; ++a) p[a] = (cfg_.origin[a] + double(cell[a]) * d_[a]) + (E_.xi0[comp0][a] ? 0.0 : 0.5 * d_[a]);
  double norms[8][3];
  int count = 0;
  for (size_t i = 0; i < cfg_.dielectrics.size() && count < 8; ++i) {
    bool cut = false;
    for (int s = 0; s < 9; ++s) {
      const double lf = dielFracAt(E_, i, scomps[s], scells[s]), af = dielFracAt(D_, i, scomps[s], scells[s]);
      if ((lf != 0 && lf != 1) || (af != 0 && af != 1)) cut = true;
    }
    if (cut) {
      const Vec3 g = cfg_.dielectrics[i].shape->gradFunc(p);
      const double nn = norm(g);
      for (int k = 0; k < 3; ++k) norms[count][k] = g[k] / nn;
      ++count;
    }
  }
  if (count == 0) { n[0] = 1.0; n[1] = 0.0; n[2] = 0.0; return; }
  double tot[3] = {0, 0, 0};
  for (int i = 0; i < count; ++i) {
    double sgn = 1.0;
    if (i > 0 && (norms[0][0] * norms[i][0] + norms[0][1] * norms[i][1] + norms[0][2] * norms[i][2]) < 0) sgn = -1.0;
    for (int k = 0; k < 3; ++k) tot[k] += sgn * norms[i][k];
  }
  const double nn = std::sqrt(tot[0] * tot[0] + tot[1] * tot[1] + tot[2] * tot[2]);
  for (int k = 0; k < 3; ++k) n[k] = tot[k] / nn;
}

// MxYeeFitInvEps::setMatrix, src/MxYeeFitInvEps.cpp:421-560 (3-D, real eps, no PML).  Entries whose
// column is not in the map are left out (Epetra never imports a value for such a column).
template <typename S>
Csr<S> YeeAssembler<S>::invEps() const {
  const int64_t nrows = (int64_t)E_.gids.size();
  UVec<int32_t> cols(nrows * 9);
  UVec<S> vals(nrows * 9);
  UVec<int32_t> cnt(nrows);
  static const int tupleInds[8][3] = {{0, 1, 5}, {0, 1, 6}, {0, 2, 5}, {0, 2, 6}, {0, 3, 7}, {0, 3, 8}, {0, 4, 7}, {0, 4, 8}};
  const size_t nd = cfg_.dielectrics.size();
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t r = 0; r < nrows; ++r) {
    const int comp0 = (int)(E_.gids[r] % 3);
    int ucell[3], cell[3];
    winCell(E_.wcell[r], ucell);
    canon(ucell, cell);
    bool inDiel = false, epsDiag = false;
    std::complex<double> epsHere(1.0, 0.0);
    for (size_t i = 0; i < nd; ++i) {                                        // :462-479
      const double l0 = dielFracAt(E_, i, comp0, cell), a0 = dielFracAt(D_, i, comp0, cell);
      const double (*e)[3] = cfg_.dielectrics[i].eps;
      if (l0 == 1 && a0 == 1) {
        inDiel = true;
        epsDiag = !(std::abs(e[0][1]) > 0.0 || std::abs(e[0][2]) > 0.0 || std::abs(e[1][2]) > 0.0);   // checkIsDiag, dEps = 0
        epsHere = std::complex<double>(e[comp0][comp0], cfg_.dielectrics[i].epsIm[comp0][comp0]);
        break;
      } else if (l0 == 0 && a0 == 0) {
        continue;
      } else {
        inDiel = true;
        break;
      }
    }
    if (!inDiel) { epsDiag = true; epsHere = std::complex<double>(1.0, 0.0); }
    int32_t* rc = cols.data() + r * 9;
    S* rv = vals.data() + r * 9;
    const bool pml = hasPml();
    bool pmlDiag = true;
    Cx tr[3][3], det(1.0, 0.0);
    if (pml) {                                                               // :514-537
      double pc[3];
      compCoord(E_, comp0, cell, pc);
      pmlDiag = pmlTransform(pc, true, tr, det);
    }
    if (epsDiag && pmlDiag) {                                                // :542-549
      const std::complex<double> fac = factor(D_, comp0, cell);
      rc[0] = (int32_t)r;
      if (pml) rv[0] = toScalar(cdiv(cdiv(cmul(cmul(fac, tr[comp0][comp0]), tr[comp0][comp0]), det), epsHere));
      else rv[0] = toScalar(lossy_ ? cdiv(fac, epsHere) : fac / epsHere.real());
      cnt[r] = 1;
      continue;
    }
    const int comp1 = (comp0 + 1) % 3, comp2 = (comp0 + 2) % 3;              // getStencil, :35-71
    int scomps[9] = {comp0, comp1, comp1, comp1, comp1, comp2, comp2, comp2, comp2};
    int scells[9][3];
    for (int s = 0; s < 9; ++s) for (int a = 0; a < 3; ++a) scells[s][a] = cell[a];
    scells[1][comp1] -= 1;
    scells[3][comp0] += 1; scells[3][comp1] -= 1;
    scells[4][comp0] += 1;
    scells[5][comp2] -= 1;
    scells[7][comp0] += 1; scells[7][comp2] -= 1;
    scells[8][comp0] += 1;
    double normal[3];
    stencilNormal(comp0, cell, scomps, scells, normal);
    int use[8], nt = 0;                                                       // getTuples, :127-175
    for (int t = 0; t < 8; ++t) {
      bool ok = true;
      if (cfg_.pec)
        for (int j = 0; j < 3; ++j) {
          const int si = tupleInds[t][j];
          if (fracAt(E_, scomps[si], scells[si]) == 0) ok = false;
        }
      if (ok) use[nt++] = t;
    }
    double svals[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::complex<double> svalsC[9];
    const double ntd = double(nt);
    for (int u = 0; u < nt; ++u) {                                            // :519-531
      const int* t = tupleInds[use[u]];
      int tcomps[3], tcells[3][3];
      for (int j = 0; j < 3; ++j) { tcomps[j] = scomps[t[j]]; for (int a = 0; a < 3; ++a) tcells[j][a] = scells[t[j]][a]; }
      if (lossy_) {
        std::complex<double> m[3][3];
        tupleUpdateC(tcomps, tcells, normal, m);
        if (pml) pmlSandwich(tr, det, m);                                     // :567
        for (int j = 0; j < 3; ++j) svalsC[t[j]] = svalsC[t[j]] + m[comp0][scomps[t[j]]] / ntd;
      } else {
        double m[3][3];
        tupleUpdate(tcomps, tcells, normal, m);
        for (int j = 0; j < 3; ++j) svals[t[j]] += m[comp0][scomps[t[j]]] / ntd;
      }
    }
    int n = 0;
    for (int s = 0; s < 9; ++s) {                                             // :533-537
      const std::complex<double> fac = factor(D_, scomps[s], scells[s]);
      int ic[3];
      interior(D_, scomps[s], scells[s], ic);
      const int32_t lid = lidOf(D_, scomps[s], ic);
      if (lid < 0) continue;
      rc[n] = lid;
      rv[n] = lossy_ ? toScalar(cmul(svalsC[s], fac)) : S(svals[s]) * toScalar(fac);
      ++n;
    }
    // sort by column, merge repeated columns in insertion order (Epetra FillComplete)
    for (int x = 1; x < n; ++x) {
      const int32_t cc = rc[x]; const S vv = rv[x];
      int y = x - 1;
      while (y >= 0 && rc[y] > cc) { rc[y + 1] = rc[y]; rv[y + 1] = rv[y]; --y; }
      rc[y + 1] = cc; rv[y + 1] = vv;
    }
    int w = 0;
    for (int x = 0; x < n; ++x) {
      if (w > 0 && rc[w - 1] == rc[x]) rv[w - 1] += rv[x];
      else { rc[w] = rc[x]; rv[w] = rv[x]; ++w; }
    }
    cnt[r] = w;
  }
  Csr<S> m;
  m.nrows = nrows; m.ncols = nrows;
  m.rowptr.resize(nrows + 1); m.rowptr[0] = 0;
  for (int64_t r = 0; r < nrows; ++r) m.rowptr[r + 1] = m.rowptr[r] + cnt[r];
  m.col.resize(m.rowptr[nrows]);
  m.val.resize(m.rowptr[nrows]);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrows; ++r) {
    std::copy(cols.begin() + r * 9, cols.begin() + r * 9 + cnt[r], m.col.begin() + m.rowptr[r]);
    std::copy(vals.begin() + r * 9, vals.begin() + r * 9 + cnt[r], m.val.begin() + m.rowptr[r]);
  }
  return m;
}

template <typename S>
double YeeAssembler<S>::muFracAt(const YeeField& F, size_t obj, int comp, const int c[3]) const {
  const int64_t e = extIndex(c);
  return e < 0 ? 0.0 : F.muFrac[obj][comp][e];
}

// tupleUpdate of MxYeeFitMu with mInvert = false, src/MxYeeFitMu.cpp:205-290: the averages of the inverse
// permittivity with H (dual-edge length fractions) and B (face area fractions) in the roles of E and D, multiplied
// the other way round: avePi * aveGamma^-1 maps the H tuple to the B tuple
template <typename S>
void YeeAssembler<S>::muTupleUpdate(const int comps[3], const int cells[3][3], const double n[3],
                                    double out[3][3]) const {
  double aveG[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, aveP[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double lsum[3] = {0, 0, 0}, asum[3] = {0, 0, 0}, gamma[3][3], pi[3][3];
  for (size_t i = 0; i < cfg_.mus.size(); ++i) {
    double lf[3] = {0, 0, 0}, af[3] = {0, 0, 0};
    for (int j = 0; j < 3; ++j) {
      const int c = comps[j];
      lf[c] = muFracAt(H_, i, c, cells[j]);
      af[c] = muFracAt(B_, i, c, cells[j]);
      lsum[c] += lf[c];
      asum[c] += af[c];
    }
    gammaPi(cfg_.mus[i].mu, n, gamma, pi);
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) { aveG[j][k] += lf[j] * gamma[j][k]; aveP[j][k] += af[j] * pi[j][k]; }
  }
  const double bg[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  gammaPi(bg, n, gamma, pi);
  for (int j = 0; j < 3; ++j) {
    const double lfj = 1.0 - lsum[j], afj = 1.0 - asum[j];
    for (int k = 0; k < 3; ++k) { aveG[j][k] += lfj * gamma[j][k]; aveP[j][k] += afj * pi[j][k]; }
  }
  double inv[3][3];
  inverse3(aveG, inv);
  matmul3(aveP, inv, out);
}

// getNormal, src/MxYeeFitMu.cpp:134-186
template <typename S>
void YeeAssembler<S>::muStencilNormal(int comp0, const int cell[3], const int scomps[9], const int scells[9][3],
                                      double n[3]) const {
  Vec3 p;
  for (int a = 0; a < 3; ++a) p[a] = (cfg_.origin[a] + double(cell[a]) * d_[a]) + (B_.xi0[comp0][a] ? 0.0 : 0.5 * d_[a]);
  double norms[8][3];
  int count = 0;
  for (size_t i = 0; i < cfg_.mus.size() && count < 8; ++i) {
    bool cut = false;
    for (int s = 0; s < 9; ++s) {
      const double lf = muFracAt(H_, i, scomps[s], scells[s]), af = muFracAt(B_, i, scomps[s], scells[s]);
      if ((lf != 0 && lf != 1) || (af != 0 && af != 1)) cut = true;
    }
    if (cut) {
      const Vec3 g = cfg_.mus[i].shape->gradFunc(p);
      const double nn = norm(g);
      for (int k = 0; k < 3; ++k) norms[count][k] = g[k] / nn;
      ++count;
    }
  }
  if (count == 0) { n[0] = 1.0; n[1] = 0.0; n[2] = 0.0; return; }
  double tot[3] = {0, 0, 0};
  for (int i = 0; i < count; ++i) {
    double sgn = 1.0;
    if (i > 0 && (norms[0][0] * norms[i][0] + norms[0][1] * norms[i][1] + norms[0][2] * norms[i][2]) < 0) sgn = -1.0;
    for (int k = 0; k < 3; ++k) tot[k] += sgn * norms[i][k];
  }
  const double nn = std::sqrt(tot[0] * tot[0] + tot[1] * tot[1] + tot[2] * tot[2]);
  for (int k = 0; k < 3; ++k) n[k] = tot[k] / nn;
}

// MxYeeFitMu::setMatrix with mInvert = false, src/MxYeeFitMu.cpp:294-463 (3-D, real mu, no PML).  Entries whose
// column is not in the map are left out (Epetra never imports a value for such a column).
template <typename S>
Csr<S> YeeAssembler<S>::muOp() const {
  const int64_t nrows = (int64_t)B_.gids.size();
  UVec<int32_t> cols(nrows * 9);
  UVec<S> vals(nrows * 9);
  UVec<int32_t> cnt(nrows);
  static const int tupleInds[8][3] = {{0, 1, 5}, {0, 1, 6}, {0, 2, 5}, {0, 2, 6}, {0, 3, 7}, {0, 3, 8}, {0, 4, 7}, {0, 4, 8}};
  const size_t nm = cfg_.mus.size();
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t r = 0; r < nrows; ++r) {
    const int comp0 = (int)(B_.gids[r] % 3);
    int ucell[3], cell[3];
    winCell(B_.wcell[r], ucell);
    canon(ucell, cell);
    bool inMu = false, muDiag = false;
    Cx muHere(1.0, 0.0);
    for (size_t i = 0; i < nm; ++i) {                                         // :333-358
      const double l0 = muFracAt(H_, i, comp0, cell), a0 = muFracAt(B_, i, comp0, cell);
      const double (*m)[3] = cfg_.mus[i].mu;
      if (l0 == 1 && a0 == 1) {
        inMu = true;
        muDiag = !(std::abs(m[0][1]) > 0.0 || std::abs(m[0][2]) > 0.0 || std::abs(m[1][2]) > 0.0);   // checkIsDiag, dEps = 0
        muHere = Cx(m[comp0][comp0], cfg_.mus[i].muIm[comp0][comp0]);
        break;
      } else if (l0 == 0 && a0 == 0) {
        continue;
      } else {
        inMu = true;
        break;
      }
    }
    if (!inMu) { muDiag = true; muHere = Cx(1.0, 0.0); }
    int32_t* rc = cols.data() + r * 9;
    S* rv = vals.data() + r * 9;
    const bool pml = hasPml();
    bool pmlDiag = true;
    Cx tr[3][3], det(1.0, 0.0);
    if (pml) {                                                                // :385-410 (mInvert = false: getS)
      double pc[3];
      compCoord(H_, comp0, cell, pc);
      pmlDiag = pmlTransform(pc, false, tr, det);
    }
    const bool cplxMat = pml || muLossy_;
    if (muDiag && pmlDiag) {                                                  // :413-424
      const std::complex<double> fac = factor(B_, comp0, cell);
      rc[0] = (int32_t)r;
      Cx v = muLossy_ ? cmul(fac, muHere) : fac * muHere.real();
      if (pml) v = cmul(v, cdiv(cmul(tr[comp0][comp0], tr[comp0][comp0]), det));
      rv[0] = toScalar(v);
      cnt[r] = 1;
      continue;
    }
    const int comp1 = (comp0 + 1) % 3, comp2 = (comp0 + 2) % 3;               // getStencil, :27-71
    int scomps[9] = {comp0, comp1, comp1, comp1, comp1, comp2, comp2, comp2, comp2};
    int scells[9][3];
    for (int s = 0; s < 9; ++s) for (int a = 0; a < 3; ++a) scells[s][a] = cell[a];
    scells[1][comp0] -= 1;
    scells[2][comp0] -= 1; scells[2][comp1] += 1;
    scells[4][comp1] += 1;
    scells[5][comp0] -= 1;
    scells[6][comp0] -= 1; scells[6][comp2] += 1;
    scells[8][comp2] += 1;
    double normal[3];
    muStencilNormal(comp0, cell, scomps, scells, normal);
    int use[8], nt = 0;                                                        // getTuples, :75-128
    for (int t = 0; t < 8; ++t) {
      bool ok = true;
      if (cfg_.pec)
        for (int j = 0; j < 3; ++j) {
          const int si = tupleInds[t][j];
          if (fracAt(B_, scomps[si], scells[si]) == 0) ok = false;
        }
      if (ok) use[nt++] = t;
    }
    double svals[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Cx svalsC[9];
    const double ntd = double(nt);
    for (int u = 0; u < nt; ++u) {                                             // :427-440
      const int* t = tupleInds[use[u]];
      int tcomps[3], tcells[3][3];
      for (int j = 0; j < 3; ++j) { tcomps[j] = scomps[t[j]]; for (int a = 0; a < 3; ++a) tcells[j][a] = scells[t[j]][a]; }
      if (cplxMat) {
        Cx m[3][3];
        muTupleUpdateC(tcomps, tcells, normal, m);
        if (pml) pmlSandwich(tr, det, m);                                     // :434
        for (int j = 0; j < 3; ++j) svalsC[t[j]] = svalsC[t[j]] + m[comp0][scomps[t[j]]] / ntd;
      } else {
        double m[3][3];
        muTupleUpdate(tcomps, tcells, normal, m);
        for (int j = 0; j < 3; ++j) svals[t[j]] += m[comp0][scomps[t[j]]] / ntd;
      }
    }
    int n = 0;
    for (int s = 0; s < 9; ++s) {                                              // :441-442
      const std::complex<double> fac = factor(B_, scomps[s], scells[s]);
      int ic[3];
      interior(B_, scomps[s], scells[s], ic);
      const int32_t lid = lidOf(B_, scomps[s], ic);
      if (lid < 0) continue;
      rc[n] = lid;
      rv[n] = cplxMat ? toScalar(cmul(svalsC[s], fac)) : S(svals[s]) * toScalar(fac);
      ++n;
    }
    for (int x = 1; x < n; ++x) {   // sort by column, merge repeated columns in insertion order (Epetra FillComplete)
      const int32_t cc = rc[x]; const S vv = rv[x];
      int y = x - 1;
      while (y >= 0 && rc[y] > cc) { rc[y + 1] = rc[y]; rv[y + 1] = rv[y]; --y; }
      rc[y + 1] = cc; rv[y + 1] = vv;
    }
    int w = 0;
    for (int x = 0; x < n; ++x) {
      if (w > 0 && rc[w - 1] == rc[x]) rv[w - 1] += rv[x];
      else { rc[w] = rc[x]; rv[w] = rv[x]; ++w; }
    }
    cnt[r] = w;
  }
  Csr<S> m;
  m.nrows = nrows; m.ncols = nrows;
  m.rowptr.resize(nrows + 1); m.rowptr[0] = 0;
  for (int64_t r = 0; r < nrows; ++r) m.rowptr[r + 1] = m.rowptr[r] + cnt[r];
  m.col.resize(m.rowptr[nrows]);
  m.val.resize(m.rowptr[nrows]);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrows; ++r) {
    std::copy(cols.begin() + r * 9, cols.begin() + r * 9 + cnt[r], m.col.begin() + m.rowptr[r]);
    std::copy(vals.begin() + r * 9, vals.begin() + r * 9 + cnt[r], m.val.begin() + m.rowptr[r]);
  }
  return m;
}

// cellAveInvMuOperator, src/MxYeeFitMu.cpp:532-600: volume-fraction average of 3/trace(mu) on the Psi map
template <typename S>
Csr<S> YeeAssembler<S>::invMuVolAve() const {
  UVec<S> d(P_.gids.size());
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < (int64_t)P_.gids.size(); ++r) {
    int ucell[3], cell[3];
    winCell(P_.wcell[r], ucell);
    canon(ucell, cell);
    double out = 0.0, sums = 0.0;
    Cx acc(0.0, 0.0);
    for (size_t i = 0; i < cfg_.mus.size(); ++i) {
      const double fr = muFracAt(P_, i, 0, cell);
      const double (*m)[3] = cfg_.mus[i].mu;
      const double (*mi)[3] = cfg_.mus[i].muIm;
      sums = sums + fr;
      if (muLossy_) acc = acc + cmul(Cx(fr, 0.0), cdiv(Cx(3.0, 0.0), Cx(m[0][0], mi[0][0]) + Cx(m[1][1], mi[1][1]) + Cx(m[2][2], mi[2][2])));
      else out = out + fr * (3.0 / (m[0][0] + m[1][1] + m[2][2]));
    }
    if (muLossy_) acc = acc + Cx((1.0 - sums) * (3.0 / 3.0), 0.0);
    else { out = out + (1.0 - sums) * (3.0 / 3.0); acc = Cx(out, 0.0); }
    // (no PML 'volume' factor here: the reference compiles that block out, src/MxYeeFitMu.cpp:764-778 `#if 0`; only
    //  the cell-averaged inverse permittivity carries det(invS), src/MxYeeFitInvEps.cpp:779-791)
    d[r] = muLossy_ ? toScalar(acc) : S(out);
  }
  return diagonal<S>(d);
}

// cellAveInvEpsOperator, src/MxYeeFitInvEps.cpp:700-790: volume-fraction average of 3/trace(eps)
template <typename S>
Csr<S> YeeAssembler<S>::invEpsVolAve() const {
  UVec<S> d(P_.gids.size());
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < (int64_t)P_.gids.size(); ++r) {
    int ucell[3], cell[3];
    winCell(P_.wcell[r], ucell);
    canon(ucell, cell);
    if (lossy_) {
      std::complex<double> acc(0.0, 0.0);
      double sums = 0.0;
      for (size_t i = 0; i < cfg_.dielectrics.size(); ++i) {
        const YeeDielectric& dl = cfg_.dielectrics[i];
        const std::complex<double> tr = std::complex<double>(dl.eps[0][0], dl.epsIm[0][0]) + std::complex<double>(dl.eps[1][1], dl.epsIm[1][1]) +
                                        std::complex<double>(dl.eps[2][2], dl.epsIm[2][2]);
        const double fr = dielFracAt(P_, i, 0, cell);
        sums = sums + fr;
        acc = acc + cmul(std::complex<double>(fr, 0.0), cdiv(std::complex<double>(3.0, 0.0), tr));
      }
      acc = acc + std::complex<double>((1.0 - sums) * (3.0 / 3.0), 0.0);
      if (hasPml()) {                                     // :779-791: det of the getS product
        double pc[3];
        Cx tr[3][3], det;
        compCoord(P_, 0, cell, pc);
        pmlTransform(pc, false, tr, det);
        acc = cmul(acc, det);
      }
      d[r] = toScalar(acc);
      continue;
    }
    double out = 0.0, sums = 0.0;
    for (size_t i = 0; i < cfg_.dielectrics.size(); ++i) {
      const double (*e)[3] = cfg_.dielectrics[i].eps;
      const double fr = dielFracAt(P_, i, 0, cell);
      sums = sums + fr;
      out = out + fr * (3.0 / (e[0][0] + e[1][1] + e[2][2]));
    }
    out = out + (1.0 - sums) * (3.0 / 3.0);
    d[r] = S(out);
  }
  return diagonal<S>(d);
}

template <typename S>
std::vector<double> YeeAssembler<S>::usedMask() const {
  std::vector<double> m(B_.gids.size());
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < (int64_t)B_.gids.size(); ++r) {
    int ucell[3], cell[3];
    winCell(B_.wcell[r], ucell);
    canon(ucell, cell);
    m[r] = factor(B_, (int)(B_.gids[r] % 3), cell) == std::complex<double>(0.0) ? 0.0 : 1.0;
  }
  return m;
}

template <typename S>
Csr<S> YeeAssembler<S>::multiplyOps(const std::map<std::string, Csr<S>>& ops,
                                    const std::vector<std::string>& names) {
  Csr<S> res = ops.at(names[0]);
  for (size_t i = 1; i < names.size(); ++i) res = multiply(ops.at(names[i]), res);
  return res;
}

template <typename S>
std::map<std::string, Csr<S>> YeeAssembler<S>::singleOps() const {   // MxEMOps::setSingleOps, src/MxEMOps.cpp:39-168
  std::map<std::string, Csr<S>> o;
  Tick tick;
  o["curlE"] = curlE(); tick("curlE");
  o["curlB"] = curlB(); tick("curlB");
  o["divB"] = divB(); tick("divB");
  o["gradPsi"] = gradPsi(); tick("gradPsi");
  if (hasPec()) {
    o["dmA"] = fracs(B_, false);
    o["dmL"] = fracs(E_, false);
    o["dmVInv"] = fracs(P_, true);
    tick("dm*");
  }
  if (hasDielectric() || hasPml()) {                                       // src/MxEMOps.cpp:66-77
    o["invEps"] = invEps();
    o["invEpsVolAve"] = invEpsVolAve();
  }
  if (hasMu() || hasPml()) {                                               // src/MxEMOps.cpp:128-141
    o["mu"] = muOp();
    o["invMuVolAve"] = invMuVolAve();
  }
  return o;
}

template <typename S>
std::map<std::string, Csr<S>> YeeAssembler<S>::waveOps() const { return waveOps(singleOps()); }

// the operator chain from primitives the caller already holds (a sim created in device-chain mode keeps singleOps():
// forming the host chain later must not assemble every Yee / Dey-Mittra / invEps primitive a second time)
template <typename S>
std::map<std::string, Csr<S>> YeeAssembler<S>::waveOps(std::map<std::string, Csr<S>> o) const {   // src/MxMagWaveOp.cpp:138-245
  Tick tick;
  const bool pec = hasPec();
  const bool diel = hasDielectric();
  tick("primitives");
  auto opt = [&](std::vector<std::string>& v, const char* nm) { if (pec) v.push_back(nm); };
  std::vector<std::string> names;
  names = {"curlB"}; if (diel) names.push_back("invEps"); opt(names, "dmL"); names.push_back("curlE");
  o["curlCurl"] = multiplyOps(o, names);                                   // :145-153
  const bool mag = hasMu();
  names.clear(); if (mag) names.push_back("mu"); opt(names, "dmA"); names.push_back("divB"); opt(names, "dmVInv");
  if (mag) names.push_back("invMuVolAve");
  if (diel) names.push_back("invEpsVolAve");
  if (mag) names.push_back("invMuVolAve");
  names.push_back("gradPsi"); if (mag) names.push_back("mu"); opt(names, "dmA");
  o["gradDiv"] = multiplyOps(o, names);                                    // :156-179
  tick("products");
  o["vecLapl"] = purge(add(o["curlCurl"], S(1.0), o["gradDiv"], S(-1.0)));  // :183-193
  tick("vecLapl");
  names = {"gradPsi"}; if (mag) names.push_back("mu"); opt(names, "dmA"); names.push_back("divB");
  Csr<S> sl = multiplyOps(o, names);                                       // :208-220
  scaleInPlace(sl, S(-1.0));
  o["scaLapl"] = sl;
  if (pec && mag) o["mRhs"] = multiplyOps(o, {"mu", "dmA"});               // :224-230
  else if (pec) o["mRhs"] = o["dmA"];                                      // :233-236
  else if (mag) o["mRhs"] = o["mu"];                                       // :237-238
  else o["mRhs"] = diagonal<S>(UVec<S>(B_.gids.size(), S(1.0)));    // :240-241
  return o;
}

template <typename S>
UVec<int64_t> YeeAssembler<S>::ownedGids(const YeeField& F) const {
  UVec<int64_t> out;
  for (size_t r = 0; r < F.gids.size(); ++r) if (F.owned[r]) out.push_back(F.gids[r]);
  return out;
}

template <typename S>
Csr<S> YeeAssembler<S>::ownedRows(const Csr<S>& m, const YeeField& rowF) const {
  Csr<S> o;
  o.ncols = m.ncols;
  std::vector<int64_t> rows;
  for (int64_t r = 0; r < m.nrows; ++r) if (rowF.owned[r]) rows.push_back(r);
  o.nrows = (int64_t)rows.size();
  o.rowptr.resize(o.nrows + 1);
  o.rowptr[0] = 0;
  for (int64_t i = 0; i < o.nrows; ++i) o.rowptr[i + 1] = o.rowptr[i] + (m.rowptr[rows[i] + 1] - m.rowptr[rows[i]]);
  o.col.resize(o.rowptr[o.nrows]);
  o.val.resize(o.rowptr[o.nrows]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < o.nrows; ++i) {
    std::copy(m.col.begin() + m.rowptr[rows[i]], m.col.begin() + m.rowptr[rows[i] + 1], o.col.begin() + o.rowptr[i]);
    std::copy(m.val.begin() + m.rowptr[rows[i]], m.val.begin() + m.rowptr[rows[i] + 1], o.val.begin() + o.rowptr[i]);
  }
  return o;
}

template class YeeAssembler<double>;
template class YeeAssembler<std::complex<double>>;

}  // namespace mx
