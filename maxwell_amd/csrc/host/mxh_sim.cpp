// C ABI over the host-side assembler (include/mxh.h, mxh_sim_*).
#include <mutex>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>

#include "mx_yee.hpp"
#include "mxh.h"
#include "mxh_internal.hpp"

static thread_local std::string g_err;

void mxh_set_error(const std::string& s) { g_err = s; }
extern "C" const char* mxh_last_error(void) { return g_err.c_str(); }

namespace {

std::vector<double> numbers(const std::string& s) {
  std::vector<double> v;
  std::string t = s;
  for (char& ch : t) if (ch == ',' || ch == ';') ch = ' ';
  std::istringstream is(t);
  double x;
  while (is >> x) v.push_back(x);
  return v;
}

std::shared_ptr<const mx::Shape> parsePrimitive(const std::string& tok) {
  const size_t lp = tok.find('('), rp = tok.rfind(')');
  if (lp == std::string::npos || rp == std::string::npos) throw std::runtime_error("bad shape token '" + tok + "'");
  std::string name = tok.substr(0, lp);
  name.erase(0, name.find_first_not_of(" \t"));
  name.erase(name.find_last_not_of(" \t") + 1);
  const std::vector<double> a = numbers(tok.substr(lp + 1, rp - lp - 1));
  using mx::Vec3;
  if (name == "cylinder" && a.size() == 7)
    return std::make_shared<mx::Cylinder>(a[0], Vec3(a[1], a[2], a[3]), Vec3(a[4], a[5], a[6]));
  if (name == "slab" && a.size() == 7)
    return std::make_shared<mx::Slab>(a[0], Vec3(a[1], a[2], a[3]), Vec3(a[4], a[5], a[6]));
  if (name == "sphere" && a.size() == 4)
    return std::make_shared<mx::Sphere>(a[0], Vec3(a[1], a[2], a[3]));
  if (name == "halfspace" && a.size() == 6)
    return std::make_shared<mx::HalfSpace>(Vec3(a[3], a[4], a[5]), Vec3(a[0], a[1], a[2]));
  if (name == "ellipsoid" && a.size() == 6)
    return std::make_shared<mx::Ellipsoid>(Vec3(a[0], a[1], a[2]), Vec3(a[3], a[4], a[5]));
  if (name == "torus" && a.size() == 8)
    return std::make_shared<mx::Torus>(a[0], a[1], Vec3(a[2], a[3], a[4]), Vec3(a[5], a[6], a[7]));
  if (name == "cone" && a.size() == 7)
    return std::make_shared<mx::Cone>(a[0], Vec3(a[1], a[2], a[3]), Vec3(a[4], a[5], a[6]));
  throw std::runtime_error("unknown shape or wrong argument count in '" + tok + "'");
}

}  // namespace

std::shared_ptr<const mx::Shape> mxh_parse_shape(const char* spec) {
  if (!spec) return nullptr;
  std::string s(spec);
  if (s.find_first_not_of(" \t\n") == std::string::npos) return nullptr;
  std::vector<std::shared_ptr<const mx::Shape>> parts;
  size_t pos = 0;
  while (pos <= s.size()) {
    size_t amp = s.find('&', pos);
    if (amp == std::string::npos) amp = s.size();
    parts.push_back(parsePrimitive(s.substr(pos, amp - pos)));
    pos = amp + 1;
  }
  if (parts.size() == 1) return parts[0];
  auto inter = std::make_shared<mx::ShapeIntersection>();
  for (auto& p : parts) inter->add(p);
  return inter;
}

// "e11,e22,e33[,o0,o1,o2] [tan=t] : shape-spec | ..." -> dielectrics (t: loss tangent, eps_ii (1 + i t)); the off-diagonal triple follows the deck's
// `eps off diag`: eps(1,2) = o0, eps(0,2) = o1, eps(0,1) = o2 (src/MxProblem.cpp:516-518)
std::vector<mx::YeeDielectric> mxh_parse_dielectrics(const char* spec) {
  std::vector<mx::YeeDielectric> out;
  if (!spec) return out;
  std::string s(spec);
  size_t pos = 0;
  while (pos <= s.size()) {
    size_t bar = s.find('|', pos);
    if (bar == std::string::npos) bar = s.size();
    const std::string item = s.substr(pos, bar - pos);
    pos = bar + 1;
    if (item.find_first_not_of(" \t\n") == std::string::npos) continue;
    const size_t colon = item.find(':');
    if (colon == std::string::npos) throw std::runtime_error("dielectric '" + item + "': expected 'eps : shape'");
    std::string head = item.substr(0, colon);
    if (head.find("mu") != std::string::npos) continue;                    // a magnetic material: mxh_parse_mus
    double losstan = 0.0;                                                   // optional "tan=<loss tangent>"
    const size_t tpos = head.find("tan=");
    if (tpos != std::string::npos) { losstan = atof(head.c_str() + tpos + 4); head = head.substr(0, tpos); }
    const std::vector<double> e = numbers(head);
    if (e.size() != 3 && e.size() != 6) throw std::runtime_error("dielectric '" + item + "': 3 or 6 permittivity numbers");
    mx::YeeDielectric d;
    d.name = "dielectric" + std::to_string(out.size());                 // src/MxProblem.cpp:493-494
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) d.eps[i][j] = 0.0;
    for (int i = 0; i < 3; ++i) { d.eps[i][i] = e[(size_t)i]; d.epsIm[i][i] = e[(size_t)i] * losstan; }
    if (e.size() == 6) {
      d.eps[0][1] = d.eps[1][0] = e[5];
      d.eps[0][2] = d.eps[2][0] = e[4];
      d.eps[1][2] = d.eps[2][1] = e[3];
    }
    d.shape = mxh_parse_shape(item.substr(colon + 1).c_str());
    if (!d.shape) throw std::runtime_error("dielectric '" + item + "': empty shape");
    out.push_back(d);
  }
  return out;
}

// "mu m11,m22,m33[,o0,o1,o2] : shape-spec" items of the same list -> magnetic materials (<Mu>, src/MxProblem.cpp:515-556)
std::vector<mx::YeeMu> mxh_parse_mus(const char* spec) {
  std::vector<mx::YeeMu> out;
  if (!spec) return out;
  std::string s(spec);
  size_t pos = 0;
  while (pos <= s.size()) {
    size_t bar = s.find('|', pos);
    if (bar == std::string::npos) bar = s.size();
    const std::string item = s.substr(pos, bar - pos);
    pos = bar + 1;
    const size_t colon = item.find(':');
    if (colon == std::string::npos) continue;
    std::string head = item.substr(0, colon);
    const size_t mpos = head.find("mu");
    if (mpos == std::string::npos) continue;
    const std::vector<double> e = numbers(head.substr(mpos + 2));
    if (e.size() != 3 && e.size() != 6) throw std::runtime_error("mu object '" + item + "': 3 or 6 permeability numbers");
    mx::YeeMu m;
    m.name = "mu" + std::to_string(out.size());                            // src/MxProblem.cpp:522-523
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m.mu[i][j] = 0.0;
    for (int i = 0; i < 3; ++i) m.mu[i][i] = e[(size_t)i];
    if (e.size() == 6) {                                                   // src/MxProblem.cpp:549-551
      m.mu[0][1] = m.mu[1][0] = e[5];
      m.mu[0][2] = m.mu[2][0] = e[4];
      m.mu[1][2] = m.mu[2][1] = e[3];
    }
    m.shape = mxh_parse_shape(item.substr(colon + 1).c_str());
    if (!m.shape) throw std::runtime_error("mu object '" + item + "': empty shape");
    out.push_back(m);
  }
  return out;
}

extern "C" int mxh_sim_create(const int n[3], const double origin[3], const double size[3],
                              const char* const lower[3], const char* const upper[3],
                              const char* pec_shape, int strict, mxh_sim** out) {
  return mxh_sim_create_ex(n, origin, size, lower, upper, pec_shape, nullptr, 0.0, strict, nullptr, nullptr, out);
}

extern "C" int mxh_sim_create_box(const int n[3], const double origin[3], const double size[3],
                                  const char* const lower[3], const char* const upper[3],
                                  const char* pec_shape, int strict, const int box_lo[3],
                                  const int box_hi[3], mxh_sim** out) {
  return mxh_sim_create_ex(n, origin, size, lower, upper, pec_shape, nullptr, 0.0, strict, box_lo, box_hi, out);
}

extern "C" int mxh_sim_create_ex(const int n[3], const double origin[3], const double size[3],
                                 const char* const lower[3], const char* const upper[3],
                                 const char* pec_shape, const char* dielectrics, double dm_frac, int strict,
                                 const int box_lo[3], const int box_hi[3], mxh_sim** out) {
  try {
    if (!n || !origin || !size || !out) throw std::runtime_error("mxh_sim_create: NULL argument");
    mx::YeeConfig cfg;
    for (int i = 0; i < 3; ++i) {
      if (n[i] < 1) throw std::runtime_error("mxh_sim_create: resolution must be >= 1");
      cfg.n[i] = n[i]; cfg.origin[i] = origin[i]; cfg.size[i] = size[i];
      cfg.lower[i] = mx::parseBc(lower ? lower[i] : "periodic");
      cfg.upper[i] = mx::parseBc(upper ? upper[i] : "periodic");
    }
    cfg.pec = mxh_parse_shape(pec_shape);
    cfg.dielectrics = mxh_parse_dielectrics(dielectrics);
    cfg.mus = mxh_parse_mus(dielectrics);
    cfg.dmFrac = dm_frac;
    cfg.strict = strict != 0;
    if (box_lo && box_hi)
      for (int i = 0; i < 3; ++i) { cfg.boxLo[i] = box_lo[i]; cfg.boxHi[i] = box_hi[i]; }
    const auto t0 = std::chrono::steady_clock::now();
    mxh_sim* s = new mxh_sim();
    s->cfg = cfg;
    s->asmb.reset(new mx::YeeAssembler<double>(cfg));
    // Single rank: only the single operators are assembled here; MxMagWaveOp forms the chain of products on the GPU
    // (mxg_csr_multiply / add / purge, bit-identical to the host chain).  A host copy of a product is made the first
    // time somebody asks for one (mxh_sim_op and friends).  MXH_DEVICE_CHAIN=0: the chain is formed on the host, here.
    // A box of a decomposed grid works the same way on its box + halo window: the solver forms the window chain on
    // this rank's GPU when it is created (mxh_sim_device_chain); anybody who asks for a product before that gets the
    // host chain (mxh_sim_ensure_chain).
    const char* dc = getenv("MXH_DEVICE_CHAIN");
    s->chain_ready = dc && atoi(dc) == 0;
    s->ops = s->chain_ready ? s->asmb->waveOps() : s->asmb->singleOps();
    s->mask = s->asmb->usedMask();
    s->setup_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *out = s;
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}

extern "C" int mxh_sim_destroy(mxh_sim* s) {
  delete s;
  return 0;
}

static int exportCsr(const mx::Csr<double>& m, int64_t* nrows, int64_t* ncols, int64_t* nnz,
                     const int64_t** rowptr, const int32_t** colidx, const double** vals) {
  if (nrows) *nrows = m.nrows;
  if (ncols) *ncols = m.ncols;
  if (nnz) *nnz = m.nnz();
  if (rowptr) *rowptr = m.rowptr.data();
  if (colidx) *colidx = m.col.data();
  if (vals) *vals = m.val.data();
  return 0;
}

// the host chain of a sim created in device-chain mode, on first demand
void mxh_sim_ensure_chain(mxh_sim* s) {
  static std::mutex mu;                       // rank threads of an in-process world may share a sim
  std::lock_guard<std::mutex> lk(mu);
  if (s->chain_ready) return;
  // from the primitives the sim already holds (device-chain mode keeps singleOps()): nothing is assembled twice
  std::map<std::string, mx::Csr<double> > singles;
  for (const char* nm : {"curlE", "curlB", "divB", "gradPsi", "dmA", "dmL", "dmVInv", "invEps", "invEpsVolAve", "mu", "invMuVolAve"})
    if (s->ops.count(nm)) singles[nm] = s->ops.at(nm);
  std::map<std::string, mx::Csr<double> > full = singles.count("curlE") ? s->asmb->waveOps(std::move(singles)) : s->asmb->waveOps();
  for (auto& kv : full) if (!s->ops.count(kv.first)) s->ops[kv.first] = std::move(kv.second);
  s->chain_ready = true;
}

// The operator chain of a sim that holds only its single operators, formed on the GPU of `ctx` with mxg_csr_multiply /
// add / purge (bit-identical to the host chain, tests/test_spalg_gpu.py) and read back into sim->ops: the order of
// YeeAssembler::waveOps (src/MxMagWaveOp.cpp:138-245).  This is how a rank of a decomposed grid gets the products on
// its box + halo window (row f2 on config 4), for real and complex problems.
namespace {
struct DevMat {
  mxg_csr* h = nullptr;
  DevMat() {}
  DevMat(const DevMat&) = delete;
  DevMat& operator=(const DevMat&) = delete;
  ~DevMat() { if (h) mxg_csr_destroy(h); }
};
void dev_check(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + mxg_last_error());
}
template <typename S>
mx::Csr<S> dev_download(const mxg_csr* h) {
  int64_t nr = 0, nc = 0, ng = 0, nz = 0;
  int cx = 0;
  dev_check(mxg_csr_info(h, &nr, &nc, &ng, &nz, &cx), "mxg_csr_info");
  if ((cx != 0) != (sizeof(S) == 2 * sizeof(double))) throw std::runtime_error("device chain: scalar type mismatch");
  mx::Csr<S> m;
  m.nrows = nr; m.ncols = nc;
  m.rowptr.resize(nr + 1); m.col.resize(nz); m.val.resize(nz);
  dev_check(mxg_csr_download(h, m.rowptr.data(), m.col.data(), reinterpret_cast<double*>(m.val.data())), "mxg_csr_download");
  return m;
}
}  // namespace

// the chain for either scalar type: `ops` holds the single operators on entry, the products are added to it
template <typename S>
void mxh_device_chain(mxg_ctx* ctx, const mx::YeeAssembler<S>& a, std::map<std::string, mx::Csr<S> >& ops) {
  const int cx = sizeof(S) == 2 * sizeof(double) ? 1 : 0;
  const bool pec = a.hasPec(), diel = a.hasDielectric(), mag = a.hasMu();
  std::map<std::string, std::unique_ptr<DevMat> > dev;
  auto get = [&](const std::string& nm) -> mxg_csr* {            // upload a single operator on first use
    auto it = dev.find(nm);
    if (it == dev.end()) {
      const mx::Csr<S>& m = ops.at(nm);
      std::unique_ptr<DevMat> d(new DevMat());
      dev_check(mxg_csr_create(ctx, m.nrows, m.ncols, 0, m.rowptr.data(), m.col.data(), reinterpret_cast<const double*>(m.val.data()), cx,
                               nullptr, &d->h), "mxg_csr_create");
      it = dev.emplace(nm, std::move(d)).first;
    }
    return it->second->h;
  };
  auto chain = [&](const std::vector<std::string>& names, const std::string& out) {   // massiveCrsMultiply, src/MxUtil.cpp:318-371
    std::unique_ptr<DevMat> res;
    const mxg_csr* cur = get(names[0]);
    for (size_t i = 1; i < names.size(); ++i) {
      std::unique_ptr<DevMat> nx(new DevMat());
      dev_check(mxg_csr_multiply(ctx, get(names[i]), cur, &nx->h), "mxg_csr_multiply");
      res = std::move(nx);
      cur = res->h;
    }
    if (!res) throw std::runtime_error("mxh_sim_device_chain: a chain needs two factors");
    dev[out] = std::move(res);
  };
  auto opt = [&](std::vector<std::string>& v, const char* nm) { if (pec) v.push_back(nm); };
  std::vector<std::string> names;
  names = {"curlB"}; if (diel) names.push_back("invEps"); opt(names, "dmL"); names.push_back("curlE");
  chain(names, "curlCurl");
  names.clear(); if (mag) names.push_back("mu"); opt(names, "dmA"); names.push_back("divB"); opt(names, "dmVInv");
  if (mag) names.push_back("invMuVolAve");
  if (diel) names.push_back("invEpsVolAve");
  if (mag) names.push_back("invMuVolAve");
  names.push_back("gradPsi"); if (mag) names.push_back("mu"); opt(names, "dmA");
  chain(names, "gradDiv");
  {
    DevMat sum;
    dev_check(mxg_csr_add(ctx, dev.at("curlCurl")->h, 1.0, dev.at("gradDiv")->h, -1.0, &sum.h), "mxg_csr_add");
    std::unique_ptr<DevMat> vl(new DevMat());
    dev_check(mxg_csr_purge(ctx, sum.h, 1.0e-12, &vl->h), "mxg_csr_purge");
    dev["vecLapl"] = std::move(vl);
  }
  names = {"gradPsi"}; if (mag) names.push_back("mu"); opt(names, "dmA"); names.push_back("divB");
  chain(names, "scaLaplPos");
  for (const char* nm : {"curlCurl", "gradDiv", "vecLapl"}) ops[nm] = dev_download<S>(dev.at(nm)->h);
  {
    mx::Csr<S> sl = dev_download<S>(dev.at("scaLaplPos")->h);
    for (auto& v : sl.val) v = v * S(-1.0);                     // scaleInPlace(sl, -1)
    ops["scaLapl"] = std::move(sl);
  }
  if (pec && mag) { chain({"mu", "dmA"}, "mRhs"); ops["mRhs"] = dev_download<S>(dev.at("mRhs")->h); }
  else if (pec) ops["mRhs"] = ops.at("dmA");
  else if (mag) ops["mRhs"] = ops.at("mu");
  else ops["mRhs"] = mx::diagonal<S>(mx::UVec<S>(a.bfield().gids.size(), S(1.0)));
}
template void mxh_device_chain<double>(mxg_ctx*, const mx::YeeAssembler<double>&, std::map<std::string, mx::Csr<double> >&);
template void mxh_device_chain<std::complex<double> >(mxg_ctx*, const mx::YeeAssembler<std::complex<double> >&,
                                                      std::map<std::string, mx::Csr<std::complex<double> > >&);

extern "C" int mxh_sim_device_chain(mxg_ctx* ctx, mxh_sim* s) {
  try {
    if (!ctx || !s) throw std::runtime_error("mxh_sim_device_chain: NULL argument");
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (s->chain_ready) return 0;
    mxh_device_chain<double>(ctx, *s->asmb, s->ops);
    s->chain_ready = true;
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}

extern "C" int mxh_sim_op(mxh_sim* s, const char* name, int64_t* nrows, int64_t* ncols, int64_t* nnz,
                          const int64_t** rowptr, const int32_t** colidx, const double** vals) {
  if (!s->ops.count(name)) mxh_sim_ensure_chain(s);
  auto it = s->ops.find(name);
  if (it == s->ops.end()) {
    mxh_set_error(std::string("MxEMOps::getOp: operator '") + name + "' not found.");
    return 1;
  }
  return exportCsr(it->second, nrows, ncols, nnz, rowptr, colidx, vals);
}

const mx::Csr<double>& mxh_sim_shifted_matrix(mxh_sim* s, double shift) {
  auto it = s->shifted.find(shift);
  mxh_sim_ensure_chain(s);
  if (it == s->shifted.end())
    it = s->shifted.emplace(shift, mx::add(s->ops.at("vecLapl"), 1.0, s->ops.at("mRhs"), -shift)).first;
  return it->second;
}

extern "C" int mxh_sim_shifted(mxh_sim* s, double shift, int64_t* nrows, int64_t* nnz,
                               const int64_t** rowptr, const int32_t** colidx, const double** vals) {
  try {
    return exportCsr(mxh_sim_shifted_matrix(s, shift), nrows, nullptr, nnz, rowptr, colidx, vals);
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}

extern "C" int mxh_sim_map(mxh_sim* s, const char* field, int64_t* n, const int64_t** gids) {
  const std::string f(field);
  const mx::YeeField* F = nullptr;
  if (f == "bfield") F = &s->asmb->bfield();
  else if (f == "efield") F = &s->asmb->efield();
  else if (f == "psifield") F = &s->asmb->psifield();
  if (!F) {
    mxh_set_error("MxEMSim::getField: field '" + f + "' not found.");
    return 1;
  }
  if (n) *n = (int64_t)F->gids.size();
  if (gids) *gids = F->gids.data();
  return 0;
}

extern "C" int mxh_sim_used_mask(mxh_sim* s, int64_t* n, const double** mask) {
  if (n) *n = (int64_t)s->mask.size();
  if (mask) *mask = s->mask.data();
  return 0;
}

extern "C" double mxh_sim_setup_seconds(const mxh_sim* s) { return s->setup_seconds; }

// ---------------------------------------------------------------------------------------
// domain decomposition (MxGrid::partition / getGridDomain, src/MxGrid.cpp:100-193;
// MxUtil::unorderedFactorizations / permutations, src/MxUtil.hpp:199-286)
// ---------------------------------------------------------------------------------------
namespace {

std::vector<std::vector<size_t>> unorderedFactorizations(size_t val, size_t maxFac) {
  std::vector<std::vector<size_t>> res;
  std::vector<size_t> div;
  for (size_t i = 2; i <= val; ++i) if (val % i == 0) div.push_back(i);
  if (div.empty()) { res.push_back(div); return res; }
  for (size_t d : div) {
    if (d > maxFac) continue;
    for (auto& sub : unorderedFactorizations(val / d, d)) {
      std::vector<size_t> fac(1, d);
      fac.insert(fac.end(), sub.begin(), sub.end());
      res.push_back(fac);
    }
  }
  return res;
}

std::vector<std::vector<size_t>> permutations(const std::vector<size_t>& v) {
  std::vector<std::vector<size_t>> res;
  const size_t size = v.size();
  if (size == 1) { res.push_back(v); return res; }
  std::vector<size_t> done;
  for (size_t i = 0; i < size; ++i) {
    const size_t cur = v[i];
    if (std::find(done.begin(), done.end(), cur) != done.end()) continue;
    std::vector<size_t> subset;
    for (size_t j = i + 1; j < i + size; ++j) subset.push_back(v[j % size]);
    for (auto& sp : permutations(subset)) {
      std::vector<size_t> perm(1, cur);
      perm.insert(perm.end(), sp.begin(), sp.end());
      res.push_back(perm);
    }
    done.push_back(cur);
  }
  return res;
}

void procGridOf(const int n[3], int nranks, int pg[3]) {
  std::vector<std::vector<size_t>> all = nranks == 1 ? std::vector<std::vector<size_t>>(1, std::vector<size_t>(1, 1))
                                                     : unorderedFactorizations((size_t)nranks, (size_t)nranks);
  std::vector<std::vector<size_t>> decomps;
  for (auto fac : all) {
    if (fac.size() > 3) continue;
    while (fac.size() < 3) fac.push_back(1);
    for (auto& p : permutations(fac)) decomps.push_back(p);
  }
  double minRatio = 0;
  size_t best = 0;
  for (size_t i = 0; i < decomps.size(); ++i) {
    double ratio = 0;
    for (int j = 0; j < 3; ++j) ratio += double(decomps[i][j]) / double(n[j]);
    if (i == 0 || ratio < minRatio) { minRatio = ratio; best = i; }
  }
  for (int j = 0; j < 3; ++j) pg[j] = (int)decomps[best][j];
}

}  // namespace

extern "C" int mxh_partition(const int n[3], int nranks, int rank, int proc_grid[3], int box_lo[3], int box_hi[3]) {
  if (nranks < 1 || rank < 0 || rank >= nranks) { mxh_set_error("mxh_partition: bad rank"); return 1; }
  int pg[3];
  procGridOf(n, nranks, pg);
  int factor = 1;
  for (int i = 2; i >= 0; --i) {                                  // src/MxGrid.cpp:171-193
    const int stdRes = int(std::floor(double(n[i]) / double(pg[i]) + 0.5));
    const int proci = (rank / factor) % pg[i];
    box_lo[i] = proci * stdRes;
    box_hi[i] = (proci == pg[i] - 1) ? n[i] + 1 : box_lo[i] + stdRes;
    factor *= pg[i];
    proc_grid[i] = pg[i];
  }
  return 0;
}

extern "C" int mxh_owner_of_cell(const int n[3], const int pg[3], const int cell[3]) {
  int pid = 0, factor = 1;
  for (int i = 2; i >= 0; --i) {
    const int stdRes = int(std::floor(double(n[i]) / double(pg[i]) + 0.5));
    int proci = stdRes > 0 ? cell[i] / stdRes : 0;
    if (proci > pg[i] - 1) proci = pg[i] - 1;
    pid += proci * factor;
    factor *= pg[i];
  }
  return pid;
}

extern "C" int mxh_sim_dist_op(mxh_sim* s, const char* name, double shift, int64_t* nrows, int64_t* nnz,
                               const int64_t** rowptr, const int64_t** col_gids, const double** vals,
                               const int64_t** row_gids, int64_t* n_owned_cols,
                               const int64_t** col_field_owned_gids, int* col_ncomp) {
  try {
    const std::string nm(name);
    const mx::YeeAssembler<double>& a = *s->asmb;
    const mx::YeeField &B = a.bfield(), &E = a.efield(), &P = a.psifield();
    const mx::YeeField* rowF = &B; const mx::YeeField* colF = &B;
    if (nm == "scaLapl" || nm == "dmVInv") { rowF = &P; colF = &P; }
    else if (nm == "divB") { rowF = &P; colF = &B; }
    else if (nm == "gradPsi") { rowF = &B; colF = &P; }
    else if (nm == "curlE") { rowF = &B; colF = &E; }
    else if (nm == "curlB") { rowF = &E; colF = &B; }
    else if (nm == "dmL") { rowF = &E; colF = &E; }
    const mx::Csr<double>* m;
    if (nm == "shifted") m = &mxh_sim_shifted_matrix(s, shift);
    else {
      if (!s->ops.count(nm)) mxh_sim_ensure_chain(s);
      auto it = s->ops.find(nm);
      if (it == s->ops.end()) throw std::runtime_error("MxEMOps::getOp: operator '" + nm + "' not found.");
      m = &it->second;
    }
    char key[96];
    std::snprintf(key, sizeof(key), "%s@%.17g", name, nm == "shifted" ? shift : 0.0);
    mxh_sim::DistOp& d = s->dist[key];
    mx::Csr<double> o = a.ownedRows(*m, *rowF);
    d.rowptr = o.rowptr;
    d.vals = o.val;
    d.colgid.resize(o.col.size());
    for (size_t q = 0; q < o.col.size(); ++q) d.colgid[q] = colF->gids[o.col[q]];
    d.rowgid = a.ownedGids(*rowF);
    d.colowned = a.ownedGids(*colF);
    if (nrows) *nrows = o.nrows;
    if (nnz) *nnz = o.nnz();
    if (rowptr) *rowptr = d.rowptr.data();
    if (col_gids) *col_gids = d.colgid.data();
    if (vals) *vals = d.vals.data();
    if (row_gids) *row_gids = d.rowgid.data();
    if (n_owned_cols) *n_owned_cols = (int64_t)d.colowned.size();
    if (col_field_owned_gids) *col_field_owned_gids = d.colowned.data();
    if (col_ncomp) *col_ncomp = colF->ncomp;
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}
