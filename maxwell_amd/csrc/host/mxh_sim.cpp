// C ABI over the host-side assembler (include/mxh.h, mxh_sim_*).
#include <chrono>
#include <cstdarg>
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

extern "C" int mxh_sim_create(const int n[3], const double origin[3], const double size[3],
                              const char* const lower[3], const char* const upper[3],
                              const char* pec_shape, int strict, mxh_sim** out) {
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
    cfg.strict = strict != 0;
    const auto t0 = std::chrono::steady_clock::now();
    mxh_sim* s = new mxh_sim();
    s->cfg = cfg;
    s->asmb.reset(new mx::YeeAssembler<double>(cfg));
    s->ops = s->asmb->waveOps();
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

extern "C" int mxh_sim_op(mxh_sim* s, const char* name, int64_t* nrows, int64_t* ncols, int64_t* nnz,
                          const int64_t** rowptr, const int32_t** colidx, const double** vals) {
  auto it = s->ops.find(name);
  if (it == s->ops.end()) {
    mxh_set_error(std::string("MxEMOps::getOp: operator '") + name + "' not found.");
    return 1;
  }
  return exportCsr(it->second, nrows, ncols, nnz, rowptr, colidx, vals);
}

const mx::Csr<double>& mxh_sim_shifted_matrix(mxh_sim* s, double shift) {
  auto it = s->shifted.find(shift);
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
