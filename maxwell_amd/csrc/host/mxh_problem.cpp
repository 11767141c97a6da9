// C ABI over the `.mx` front end (include/mxh.h, mxh_problem_*): MxProblem -> mxh_sim / mxh_solver.
#include <chrono>
#include <cstring>

#include "mx/MxProblem.h"
#include "mxh.h"
#include "mxh_internal.hpp"
#include "mxh_solver_internal.hpp"

struct mxh_problem {
  std::unique_ptr<MxProblem<3> > prob;   // parsed without a communicator: no device needed
  std::unique_ptr<mxh_sim> sim;
  // complex decks (Bloch phase shifts): operators assembled in complex arithmetic, exported interleaved
  std::map<std::string, mx::Csr<std::complex<double> > > cops;
};

namespace {

// MxProblem reports progress on stdout like the reference; library callers get it on demand only
struct QuietCout {
  std::streambuf* old;
  std::ostringstream sink;
  explicit QuietCout(bool quiet) : old(nullptr) { if (quiet) old = std::cout.rdbuf(sink.rdbuf()); }
  ~QuietCout() { if (old) std::cout.rdbuf(old); }
};

int finish(std::unique_ptr<mxh_problem>& p, mxh_problem** out) {
  *out = p.release();
  return 0;
}

}  // namespace

extern "C" int mxh_problem_parse(const char* mx_text, int strict, mxh_problem** out) {
  try {
    if (!mx_text || !out) throw std::runtime_error("mxh_problem_parse: NULL argument");
    QuietCout q(getenv("MXH_VERBOSE") == nullptr);
    std::unique_ptr<mxh_problem> p(new mxh_problem());
    p->prob.reset(new MxProblem<3>(MxUtil::XML::Node::parse(mx_text), RCP<MxComm>(), strict != 0));
    return finish(p, out);
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}

extern "C" int mxh_problem_load(const char* path, int strict, mxh_problem** out) {
  try {
    if (!path || !out) throw std::runtime_error("mxh_problem_load: NULL argument");
    QuietCout q(getenv("MXH_VERBOSE") == nullptr);
    std::unique_ptr<mxh_problem> p(new mxh_problem());
    p->prob.reset(new MxProblem<3>(MxUtil::XML::Node::parseFile(path), RCP<MxComm>(), strict != 0));
    return finish(p, out);
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}

extern "C" int mxh_problem_destroy(mxh_problem* p) {
  delete p;
  return 0;
}

extern "C" int mxh_problem_param(mxh_problem* p, const char* key, char* buf, size_t buflen) {
  if (!p || !key || !buf || buflen == 0) { mxh_set_error("mxh_problem_param: NULL argument"); return 1; }
  const Teuchos::ParameterList& pl = p->prob->getParameters();
  if (!pl.isParameter(key)) { mxh_set_error(std::string("parameter '") + key + "' is not set"); return 1; }
  const std::string v = pl.get(key, "");
  std::strncpy(buf, v.c_str(), buflen - 1);
  buf[buflen - 1] = '\0';
  return 0;
}

extern "C" int mxh_problem_grid(mxh_problem* p, int n[3], double origin[3], double size[3], int lower[3],
                                int upper[3], double* dm_frac, int* has_pec) {
  const mx::YeeConfig& c = p->prob->getConfig();
  for (int i = 0; i < 3; ++i) {
    if (n) n[i] = c.n[i];
    if (origin) origin[i] = c.origin[i];
    if (size) size[i] = c.size[i];
    if (lower) lower[i] = (int)c.lower[i];
    if (upper) upper[i] = (int)c.upper[i];
  }
  if (dm_frac) *dm_frac = c.dmFrac;
  if (has_pec) *has_pec = c.pec ? 1 : 0;
  return 0;
}

extern "C" int mxh_problem_pec_func(mxh_problem* p, const double xyz[3], double* f) {
  const mx::YeeConfig& c = p->prob->getConfig();
  if (!c.pec) { mxh_set_error("mxh_problem_pec_func: the deck has no <PEC> block"); return 1; }
  *f = c.pec->func(mx::Vec3(xyz[0], xyz[1], xyz[2]));
  return 0;
}

extern "C" int mxh_problem_is_complex(mxh_problem* p) { return p->prob->isComplex() ? 1 : 0; }

extern "C" int mxh_problem_op(mxh_problem* p, const char* name, int64_t* nrows, int64_t* ncols, int64_t* nnz,
                              const int64_t** rowptr, const int32_t** colidx, const double** vals, int* is_complex) {
  try {
    if (!p->prob->isComplex()) {
      mxh_sim* sim = nullptr;
      if (mxh_problem_sim(p, &sim) != 0) return 1;
      if (is_complex) *is_complex = 0;
      return mxh_sim_op(sim, name, nrows, ncols, nnz, rowptr, colidx, vals);
    }
    if (p->cops.empty()) {
      mx::YeeAssembler<std::complex<double> > a(p->prob->getConfig());
      p->cops = a.waveOps();
    }
    auto it = p->cops.find(name);
    if (it == p->cops.end()) throw std::runtime_error(std::string("MxEMOps::getOp: operator '") + name + "' not found.");
    const mx::Csr<std::complex<double> >& m = it->second;
    if (nrows) *nrows = m.nrows;
    if (ncols) *ncols = m.ncols;
    if (nnz) *nnz = m.nnz();
    if (rowptr) *rowptr = m.rowptr.data();
    if (colidx) *colidx = m.col.data();
    if (vals) *vals = reinterpret_cast<const double*>(m.val.data());
    if (is_complex) *is_complex = 1;
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}

extern "C" int mxh_problem_sim(mxh_problem* p, mxh_sim** out) {
  try {
    if (p->prob->isComplex())
      throw std::runtime_error("mxh_problem_sim: the deck is complex (phase shifts); use mxh_problem_op / mxh_problem_solver");
    if (!p->sim) {
      const auto t0 = std::chrono::steady_clock::now();
      std::unique_ptr<mxh_sim> s(new mxh_sim());
      s->cfg = p->prob->getConfig();
      s->asmb.reset(new mx::YeeAssembler<double>(s->cfg));
      s->ops = s->asmb->waveOps();
      s->mask = s->asmb->usedMask();
      s->setup_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      p->sim = std::move(s);
    }
    *out = p->sim.get();
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}

extern "C" int mxh_problem_solver(mxg_ctx* ctx, mxh_problem* p, int verbose, mxh_solver** out) {
  try {
    if (!ctx || !p || !out) throw std::runtime_error("mxh_problem_solver: NULL argument");
    Teuchos::ParameterList pl = p->prob->getParameters();
    if (verbose >= 0) pl.set("eigensolver : output", verbose);
    std::unique_ptr<mxh_solver> s(new mxh_solver());
    mxh_solver_default_params(&s->params);
    s->linType = pl.get("linear solver : type", "gmres");
    s->precType = pl.get("linear solver : prec type", "none");
    if (p->prob->isComplex()) {
      mxh_solver_build_complex(*s, ctx, p->prob->getConfig(), pl);
    } else if (mxg_ctx_nranks(ctx) > 1) {
      // decomposed grid: this rank's MxGridDomain box of the deck's grid (src/MxGrid.cpp:171-193)
      std::shared_ptr<mxh_sim> box(new mxh_sim());
      box->cfg = p->prob->getConfig();
      int pg[3];
      if (mxh_partition(box->cfg.n, mxg_ctx_nranks(ctx), mxg_ctx_rank(ctx), pg, box->cfg.boxLo, box->cfg.boxHi) != 0) return 1;
      box->asmb.reset(new mx::YeeAssembler<double>(box->cfg));
      // the window chain on this rank's GPU (MXH_DEVICE_CHAIN=0: on the host), as mxh_solver_create does
      box->ops = box->asmb->singleOps();
      box->chain_ready = false;
      {
        const char* dc = getenv("MXH_DEVICE_CHAIN");
        if (dc && atoi(dc) == 0) mxh_sim_ensure_chain(box.get());
        else if (mxh_sim_device_chain(ctx, box.get()) != 0) return 1;
      }
      box->mask = box->asmb->usedMask();
      s->ownSim = box;
      mxh_solver_build(*s, ctx, box.get(), pl);
    } else {
      mxh_sim* sim = nullptr;
      if (mxh_problem_sim(p, &sim) != 0) return 1;
      mxh_solver_build(*s, ctx, sim, pl);
    }
    *out = s.release();
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  } catch (int) {
    mxh_set_error(std::string("MxSolver setup failed: ") + mxg_last_error());
    return 1;
  }
}
