// C ABI over the eigen-driver (include/mxh.h, mxh_solver_*): MxSolver + MxMagWaveOp.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>

#include "mx/MxSolver.h"
#include "mxh.h"
#include "mxh_internal.hpp"
#include "mxh_solver_internal.hpp"

extern "C" void mxh_solver_default_params(mxh_solver_params* p) {
  // defaults of src/MxProblem.cpp:26-46
  p->nev = 10; p->basis = 20; p->block_size = 1; p->max_restarts = 100; p->tol = 1.0e-6; p->shift = 0.0;
  p->lin_type = "gmres"; p->lin_basis = 40; p->lin_tol = 1.0e-6; p->prec_type = "none";
  p->verbose = 0; p->seed = 12345;
  p->amg_smoother = "Chebyshev"; p->amg_sweeps = 2;      // src/MxProblem.cpp:56-57
  p->eig_type = "krylov-schur";                          // src/MxProblem.cpp:28
}

void mxh_solver_build(mxh_solver& s, mxg_ctx* ctx, mxh_sim* sim, Teuchos::ParameterList const& pl) {
  s.comm = rcp(new MxComm(ctx));
  s.params.shift = pl.get("eigensolver : shift", 0.0);
  // a box of a decomposed grid: the chain of products on its window is formed on this rank's GPU now (a single-rank
  // sim leaves it to MxMagWaveOp, which never reads the products back)
  if (sim->asmb->isWindowed() && !sim->chain_ready && mxh_sim_device_chain(ctx, sim) != 0)
    throw std::runtime_error(mxh_last_error());
  const bool timing = getenv("MXH_VERBOSE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  s.sim = rcp(new MxEMSim<3>(sim->cfg, s.comm, pl, sim->asmb));
  s.sim->setPrebuiltOps(&sim->ops);     // single operators, or the whole chain when it has been formed
  const auto t1 = std::chrono::steady_clock::now();
  s.solver.reset(new MxSolver<3, double>(s.sim, pl));
  if (timing)
    fprintf(stderr, "[mxh_solver] MxEMSim (grid, fields, maps) %.3f s, MxSolver (operator, preconditioner, eigenproblem) %.3f s\n",
            std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
}

void mxh_solver_build_complex(mxh_solver& s, mxg_ctx* ctx, mx::YeeConfig const& cfg, Teuchos::ParameterList const& pl) {
  s.comm = rcp(new MxComm(ctx));
  s.params.shift = pl.get("eigensolver : shift", 0.0);
  if (s.comm->numProc() > 1) {
    // decomposed grid: this rank's box of the grid, complex window assembly and operator chain
    mx::YeeConfig box = cfg;
    int pg[3];
    if (mxh_partition(box.n, s.comm->numProc(), s.comm->myPID(), pg, box.boxLo, box.boxHi) != 0) throw std::runtime_error(mxh_last_error());
    s.boxAsmC = std::make_shared<mx::YeeAssembler<MxComplex> >(box);
    // the window chain on this rank's GPU (MXH_DEVICE_CHAIN=0: on the host), like the real boxes
    {
      const char* dc = getenv("MXH_DEVICE_CHAIN");
      if (dc && atoi(dc) == 0) s.boxOpsC = s.boxAsmC->waveOps();
      else { s.boxOpsC = s.boxAsmC->singleOps(); mxh_device_chain<MxComplex>(ctx, *s.boxAsmC, s.boxOpsC); }
    }
    s.sim = rcp(new MxEMSim<3>(box, s.comm, pl, nullptr, s.boxAsmC));
    s.sim->setPrebuiltOps(&s.boxOpsC);
  } else {
    s.sim = rcp(new MxEMSim<3>(cfg, s.comm, pl));
  }
  s.csolver.reset(new MxSolver<3, MxComplex>(s.sim, pl));
}

extern "C" int mxh_solver_create(mxg_ctx* ctx, mxh_sim* sim, const mxh_solver_params* p, mxh_solver** out) {
  try {
    if (!ctx || !sim || !p || !out) throw std::runtime_error("mxh_solver_create: NULL argument");
    std::unique_ptr<mxh_solver> s(new mxh_solver());
    s->params = *p;
    s->linType = p->lin_type ? p->lin_type : "gmres";
    s->precType = p->prec_type ? p->prec_type : "none";
    Teuchos::ParameterList pl;
    pl.set("eigensolver : type", p->eig_type ? p->eig_type : "krylov-schur");
    pl.set("eigensolver : nev", p->nev);
    pl.set("eigensolver : basis", p->basis);
    pl.set("eigensolver : block size", p->block_size);
    pl.set("eigensolver : max restarts", p->max_restarts);
    pl.set("eigensolver : tol", p->tol);
    pl.set("eigensolver : shift", p->shift);
    pl.set("eigensolver : spectrum", "LM");
    pl.set("eigensolver : output", p->verbose);
    pl.set("eigensolver : invert", true);
    pl.set("linear solver : type", s->linType.c_str());
    pl.set("linear solver : basis", p->lin_basis);
    pl.set("linear solver : tol", p->lin_tol);
    pl.set("linear solver : prec type", s->precType.c_str());
    s->amgSmoother = p->amg_smoother ? p->amg_smoother : "Chebyshev";
    pl.set("linear solver : amg : smoother", s->amgSmoother.c_str());
    pl.set("linear solver : amg : sweeps", p->amg_sweeps > 0 ? p->amg_sweeps : 2);
    mxh_solver_build(*s, ctx, sim, pl);
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

extern "C" int mxh_solver_destroy(mxh_solver* s) {
  delete s;
  return 0;
}

extern "C" int mxh_solver_solve(mxh_solver* s, int* nconv) {
  try {
    if (s->csolver) { s->csolver->solve(); if (nconv) *nconv = (int)s->csolver->eigenvalues().size(); return 0; }
    s->solver->solve();
    if (nconv) *nconv = (int)s->solver->eigenvalues().size();
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  } catch (int) {
    mxh_set_error(std::string("MxSolver::solve failed: ") + mxg_last_error());
    return 1;
  }
}

extern "C" int mxh_solver_result(mxh_solver* s, int i, double theta[2], double k2[2], double freq[2],
                                 double* residual, double* divergence) {
  const auto& ev = s->csolver ? s->csolver->eigenvalues() : s->solver->eigenvalues();
  if (i < 0 || i >= (int)ev.size()) {
    mxh_set_error("mxh_solver_result: index out of range");
    return 1;
  }
  const std::complex<double> kk = 1.0 / ev[i] + s->params.shift;   // src/MxMagWaveOp.cpp:1171-1179
  if (theta) { theta[0] = ev[i].real(); theta[1] = ev[i].imag(); }
  if (k2) { k2[0] = kk.real(); k2[1] = kk.imag(); }
  const auto& fr = s->csolver ? s->csolver->frequencies() : s->solver->frequencies();
  if (freq) { freq[0] = fr[i].real(); freq[1] = fr[i].imag(); }
  if (residual) *residual = s->csolver ? s->csolver->residuals()[i] : s->solver->residuals()[i];
  if (divergence) *divergence = s->csolver ? s->csolver->divergences()[i] : s->solver->divergences()[i];
  return 0;
}

extern "C" int mxh_solver_stats(mxh_solver* s, int* num_applies, int* vec_lin_iters, int* sca_lin_iters,
                                double* vec_solve_seconds, double* sca_solve_seconds,
                                double* eigensolve_seconds) {
  if (s->csolver) {
    const auto& op = s->csolver->waveOp();
    if (num_applies) *num_applies = op.numApplies();
    if (vec_lin_iters) *vec_lin_iters = op.numVecLinIters();
    if (sca_lin_iters) *sca_lin_iters = op.numScaLinIters();
    if (vec_solve_seconds) *vec_solve_seconds = op.vecSolveTime();
    if (sca_solve_seconds) *sca_solve_seconds = op.scaSolveTime();
    if (eigensolve_seconds) *eigensolve_seconds = s->csolver->eigensolveSeconds();
    return 0;
  }
  const auto& op = s->solver->waveOp();
  if (num_applies) *num_applies = op.numApplies();
  if (vec_lin_iters) *vec_lin_iters = op.numVecLinIters();
  if (sca_lin_iters) *sca_lin_iters = op.numScaLinIters();
  if (vec_solve_seconds) *vec_solve_seconds = op.vecSolveTime();
  if (sca_solve_seconds) *sca_solve_seconds = op.scaSolveTime();
  if (eigensolve_seconds) *eigensolve_seconds = s->solver->eigensolveSeconds();
  return 0;
}

extern "C" int mxh_solver_eigvec(mxh_solver* s, int i, double* out) {
  try {
    if (s->csolver) {     // complex: interleaved (re, im), 2 * rows doubles
      std::vector<MxComplex> col;
      s->csolver->magneticEigenvectors()->getColumn((size_t)i, col);
      std::memcpy(out, col.data(), sizeof(MxComplex) * col.size());
      return 0;
    }
    std::vector<double> col;
    s->solver->magneticEigenvectors()->getColumn((size_t)i, col);
    std::copy(col.begin(), col.end(), out);
    return 0;
  } catch (...) {
    mxh_set_error("mxh_solver_eigvec failed");
    return 1;
  }
}

// ---------------------------------------------------------------------------------------
// Row-partitioned operators of a decomposed grid through the C++ classes (MxEMSim maps of the owned entries,
// MxCrsMatrix with the ghost import agreed between the ranks)
// ---------------------------------------------------------------------------------------
struct mxh_dist_ops {
  RCP<MxComm> comm;
  RCP<MxEMSim<3> > sim;
  RCP<MxEMOps<3, double> > ops;
  mxh_sim* src = nullptr;
  std::map<std::string, RCP<MxCrsMatrix<double> > > made;
  std::map<std::string, std::pair<std::vector<int64_t>, std::vector<int64_t> > > gids;
};

extern "C" int mxh_dist_ops_create(mxg_ctx* ctx, mxh_sim* sim, mxh_dist_ops** out) {
  try {
    if (!ctx || !sim || !out) throw std::runtime_error("mxh_dist_ops_create: NULL argument");
    std::unique_ptr<mxh_dist_ops> d(new mxh_dist_ops());
    d->comm = rcp(new MxComm(ctx));
    d->sim = rcp(new MxEMSim<3>(sim->cfg, d->comm, Teuchos::ParameterList(), sim->asmb));
    d->sim->setPrebuiltOps(&sim->ops);
    d->ops = rcp(new MxEMOps<3, double>(d->sim, false));
    d->src = sim;
    *out = d.release();
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  } catch (int) {
    mxh_set_error(std::string("mxh_dist_ops_create failed: ") + mxg_last_error());
    return 1;
  }
}

extern "C" int mxh_dist_ops_matrix(mxh_dist_ops* d, const char* name, double shift, mxg_csr** A, int64_t* nrows,
                                   const int64_t** row_gids, int64_t* ncols_owned, const int64_t** col_gids,
                                   int64_t* nghost) {
  try {
    const std::string nm(name);
    char key[96];
    std::snprintf(key, sizeof(key), "%s@%.17g", name, nm == "shifted" ? shift : 0.0);
    auto it = d->made.find(key);
    if (it == d->made.end()) {
      RCP<MxCrsMatrix<double> > m;
      if (nm == "shifted") {
        RCP<MxMap> b = d->sim->getField("bfield").getMap();
        const mx::YeeField& B = d->sim->getAssembler().bfield();
        m = d->sim->isDistributed() ? d->ops->distributed(b, b, B, B, mxh_sim_shifted_matrix(d->src, shift))
                                    : rcp(new MxCrsMatrix<double>(b, b, mxh_sim_shifted_matrix(d->src, shift)));
      } else {
        if (!d->src->ops.count(nm)) mxh_sim_ensure_chain(d->src);
        auto op = d->src->ops.find(nm);
        if (op == d->src->ops.end()) throw std::runtime_error("MxEMOps::getOp: operator '" + nm + "' not found.");
        d->ops->adopt(nm, op->second);
        m = d->ops->getOp(nm);
      }
      it = d->made.emplace(key, m).first;
      auto& g = d->gids[key];
      for (MxIndex v : m->getRangeMap()->getNodeIndices()) g.first.push_back((int64_t)v);
      for (MxIndex v : m->getDomainMap()->getNodeIndices()) g.second.push_back((int64_t)v);
    }
    auto& g = d->gids[key];
    if (A) *A = it->second->getRawMatrix();
    if (nrows) *nrows = (int64_t)g.first.size();
    if (row_gids) *row_gids = g.first.data();
    if (ncols_owned) *ncols_owned = (int64_t)g.second.size();
    if (col_gids) *col_gids = g.second.data();
    if (nghost) *nghost = it->second->numGhosts();
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  } catch (int) {
    mxh_set_error(std::string("mxh_dist_ops_matrix failed: ") + mxg_last_error());
    return 1;
  }
}

extern "C" int mxh_dist_ops_destroy(mxh_dist_ops* d) {
  delete d;
  return 0;
}
