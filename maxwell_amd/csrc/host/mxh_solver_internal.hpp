// mxh_solver: shared by mxh_solver.cpp (parameter struct entry) and mxh_problem.cpp (.mx deck entry).
#ifndef MXH_SOLVER_INTERNAL_HPP
#define MXH_SOLVER_INTERNAL_HPP

#include <memory>
#include <string>

#include "mx/MxSolver.h"
#include "mxh.h"
#include "mxh_internal.hpp"

struct mxh_solver {
  std::shared_ptr<mxh_sim> ownSim;                         // box simulation made for this solver (declared first: destroyed last)
  std::shared_ptr<mx::YeeAssembler<MxComplex> > boxAsmC;   // complex problem on a decomposed grid: window assembler + chain
  MxEMSim<3>::OpCacheC boxOpsC;
  RCP<MxComm> comm;
  RCP<MxEMSim<3> > sim;
  std::unique_ptr<MxSolver<3, double> > solver;            // real problems
  std::unique_ptr<MxSolver<3, MxComplex> > csolver;        // complex problems (Bloch phase shifts)
  mxh_solver_params params;
  std::string linType, precType, amgSmoother;
};

// comm + MxEMSim (sharing the assembler of `sim`) + MxSolver from a complete parameter list
void mxh_solver_build(mxh_solver& s, mxg_ctx* ctx, mxh_sim* sim, Teuchos::ParameterList const& pl);
// complex problems (phase shifts): MxEMSim assembles in complex arithmetic itself
void mxh_solver_build_complex(mxh_solver& s, mxg_ctx* ctx, mx::YeeConfig const& cfg, Teuchos::ParameterList const& pl);

#endif
