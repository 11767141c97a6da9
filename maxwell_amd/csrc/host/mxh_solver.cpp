// placeholder until the eigen-driver lands (replaced in the next commit)
#include "mxh.h"
#include "mxh_internal.hpp"
extern "C" void mxh_solver_default_params(mxh_solver_params* p) {
  p->nev = 10; p->basis = 20; p->block_size = 1; p->max_restarts = 100; p->tol = 1e-6; p->shift = 0.0;
  p->lin_type = "gmres"; p->lin_basis = 40; p->lin_tol = 1e-6; p->prec_type = "none"; p->verbose = 0; p->seed = 12345;
}
extern "C" int mxh_solver_create(mxg_ctx*, mxh_sim*, const mxh_solver_params*, mxh_solver**) { mxh_set_error("not implemented"); return 1; }
extern "C" int mxh_solver_destroy(mxh_solver*) { return 0; }
extern "C" int mxh_solver_solve(mxh_solver*, int*) { mxh_set_error("not implemented"); return 1; }
extern "C" int mxh_solver_result(mxh_solver*, int, double*, double*, double*, double*, double*) { return 1; }
extern "C" int mxh_solver_stats(mxh_solver*, int*, int*, int*, double*, double*, double*) { return 1; }
extern "C" int mxh_solver_eigvec(mxh_solver*, int, double*) { return 1; }
