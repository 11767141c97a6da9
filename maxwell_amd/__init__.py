"""maxwell_amd -- MI355X-native Krylov hot path of the maxwell eigensolver.

Thin ctypes front end over the two native libraries built in-tree:

  libmxg.so     include/mxg.h : HIP kernels (CSR SpMV/SpMM, block BLAS of the Krylov basis,
                                CG / GMRES) behind a C ABI
  libmxhost.so  include/mxh.h : host C++ (Yee / Dey-Mittra assembly, MxMagWaveOp, MxSolver)

There is no CPU fallback: importing works without a GPU (so the build and the symbol checks
run anywhere), but every compute call goes to the HIP library and raises if it is missing.
"""
from ._lib import load_mxg, load_mxhost, MxError  # noqa: F401
from .api import Ctx, CsrMatrix, LocalWorld, MultiVec, Multigrid, Problem, Sim, Solver, SolverParams  # noqa: F401
