// One rank = one GPU.  Replaces MxComm / Epetra_MpiComm (reference src/MxComm.hpp:14-37).
#ifndef MX_COMM
#define MX_COMM

#include "MxTypes.h"

class MxComm {
 public:
  // serial: one context on `device`
  explicit MxComm(int device = 0) : owns_(true) {
    MX_ABI(mxg_ctx_create(device, 0, 1, NULL, &ctx_), "MxComm");
  }
  // one process per GPU: rank / nranks / RCCL unique id broadcast by the launcher
  MxComm(int device, int rank, int nranks, const void* ncclUniqueId) : owns_(true) {
    MX_ABI(mxg_ctx_create(device, rank, nranks, ncclUniqueId, &ctx_), "MxComm");
  }
  // adopt an existing context (not destroyed)
  explicit MxComm(mxg_ctx* ctx) : ctx_(ctx), owns_(false) {}
  ~MxComm() { if (owns_ && ctx_) mxg_ctx_destroy(ctx_); }
  int myPID() const { return mxg_ctx_rank(ctx_); }
  int numProc() const { return mxg_ctx_nranks(ctx_); }
  void barrier() const { mxg_barrier(ctx_); }
  mxg_ctx* getCtx() const { return ctx_; }
 private:
  MxComm(const MxComm&);
  MxComm& operator=(const MxComm&);
  mxg_ctx* ctx_ = nullptr;
  bool owns_;
};

#endif
