// Collectives of one rank-per-box run: RCCL over xGMI (one process per GPU), or an in-process world
// whose ranks are threads of one process sharing a device (several MxGridDomain boxes on one GPU; this is
// also how the N > 1 code path is exercised on a single-GPU box, where RCCL refuses two ranks per device).
// Replaces the MPI layer under Epetra (reference src/MxComm.hpp:14-37): all-reduce for inner products,
// neighbour exchange for the ghost import, all-gather of host data for the setup-time plan agreement.
#include <chrono>
#include <condition_variable>
#include <mutex>

#include "mxg_internal.h"

struct mxg_local_world {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0;
  unsigned long long gen = 0;
  bool broken = false;
  std::vector<std::vector<double>> red;                 // per rank: all-reduce staging
  struct Send { int peer; const double* ptr; int64_t n; };
  std::vector<std::vector<Send>> sends;                 // per rank: sends of the current exchange, in issue order
  std::vector<const void*> gptr;                        // per rank: all-gather source
  std::vector<int64_t> gcnt;
  std::vector<int> opkind;                              // per rank: which collective it is in (checked at the rendezvous:
  std::vector<long long> opsize;                        //  over RCCL a mismatch would be a hang, here it is an error)
  // every rank calls the collectives in the same order; a rank that never arrives (it failed) breaks the
  // world for everyone after the time limit instead of hanging the process
  int barrier() {
    std::unique_lock<std::mutex> lock(m);
    if (broken) return 1;
    const unsigned long long g = gen;
    if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); return 0; }
    const bool ok = cv.wait_for(lock, std::chrono::seconds(300), [&] { return gen != g || broken; });
    if (!ok || broken) { broken = true; cv.notify_all(); return 1; }
    return 0;
  }
};

extern "C" int mxg_local_world_create(int nranks, mxg_local_world** out) {
  MXG_CHECK(out && nranks >= 1, "mxg_local_world_create: bad arguments");
  mxg_local_world* w = new mxg_local_world();
  w->n = nranks;
  w->red.resize(nranks);
  w->sends.resize(nranks);
  w->gptr.assign(nranks, nullptr);
  w->gcnt.assign(nranks, 0);
  w->opkind.assign(nranks, 0);
  w->opsize.assign(nranks, 0);
  *out = w;
  return 0;
}

extern "C" int mxg_local_world_destroy(mxg_local_world* w) {
  delete w;
  return 0;
}

extern "C" int mxg_ctx_create_local(int device, int rank, int nranks, mxg_local_world* w, mxg_ctx** out) {
  MXG_CHECK(w && out, "mxg_ctx_create_local: NULL argument");
  MXG_CHECK(nranks == w->n && rank >= 0 && rank < nranks, "mxg_ctx_create_local: bad rank %d / %d (world of %d)", rank,
            nranks, w->n);
  MXG_TRY(mxg_ctx_create(device, 0, 1, nullptr, out));
  (*out)->rank = rank;
  (*out)->nranks = nranks;
  (*out)->lw = nranks > 1 ? w : nullptr;
  return 0;
}

// every rank announces the collective it is entering; after the rendezvous they must all agree
#define MXG_LW_ENTER(w, c, kind, size)                                                                          \
  do {                                                                                                          \
    (w)->opkind[(c)->rank] = (kind);                                                                            \
    (w)->opsize[(c)->rank] = (long long)(size);                                                                 \
  } while (0)
#define MXG_LW_AGREE(w, c, what, check_size)                                                                    \
  do {                                                                                                          \
    for (int r_ = 0; r_ < (w)->n; ++r_)                                                                         \
      if ((w)->opkind[r_] != (w)->opkind[(c)->rank] || ((check_size) && (w)->opsize[r_] != (w)->opsize[(c)->rank])) { \
        mxg_set_error("%s: rank %d is in collective %d (size %lld), rank %d in %d (size %lld): the ranks diverged", what, \
                      (c)->rank, (w)->opkind[(c)->rank], (w)->opsize[(c)->rank], r_, (w)->opkind[r_], (w)->opsize[r_]); \
        return 1;                                                                                               \
      }                                                                                                         \
  } while (0)

#define MXG_LW_BARRIER(w)                                                                  \
  do {                                                                                     \
    if ((w)->barrier() != 0) {                                                             \
      mxg_set_error("in-process world: a rank did not reach the collective (time limit)"); \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

bool mxg_comm_on(const mxg_ctx* c) { return c->comm != nullptr || c->lw != nullptr; }

// C2: the MPI_Allreduce inside Epetra_MultiVector::Dot / Norm2 / Multiply('T', ...); in place, on stream s
int mxg_comm_allreduce(mxg_ctx* c, double* d, size_t count, hipStream_t s) {
  if (c->comm) {
    MXG_NCCL(ncclAllReduce(d, d, count, ncclDouble, ncclSum, c->comm, s));
    return 0;
  }
  mxg_local_world* w = c->lw;
  if (!w) return 0;
  std::vector<double>& mine = w->red[c->rank];
  mine.resize(count);
  MXG_HIP(hipMemcpyAsync(mine.data(), d, sizeof(double) * count, hipMemcpyDeviceToHost, s));
  MXG_HIP(hipStreamSynchronize(s));
  MXG_LW_ENTER(w, c, 1, count);
  MXG_LW_BARRIER(w);
  MXG_LW_AGREE(w, c, "all-reduce", true);
  std::vector<double> sum(count, 0.0);
  for (int r = 0; r < w->n; ++r) {          // rank order: every rank forms the same sum bit for bit
    MXG_CHECK(w->red[r].size() == count, "all-reduce: rank %d brought %zu values, rank %d %zu", r, w->red[r].size(),
              c->rank, count);
    for (size_t i = 0; i < count; ++i) sum[i] += w->red[r][i];
  }
  MXG_LW_BARRIER(w);                        // nobody overwrites its staging while others still read it
  MXG_HIP(hipMemcpyAsync(d, sum.data(), sizeof(double) * count, hipMemcpyHostToDevice, s));
  MXG_HIP(hipStreamSynchronize(s));
  return 0;
}

// C1: grouped neighbour exchange; the i-th send of rank a to rank b matches the i-th receive of b from a
int mxg_comm_exchange(mxg_ctx* c, const MxgXfer* x, int n, hipStream_t s) {
  if (c->comm) {
    MXG_NCCL(ncclGroupStart());
    for (int i = 0; i < n; ++i) {
      if (x[i].ns > 0) MXG_NCCL(ncclSend(x[i].send, x[i].ns, ncclDouble, x[i].peer, c->comm, s));
      if (x[i].nr > 0) MXG_NCCL(ncclRecv(x[i].recv, x[i].nr, ncclDouble, x[i].peer, c->comm, s));
    }
    MXG_NCCL(ncclGroupEnd());
    return 0;
  }
  mxg_local_world* w = c->lw;
  MXG_CHECK(w != nullptr, "halo exchange needs a communicator (nranks > 1)");
  MXG_HIP(hipStreamSynchronize(s));         // the packed send buffers are complete
  std::vector<mxg_local_world::Send>& mine = w->sends[c->rank];
  mine.clear();
  for (int i = 0; i < n; ++i)
    if (x[i].ns > 0) mine.push_back({x[i].peer, x[i].send, x[i].ns});
  MXG_LW_ENTER(w, c, 2, 0);
  MXG_LW_BARRIER(w);
  MXG_LW_AGREE(w, c, "neighbour exchange", false);
  std::vector<size_t> cursor(w->n, 0);
  for (int i = 0; i < n; ++i) {
    if (x[i].nr <= 0) continue;
    const int p = x[i].peer;
    const std::vector<mxg_local_world::Send>& theirs = w->sends[p];
    size_t& q = cursor[p];
    while (q < theirs.size() && theirs[q].peer != c->rank) ++q;
    MXG_CHECK(q < theirs.size() && theirs[q].n == x[i].nr, "halo exchange: rank %d has no matching send for rank %d", p,
              c->rank);
    MXG_HIP(hipMemcpyAsync(x[i].recv, theirs[q].ptr, sizeof(double) * x[i].nr, hipMemcpyDeviceToDevice, s));
    ++q;
  }
  MXG_HIP(hipStreamSynchronize(s));
  MXG_LW_BARRIER(w);                        // send buffers may be reused
  return 0;
}

int mxg_comm_allgather_dev(mxg_ctx* c, const double* send, size_t n, double* recv, hipStream_t s) {
  if (c->comm) {
    MXG_NCCL(ncclAllGather(send, recv, n, ncclDouble, c->comm, s));
    return 0;
  }
  mxg_local_world* w = c->lw;
  if (!w) {
    MXG_HIP(hipMemcpyAsync(recv, send, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    return 0;
  }
  MXG_HIP(hipStreamSynchronize(s));
  w->gptr[c->rank] = send;
  MXG_LW_ENTER(w, c, 3, n);
  MXG_LW_BARRIER(w);
  MXG_LW_AGREE(w, c, "all-gather (device)", true);
  for (int r = 0; r < w->n; ++r)
    MXG_HIP(hipMemcpyAsync(recv + n * r, w->gptr[r], sizeof(double) * n, hipMemcpyDeviceToDevice, s));
  MXG_HIP(hipStreamSynchronize(s));
  MXG_LW_BARRIER(w);
  return 0;
}

extern "C" int mxg_comm_allgather_i64(mxg_ctx* c, int64_t v, int64_t* out) {
  MXG_CHECK(c && out, "mxg_comm_allgather_i64: NULL argument");
  if (!mxg_comm_on(c)) { out[0] = v; return 0; }
  if (c->comm) {
    int64_t* d = nullptr;
    MXG_HIP(hipMalloc(&d, sizeof(int64_t) * (c->nranks + 1)));
    MXG_HIP(hipMemcpyAsync(d + c->nranks, &v, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    MXG_NCCL(ncclAllGather(d + c->nranks, d, 1, ncclInt64, c->comm, c->stream));
    MXG_HIP(hipMemcpyAsync(out, d, sizeof(int64_t) * c->nranks, hipMemcpyDeviceToHost, c->stream));
    MXG_HIP(hipStreamSynchronize(c->stream));
    hipFree(d);
    return 0;
  }
  mxg_local_world* w = c->lw;
  w->gcnt[c->rank] = v;
  MXG_LW_ENTER(w, c, 4, 0);
  MXG_LW_BARRIER(w);
  MXG_LW_AGREE(w, c, "all-gather (int64)", false);
  for (int r = 0; r < w->n; ++r) out[r] = w->gcnt[r];
  MXG_LW_BARRIER(w);
  return 0;
}

// host data of every rank, concatenated in rank order; counts[r] = bytes of rank r (from mxg_comm_allgather_i64)
extern "C" int mxg_comm_allgatherv(mxg_ctx* c, const void* send, int64_t nbytes, const int64_t* counts, void* recv) {
  MXG_CHECK(c && counts && (recv || nbytes == 0), "mxg_comm_allgatherv: NULL argument");
  if (!mxg_comm_on(c)) { if (nbytes > 0) memcpy(recv, send, (size_t)nbytes); return 0; }
  MXG_CHECK(counts[c->rank] == nbytes, "mxg_comm_allgatherv: counts[rank] != nbytes");
  if (c->comm) {
    int64_t mx = 0, total = 0;
    for (int r = 0; r < c->nranks; ++r) { mx = std::max(mx, counts[r]); total += counts[r]; }
    if (total == 0) return 0;
    mx = (mx + 15) & ~(int64_t)15;
    char* d = nullptr;                       // [slot of every rank | my slot]
    MXG_HIP(hipMalloc(&d, (size_t)mx * (c->nranks + 1)));
    char* mine = d + (size_t)mx * c->nranks;
    if (nbytes > 0) MXG_HIP(hipMemcpyAsync(mine, send, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    MXG_NCCL(ncclAllGather(mine, d, (size_t)mx, ncclChar, c->comm, c->stream));
    int64_t off = 0;
    for (int r = 0; r < c->nranks; ++r) {
      if (counts[r] > 0)
        MXG_HIP(hipMemcpyAsync((char*)recv + off, d + (size_t)mx * r, (size_t)counts[r], hipMemcpyDeviceToHost, c->stream));
      off += counts[r];
    }
    MXG_HIP(hipStreamSynchronize(c->stream));
    hipFree(d);
    return 0;
  }
  mxg_local_world* w = c->lw;
  w->gptr[c->rank] = send;
  MXG_LW_ENTER(w, c, 5, 0);
  MXG_LW_BARRIER(w);
  MXG_LW_AGREE(w, c, "all-gather (host bytes)", false);
  int64_t off = 0;
  for (int r = 0; r < w->n; ++r) {
    if (counts[r] > 0) memcpy((char*)recv + off, w->gptr[r], (size_t)counts[r]);
    off += counts[r];
  }
  MXG_LW_BARRIER(w);
  return 0;
}

// sum of host doubles over the ranks (setup: global map sizes)
extern "C" int mxg_comm_allreduce_host(mxg_ctx* c, double* v, int n) {
  MXG_CHECK(c && v && n >= 0, "mxg_comm_allreduce_host: bad arguments");
  if (!mxg_comm_on(c) || n == 0) return 0;
  MXG_TRY(mxg_ctx_reserve_reduce(c, (size_t)n));
  MXG_HIP(hipMemcpyAsync(c->d_result, v, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  MXG_TRY(mxg_comm_allreduce(c, c->d_result, (size_t)n, c->stream));
  MXG_HIP(hipMemcpyAsync(v, c->d_result, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  return 0;
}
