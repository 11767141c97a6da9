// Context: device, stream, RCCL communicator, workspaces, timers.
// Replaces MxComm / Epetra_MpiComm (reference src/MxComm.hpp:14-37): one process per GPU,
// collectives go through RCCL on the ctx stream.
#include <cstdarg>
#include <mutex>
#include <set>

#include "mxg_internal.h"

static thread_local std::string g_last_error;

// live contexts: destroy calls that arrive after their context is gone (interpreter shutdown
// destroys objects in arbitrary order) must not touch the device through a dangling pointer
static std::mutex g_ctx_mutex;
static std::set<const mxg_ctx*> g_live_ctx;
bool mxg_ctx_alive(const mxg_ctx* c) {
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  return g_live_ctx.count(c) != 0;
}

void mxg_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

extern "C" const char* mxg_last_error(void) { return g_last_error.c_str(); }

extern "C" int mxg_get_unique_id(void* out_id) {
  static_assert(sizeof(ncclUniqueId) <= MXG_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId id;
  MXG_NCCL(ncclGetUniqueId(&id));
  memset(out_id, 0, MXG_UNIQUE_ID_BYTES);
  memcpy(out_id, &id, sizeof(id));
  return 0;
}

extern "C" int mxg_ctx_create(int device, int rank, int nranks, const void* nccl_unique_id,
                              mxg_ctx** out) {
  MXG_CHECK(out != nullptr, "mxg_ctx_create: out is NULL");
  int ndev = 0;
  MXG_HIP(hipGetDeviceCount(&ndev));
  MXG_CHECK(ndev > 0, "mxg_ctx_create: no HIP device visible");
  MXG_CHECK(device >= 0 && device < ndev, "mxg_ctx_create: device %d out of range (%d visible)",
            device, ndev);
  MXG_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "mxg_ctx_create: bad rank %d / %d", rank,
            nranks);
  MXG_HIP(hipSetDevice(device));
  mxg_ctx* c = new mxg_ctx();
  c->device = device;
  c->rank = rank;
  c->nranks = nranks;
  hipDeviceProp_t prop;
  MXG_HIP(hipGetDeviceProperties(&prop, device));
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  MXG_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  MXG_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  MXG_HIP(hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming));
  MXG_HIP(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
  MXG_HIP(hipEventCreateWithFlags(&c->ev_scal, hipEventDisableTiming));
  MXG_HIP(hipEventCreate(&c->ev0));
  MXG_HIP(hipEventCreate(&c->ev1));
  MXG_CHECK(nranks == 1 || nccl_unique_id != nullptr, "mxg_ctx_create: nranks > 1 needs an RCCL unique id");
  if (nccl_unique_id != nullptr) {   // also for nranks == 1: a 1-rank communicator (self send/recv works)
    ncclUniqueId id;
    memcpy(&id, nccl_unique_id, sizeof(id));
    MXG_NCCL(ncclCommInitRank(&c->comm, nranks, id, rank));
  }
  if (mxg_ctx_reserve_reduce(c, 64) != 0) return 1;
  { std::lock_guard<std::mutex> lock(g_ctx_mutex); g_live_ctx.insert(c); }
  *out = c;
  return 0;
}

// ---- vector storage cache ------------------------------------------------------------------------------------------
// Reuse is ordered by ctx->stream: a released buffer may still be read by kernels already enqueued there, and the next
// owner's first use is enqueued behind them.  (Vectors are only ever touched on ctx->stream, or on ctx->comm_stream
// between events that ctx->stream waits for.)
namespace {
constexpr size_t VEC_CACHE_MAX_BYTES = (size_t)8 << 30, VEC_CACHE_MAX_ONE = (size_t)2 << 30, VEC_CACHE_MAX_COUNT = 32;
}

int mxg_vec_alloc(mxg_ctx* c, size_t bytes, void** out) {
  {
  std::lock_guard<std::mutex> lock(c->vec_cache_mutex);
  for (size_t i = c->vec_cache.size(); i-- > 0;)
    if (c->vec_cache[i].first == bytes) {
      *out = c->vec_cache[i].second;
      c->vec_cache_bytes -= bytes;
      c->vec_cache.erase(c->vec_cache.begin() + (long)i);
      return 0;
    }
  }
  if (hipMalloc(out, bytes) == hipSuccess) return 0;
  (void)hipGetLastError();
  mxg_vec_cache_flush(c);                     // out of memory: give the cached buffers back and try once more
  MXG_HIP(hipMalloc(out, bytes));
  return 0;
}

void mxg_vec_release(mxg_ctx* c, void* p, size_t bytes) {
  static const bool off = getenv("MXG_VEC_CACHE") && atoi(getenv("MXG_VEC_CACHE")) == 0;   // experiments
  std::vector<void*> drop;                    // the newest buffers stay: the oldest make room (another phase's sizes)
  {
    std::lock_guard<std::mutex> lock(c->vec_cache_mutex);
    if (off || bytes > VEC_CACHE_MAX_ONE) drop.push_back(p);
    else {
      c->vec_cache.emplace_back(bytes, p);
      c->vec_cache_bytes += bytes;
      while (c->vec_cache.size() > VEC_CACHE_MAX_COUNT || c->vec_cache_bytes > VEC_CACHE_MAX_BYTES) {
        drop.push_back(c->vec_cache.front().second);
        c->vec_cache_bytes -= c->vec_cache.front().first;
        c->vec_cache.erase(c->vec_cache.begin());
      }
    }
  }
  if (drop.empty()) return;
  hipStreamSynchronize(c->stream);
  for (void* q : drop) hipFree(q);
}

void mxg_vec_cache_flush(mxg_ctx* c) {
  std::lock_guard<std::mutex> lock(c->vec_cache_mutex);
  if (c->vec_cache.empty()) return;
  hipStreamSynchronize(c->stream);
  for (auto& e : c->vec_cache) hipFree(e.second);
  c->vec_cache.clear();
  c->vec_cache_bytes = 0;
}

extern "C" int mxg_ctx_destroy(mxg_ctx* c) {
  if (!c || !mxg_ctx_alive(c)) return 0;
  { std::lock_guard<std::mutex> lock(g_ctx_mutex); g_live_ctx.erase(c); }
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  mxg_vec_cache_flush(c);
  if (c->comm) ncclCommDestroy(c->comm);
  if (c->d_partial) hipFree(c->d_partial);
  if (c->d_result) hipFree(c->d_result);
  if (c->h_result) hipHostFree(c->h_result);
  if (c->d_small) hipFree(c->d_small);
  if (c->d_scratch) hipFree(c->d_scratch);
  hipEventDestroy(c->ev0);
  hipEventDestroy(c->ev1);
  if (c->ev_pack) hipEventDestroy(c->ev_pack);
  if (c->ev_halo) hipEventDestroy(c->ev_halo);
  if (c->ev_scal) hipEventDestroy(c->ev_scal);
  if (c->comm_stream) { hipStreamSynchronize(c->comm_stream); hipStreamDestroy(c->comm_stream); }
  hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

extern "C" int mxg_ctx_sync(mxg_ctx* c) {
  MXG_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int mxg_ctx_rank(const mxg_ctx* c) { return c->rank; }
extern "C" int mxg_ctx_nranks(const mxg_ctx* c) { return c->nranks; }

// ranks the RCCL communicator itself reports (ncclCommCount); 1 without a communicator, -1 on error:
// lets a launcher verify that RCCL really connected as many ranks as it started
extern "C" int mxg_ctx_comm_ranks(const mxg_ctx* c) {
  if (!c->comm) return c->lw ? c->nranks : 1;
  int n = -1;
  if (ncclCommCount(c->comm, &n) != ncclSuccess) return -1;
  return n;
}

extern "C" int mxg_timer_start(mxg_ctx* c) {
  MXG_HIP(hipEventRecord(c->ev0, c->stream));
  return 0;
}

extern "C" int mxg_timer_stop(mxg_ctx* c, float* ms) {
  MXG_HIP(hipEventRecord(c->ev1, c->stream));
  MXG_HIP(hipEventSynchronize(c->ev1));
  MXG_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}

extern "C" int mxg_barrier(mxg_ctx* c) {
  if (mxg_comm_on(c)) {
    MXG_HIP(hipMemsetAsync(c->d_result, 0, sizeof(double), c->stream));
    MXG_TRY(mxg_comm_allreduce(c, c->d_result, 1, c->stream));
  }
  MXG_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int mxg_ctx_reserve_reduce(mxg_ctx* c, size_t width) {
  if (width <= c->red_width) return 0;
  MXG_HIP(hipStreamSynchronize(c->stream));
  if (c->d_partial) hipFree(c->d_partial);
  if (c->d_result) hipFree(c->d_result);
  if (c->h_result) hipHostFree(c->h_result);
  size_t w = 64;
  while (w < width) w *= 2;
  MXG_HIP(hipMalloc(&c->d_partial, sizeof(double) * w * MXG_MAX_RED_BLOCKS));
  MXG_HIP(hipMalloc(&c->d_result, sizeof(double) * w));
  MXG_HIP(hipHostMalloc(&c->h_result, sizeof(double) * w, hipHostMallocDefault));
  void* dev = nullptr;
  c->h_result_dev = hipHostGetDevicePointer(&dev, c->h_result, 0) == hipSuccess ? (double*)dev : nullptr;
  c->red_width = w;
  return 0;
}

int mxg_ctx_reserve_small(mxg_ctx* c, size_t doubles) {
  if (doubles <= c->small_cap) return 0;
  MXG_HIP(hipStreamSynchronize(c->stream));
  if (c->d_small) hipFree(c->d_small);
  size_t w = 4096;
  while (w < doubles) w *= 2;
  MXG_HIP(hipMalloc(&c->d_small, sizeof(double) * w));
  c->small_cap = w;
  return 0;
}

int mxg_ctx_reserve_scratch(mxg_ctx* c, size_t doubles) {
  if (doubles <= c->scratch_cap) return 0;
  MXG_HIP(hipStreamSynchronize(c->stream));
  if (c->d_scratch) hipFree(c->d_scratch);
  MXG_HIP(hipMalloc(&c->d_scratch, sizeof(double) * doubles));
  c->scratch_cap = doubles;
  return 0;
}

int mxg_ctx_finish_reduce(mxg_ctx* c, size_t width, double* host_out) {
  // C2: the MPI_Allreduce inside Epetra_MultiVector::Dot / Norm2 / Multiply('T', ...)
  if (mxg_comm_on(c)) MXG_TRY(mxg_comm_allreduce(c, c->d_result, width, c->stream));
  MXG_HIP(hipMemcpyAsync(c->h_result, c->d_result, sizeof(double) * width, hipMemcpyDeviceToHost,
                         c->stream));
  MXG_HIP(hipStreamSynchronize(c->stream));
  memcpy(host_out, c->h_result, sizeof(double) * width);
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// Stream ceiling of THIS device: the byte mix of one CSR product (read `read_bytes` once with 16-byte non-temporal
// loads, write `write_bytes` once with non-temporal stores) with no gather, no LDS and no index arithmetic.  bench.py
// reports it beside the SpMV roofline fraction: the devices of a pool differ by several per cent, and what a pure
// stream reaches on the same device in the same run is the ceiling the kernel can be held against.
// ---------------------------------------------------------------------------------------------------------------
namespace {
typedef double dbl2s __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_stream_ceiling(const dbl2s* __restrict__ src, long long n16, dbl2s* __restrict__ dst,
                                                         long long m16, double* __restrict__ sink) {
  // every workgroup owns one contiguous chunk, XCD w % 8 a contiguous eighth (the mapping of the SpMV kernels); four
  // 16-byte non-temporal loads in flight per thread.  Of the shapes tried (grid-stride / chunked, plain / non-temporal,
  // 4 / 8 loads in flight, 4 / 8 / 16 workgroups per CU: scripts/exp_tile.hip) this one reads fastest: 6.0 TB/s.
  const long long nb = gridDim.x, per = (nb + 7) / 8, b = (blockIdx.x % 8) * per + blockIdx.x / 8;
  dbl2s acc; acc.x = 0.0; acc.y = 0.0;
  if (b < nb) {
    const long long c0 = n16 * b / nb, c1 = n16 * (b + 1) / nb;
    for (long long i = c0 + threadIdx.x; i < c1; i += 256 * 4) {
      dbl2s t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const long long q = i + 256 * u; t[u] = q < c1 ? __builtin_nontemporal_load(src + q) : acc * 0.0; }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += t[u];
    }
    const long long w0 = m16 * b / nb, w1 = m16 * (b + 1) / nb;
    for (long long j = w0 + threadIdx.x; j < w1; j += 256) __builtin_nontemporal_store(acc, dst + j);
  }
  if (acc.x == 1.2345e300) sink[0] = acc.y;                   // keeps the loads alive
}
}  // namespace

extern "C" int mxg_bench_stream(mxg_ctx* c, int64_t read_bytes, int64_t write_bytes, int reps, float* ms_per_pass) {
  MXG_CHECK(c && ms_per_pass && read_bytes >= 16 && write_bytes >= 16 && reps > 0, "mxg_bench_stream: bad argument");
  MXG_HIP(hipSetDevice(c->device));
  void *src = nullptr, *dst = nullptr;
  double* sink = nullptr;
  MXG_HIP(hipMalloc(&src, (size_t)read_bytes));
  if (hipMalloc(&dst, (size_t)write_bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) {
    hipFree(src); if (dst) hipFree(dst);
    mxg_set_error("mxg_bench_stream: out of device memory");
    return 1;
  }
  hipMemsetAsync(src, 0, (size_t)read_bytes, c->stream);
  const unsigned grid = (unsigned)(c->num_cu * 8);
  for (int r = 0; r < 3; ++r)
    hipLaunchKernelGGL(k_stream_ceiling, dim3(grid), dim3(256), 0, c->stream, (const dbl2s*)src, (long long)(read_bytes / 16),
                       (dbl2s*)dst, (long long)(write_bytes / 16), sink);
  hipEventRecord(c->ev0, c->stream);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_stream_ceiling, dim3(grid), dim3(256), 0, c->stream, (const dbl2s*)src, (long long)(read_bytes / 16),
                       (dbl2s*)dst, (long long)(write_bytes / 16), sink);
  hipEventRecord(c->ev1, c->stream);
  hipError_t e = hipEventSynchronize(c->ev1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
  hipFree(src); hipFree(dst); hipFree(sink);
  if (e != hipSuccess) { mxg_set_error("mxg_bench_stream: %s", hipGetErrorString(e)); return 1; }
  *ms_per_pass = ms / reps;
  return 0;
}
