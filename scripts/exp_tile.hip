// Experiment harness (NOT part of libmxg.so): compiles the tile product's translation unit a second time into
// scripts/bin/libexp_tile.so and times its parts separately on a matrix handle made by libmxg.so - tiles only, group
// rows only, long rows only - plus variants of a pure stream kernel.  Build (scripts/ab_tile_parts.py does it):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude -Imaxwell_amd/csrc/device \
//         scripts/exp_tile.hip -o scripts/bin/libexp_tile.so -Lmaxwell_amd/lib -lmxg -Wl,-Bsymbolic,-rpath,$PWD/maxwell_amd/lib
#include "../maxwell_amd/csrc/device/mxg_tile.hip"

// parts: 1 short-pattern tiles, 2 long-pattern tiles, 4 group rows, 8 long rows (results are wrong unless parts == 15)
extern "C" int exp_tile_time(mxg_ctx* c, mxg_csr* a, const mxg_mv* x, mxg_mv* y, int parts, int order, int reps, float* ms) {
  if (!a->tile_on || a->is_complex) return 1;
  TileArgs A{};
  A.ntA = (parts & 1) ? a->tile_ntA : 0;
  A.ntB = (parts & 2) ? a->tile_ntB : 0;
  A.grow = a->d_grow; A.gcol = a->d_gcol; A.gval = a->d_gval;
  A.lrows = a->d_lrows;
  A.rowptr = (const long long*)a->d_rowptr; A.colidx = a->d_colidx; A.vals = a->d_vals;
  A.ldx = x->ld; A.ldy = y->ld; A.order = order;
  A.nowned = a->ncols_owned;
  A.ng0 = 0; A.ngn = (parts & 4) ? a->gen_ngrp : 0; A.nl0 = 0; A.nln = (parts & 8) ? a->gen_nlong : 0;
  A.ngbG = ceil_div(A.ngn, TILE_WAVES); A.ngbL = ceil_div(A.nln, TILE_BLOCK); A.ngb = up8(A.ngbG) + up8(A.ngbL);
  A.ntbA = ceil_div(A.ntA, (long long)TILE_WAVES * TILE_TPW); A.ntbB = ceil_div(A.ntB, (long long)TILE_WAVES);
  const long long grid = A.ngb + up8(A.ntbA);
  if (grid <= 0) { *ms = 0.f; return 0; }
  for (int r = 0; r < 3; ++r)
    hipLaunchKernelGGL((k_spmv_tile<false, false, 0, false>), dim3((unsigned)grid, (unsigned)x->k), dim3(TILE_BLOCK), 0, c->stream,
                       (const TileDesc*)a->d_tiledescA, (const void*)a->d_tiledictA, (const GenGroup*)a->d_ggrp,
                       (const double*)x->colptr(0), y->colptr(0), A);
  hipEventRecord(c->ev0, c->stream);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((k_spmv_tile<false, false, 0, false>), dim3((unsigned)grid, (unsigned)x->k), dim3(TILE_BLOCK), 0, c->stream,
                       (const TileDesc*)a->d_tiledescA, (const void*)a->d_tiledictA, (const GenGroup*)a->d_ggrp,
                       (const double*)x->colptr(0), y->colptr(0), A);
  hipEventRecord(c->ev1, c->stream);
  if (hipEventSynchronize(c->ev1) != hipSuccess) return 2;
  hipEventElapsedTime(ms, c->ev0, c->ev1);
  *ms /= reps;
  return 0;
}

// tiles on the compute stream, group rows on the communication stream, concurrently (two launches): ms per pair
extern "C" int exp_tile_two_streams(mxg_ctx* c, mxg_csr* a, const mxg_mv* x, mxg_mv* y, int reps, float* ms) {
  if (!a->tile_on || a->is_complex) return 1;
  auto args = [&](int parts) {
    TileArgs A{};
    A.ntA = (parts & 1) ? a->tile_ntA : 0;
    A.grow = a->d_grow; A.gcol = a->d_gcol; A.gval = a->d_gval; A.lrows = a->d_lrows;
    A.rowptr = (const long long*)a->d_rowptr; A.colidx = a->d_colidx; A.vals = a->d_vals;
    A.ldx = x->ld; A.ldy = y->ld; A.order = 0; A.nowned = a->ncols_owned;
    A.ng0 = 0; A.ngn = (parts & 4) ? a->gen_ngrp : 0; A.nl0 = 0; A.nln = 0;
    A.ngbG = ceil_div(A.ngn, TILE_WAVES); A.ngbL = 0; A.ngb = up8(A.ngbG);
    A.ntbA = ceil_div(A.ntA, (long long)TILE_WAVES * TILE_TPW); A.ntbB = 0;
    return A;
  };
  const TileArgs T = args(1), G = args(4);
  const long long gt = T.ngb + up8(T.ntbA), gg = G.ngb + up8(G.ntbA);
  hipEvent_t e2;
  hipEventCreate(&e2);
  auto pair = [&]() {
    hipEventRecord(c->ev_pack, c->stream);
    hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0);
    hipLaunchKernelGGL((k_spmv_tile<false, false, 0, false>), dim3((unsigned)gg), dim3(TILE_BLOCK), 0, c->comm_stream,
                       (const TileDesc*)a->d_tiledescA, (const void*)a->d_tiledictA, (const GenGroup*)a->d_ggrp,
                       (const double*)x->colptr(0), y->colptr(0), G);
    hipLaunchKernelGGL((k_spmv_tile<false, false, 0, false>), dim3((unsigned)gt), dim3(TILE_BLOCK), 0, c->stream,
                       (const TileDesc*)a->d_tiledescA, (const void*)a->d_tiledictA, (const GenGroup*)a->d_ggrp,
                       (const double*)x->colptr(0), y->colptr(0), T);
    hipEventRecord(e2, c->comm_stream);
    hipStreamWaitEvent(c->stream, e2, 0);
  };
  for (int r = 0; r < 3; ++r) pair();
  hipEventRecord(c->ev0, c->stream);
  for (int r = 0; r < reps; ++r) pair();
  hipEventRecord(c->ev1, c->stream);
  if (hipEventSynchronize(c->ev1) != hipSuccess) return 2;
  hipEventElapsedTime(ms, c->ev0, c->ev1);
  *ms /= reps;
  hipEventDestroy(e2);
  return 0;
}

// ---- pure stream variants: read n16 16-byte pieces, write m16 ------------------------------------------------------
typedef double d2 __attribute__((ext_vector_type(2)));
template <bool NT, int U, bool CHUNK>
__global__ __launch_bounds__(256) void k_stream(const d2* __restrict__ src, long long n16, d2* __restrict__ dst, long long m16,
                                                double* __restrict__ sink) {
  d2 acc; acc.x = 0.0; acc.y = 0.0;
  if (CHUNK) {      // every workgroup owns one contiguous chunk (XCD-contiguous eighths like the SpMV kernels)
    const long long nb = gridDim.x, per = (nb + 7) / 8, b = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (b < nb) {
      const long long c0 = n16 * b / nb, c1 = n16 * (b + 1) / nb;
      for (long long i = c0 + threadIdx.x; i < c1; i += 256 * U) {
        d2 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long long q = i + 256 * u; t[u] = q < c1 ? (NT ? __builtin_nontemporal_load(src + q) : src[q]) : acc * 0.0; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += t[u];
      }
      const long long w0 = m16 * b / nb, w1 = m16 * (b + 1) / nb;
      for (long long j = w0 + threadIdx.x; j < w1; j += 256) __builtin_nontemporal_store(acc, dst + j);
    }
  } else {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i < n16; i += U * stride) {
      d2 t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const long long q = i + u * stride; t[u] = q < n16 ? (NT ? __builtin_nontemporal_load(src + q) : src[q]) : acc * 0.0; }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += t[u];
    }
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < m16; j += stride) __builtin_nontemporal_store(acc, dst + j);
  }
  if (acc.x == 1.2345e300) sink[0] = acc.y;
}

extern "C" int exp_stream_time(mxg_ctx* c, long long read_bytes, long long write_bytes, int variant, int wgs_per_cu, int reps, float* ms) {
  void *src = nullptr, *dst = nullptr; double* sink = nullptr;
  if (hipMalloc(&src, read_bytes) != hipSuccess || hipMalloc(&dst, write_bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) return 1;
  hipMemsetAsync(src, 0, read_bytes, c->stream);
  const long long n16 = read_bytes / 16, m16 = write_bytes / 16;
  auto launch = [&]() {
    const unsigned grid = variant >= 4 ? (unsigned)(n16 / 2048 / 8 * 8) : (unsigned)(c->num_cu * wgs_per_cu);   // variants >= 4: one 32 KiB chunk per workgroup
    switch (variant) {
      case 0: hipLaunchKernelGGL((k_stream<true, 4, false>), dim3(grid), dim3(256), 0, c->stream, (const d2*)src, n16, (d2*)dst, m16, sink); break;
      case 1: hipLaunchKernelGGL((k_stream<false, 4, false>), dim3(grid), dim3(256), 0, c->stream, (const d2*)src, n16, (d2*)dst, m16, sink); break;
      case 2: hipLaunchKernelGGL((k_stream<true, 8, false>), dim3(grid), dim3(256), 0, c->stream, (const d2*)src, n16, (d2*)dst, m16, sink); break;
      case 3: hipLaunchKernelGGL((k_stream<true, 4, true>), dim3(grid), dim3(256), 0, c->stream, (const d2*)src, n16, (d2*)dst, m16, sink); break;
      case 4: hipLaunchKernelGGL((k_stream<true, 8, true>), dim3(grid), dim3(256), 0, c->stream, (const d2*)src, n16, (d2*)dst, m16, sink); break;
      default: hipLaunchKernelGGL((k_stream<false, 8, true>), dim3(grid), dim3(256), 0, c->stream, (const d2*)src, n16, (d2*)dst, m16, sink); break;
    }
  };
  for (int r = 0; r < 3; ++r) launch();
  hipEventRecord(c->ev0, c->stream);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(c->ev1, c->stream);
  hipError_t e = hipEventSynchronize(c->ev1);
  if (e == hipSuccess) hipEventElapsedTime(ms, c->ev0, c->ev1);
  *ms /= reps;
  hipFree(src); hipFree(dst); hipFree(sink);
  return e == hipSuccess ? 0 : 2;
}
