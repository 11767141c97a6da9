// Micro-benchmark behind DESIGN.md section 4.1c: what bounds a 7-point stencil product y[r] = sum_l c_l x[r + off_l]
// on the Yee numbering (offsets 0, +-3, +-3(N+1), +-3(N+1)^2; 20.6 M rows) once the matrix stream is gone?
//   v0  one thread per row, 7 global gathers                       (what k_spmv_pat does)
//   v1  the same with all offsets inside the row's own 2 KB window (every gather an L1 hit)
//   v2  two consecutive rows per thread, 16-byte gathers
//   v3  x windows of a 256-row block staged in LDS (5 windows), 7 LDS reads per row
//   v4  x windows of a TJ x TI tile of grid lines staged in LDS: each x entry fetched once per tile
// Build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_gather.hip -o scripts/bin/ubench_gather
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NXCD = 8;
__device__ inline long long xcd_block(long long w, long long nblocks) {
  const long long per = (nblocks + NXCD - 1) / NXCD;
  return (w % NXCD) * per + (w / NXCD);
}

struct Legs { long long off[7]; double c[7]; };

__global__ __launch_bounds__(256) void k_v0(const double* __restrict__ x, double* __restrict__ y, long long n, long long lo,
                                             long long hi, Legs L, long long nblocks) {
  const long long b = xcd_block(blockIdx.x, nblocks);
  if (b >= nblocks) return;
  const long long r = lo + b * 256 + threadIdx.x;
  if (r >= hi) return;
  double xv[7];
#pragma unroll
  for (int l = 0; l < 7; ++l) xv[l] = x[r + L.off[l]];
  double s = 0.0;
#pragma unroll
  for (int l = 0; l < 7; ++l) s += L.c[l] * xv[l];
  __builtin_nontemporal_store(s, y + r);
}

typedef double dbl2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(128) void k_v2(const double* __restrict__ x, double* __restrict__ y, long long n, long long lo,
                                             long long hi, Legs L, long long nblocks) {
  const long long b = xcd_block(blockIdx.x, nblocks);
  if (b >= nblocks) return;
  const long long r = lo + b * 256 + 2 * threadIdx.x;
  if (r + 1 >= hi) return;
  dbl2 xv[7];
#pragma unroll
  for (int l = 0; l < 7; ++l) {
    const double* p = x + r + L.off[l];
    // dword-aligned 16-byte global load (global_load_dwordx4 needs no more)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(xv[l]) : "v"(p) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int l = 0; l < 7; ++l) { s0 += L.c[l] * xv[l].x; s1 += L.c[l] * xv[l].y; }
  dbl2 o; o.x = s0; o.y = s1;
  __builtin_nontemporal_store(o, reinterpret_cast<dbl2*>(y + r));
}

// LDS windows of one 256-row block: [r0-3, r0+259) and the four far windows [r0+off, r0+off+256)
__global__ __launch_bounds__(256) void k_v3(const double* __restrict__ x, double* __restrict__ y, long long n, long long lo,
                                             long long hi, Legs L, long long nblocks) {
  __shared__ double w[5][264];
  const long long b = xcd_block(blockIdx.x, nblocks);
  if (b >= nblocks) return;
  const long long r0 = lo + b * 256;
  const int t = threadIdx.x;
  // legs: 0: 0, 1: -3, 2: +3, 3/4: -+y, 5/6: -+z
  w[0][t + 3] = x[r0 + t];
  if (t < 3) { w[0][t] = x[r0 - 3 + t]; w[0][259 + t] = x[r0 + 256 + t]; }
#pragma unroll
  for (int q = 0; q < 4; ++q) w[1 + q][t] = x[r0 + L.off[3 + q] + t];
  __syncthreads();
  const long long r = r0 + t;
  if (r >= hi) return;
  double s = 0.0;
  s += L.c[0] * w[0][t + 3];
  s += L.c[1] * w[0][t];
  s += L.c[2] * w[0][t + 6];
#pragma unroll
  for (int q = 0; q < 4; ++q) s += L.c[3 + q] * w[1 + q][t];
  __builtin_nontemporal_store(s, y + r);
}

// Tile of TJ lines (stride sy) x TI planes (stride sz), SEG rows of each line: self windows + y halo + z halo in LDS.
// Every x entry of the tile is fetched once; per line 1 + 2/TJ + 2/TI windows instead of 5.
template <int TJ, int TI, int SEG>
__global__ __launch_bounds__(256) void k_v4(const double* __restrict__ x, double* __restrict__ y, long long lo, long long sy,
                                             long long sz, int nseg, int ntj, int nti, Legs L, long long ntiles) {
  extern __shared__ double lds[];
  constexpr int WS = SEG + 8;                                   // window stride in LDS (3 halo entries each side, padded)
  const long long tile = xcd_block(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  const int seg = (int)(tile % nseg);
  const int tj = (int)((tile / nseg) % ntj);
  const int ti = (int)(tile / ((long long)nseg * ntj));
  const long long base = lo + (long long)ti * TI * sz + (long long)tj * TJ * sy + (long long)seg * SEG;
  // window (j, i), j in [-1, TJ], i in [-1, TI] without the four corners; index ((i+1) * (TJ+2) + (j+1))
  const int t = threadIdx.x;
  for (int wq = 0; wq < (TJ + 2) * (TI + 2); ++wq) {
    const int j = wq % (TJ + 2) - 1, i = wq / (TJ + 2) - 1;
    const bool edge_j = j < 0 || j >= TJ, edge_i = i < 0 || i >= TI;
    if (edge_j && edge_i) continue;
    const double* src = x + base + (long long)i * sz + (long long)j * sy - 4;
    double* dst = lds + (size_t)wq * WS;
    for (int q = t; q < WS; q += 256) dst[q] = src[q];
  }
  __syncthreads();
  for (int li = 0; li < TI; ++li)
    for (int lj = 0; lj < TJ; ++lj) {
      const double* c0 = lds + (size_t)((li + 1) * (TJ + 2) + (lj + 1)) * WS + 4;
      const double* ym = lds + (size_t)((li + 1) * (TJ + 2) + lj) * WS + 4;
      const double* yp = lds + (size_t)((li + 1) * (TJ + 2) + (lj + 2)) * WS + 4;
      const double* zm = lds + (size_t)(li * (TJ + 2) + (lj + 1)) * WS + 4;
      const double* zp = lds + (size_t)((li + 2) * (TJ + 2) + (lj + 1)) * WS + 4;
      for (int q = t; q < SEG; q += 256) {
        double s = 0.0;
        s += L.c[0] * c0[q];
        s += L.c[1] * c0[q - 3];
        s += L.c[2] * c0[q + 3];
        s += L.c[3] * ym[q];
        s += L.c[4] * yp[q];
        s += L.c[5] * zm[q];
        s += L.c[6] * zp[q];
        __builtin_nontemporal_store(s, y + base + (long long)li * sz + (long long)lj * sy + q);
      }
    }
}

// v5: prototype of the pipelined pair product.  Pair ids (2 B per pair of rows) and a pattern dictionary in global memory
// (PairEntry of mxg_pat.hip), persistent wavefronts that take steps of U x 63 pairs cyclically within their XCD's
// eighth of the rows; per step: ids of step j+2 and pattern of step j+1 are requested before the gathers of step j.
struct PairEntry { int delta[8]; double va[8]; double vb[8]; int len, pad[23]; };
struct PRegs { int dl[8]; double va[8], vb[8]; int len; };
typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__device__ inline void pload(const PairEntry* e, PRegs& R) {
  const u4 d0 = *reinterpret_cast<const u4*>(e->delta), d1 = *reinterpret_cast<const u4*>(e->delta + 4);
  R.dl[0] = d0.x; R.dl[1] = d0.y; R.dl[2] = d0.z; R.dl[3] = d0.w; R.dl[4] = d1.x; R.dl[5] = d1.y; R.dl[6] = d1.z; R.dl[7] = d1.w;
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const dbl2 a = *reinterpret_cast<const dbl2*>(e->va + i), b = *reinterpret_cast<const dbl2*>(e->vb + i);
    R.va[i] = a.x; R.va[i + 1] = a.y; R.vb[i] = b.x; R.vb[i + 1] = b.y;
  }
  R.len = e->len;
}
template <int U, bool PIPE>
__global__ __launch_bounds__(256) void k_v5(const unsigned short* __restrict__ pairid, const PairEntry* __restrict__ dict,
                                             const double* __restrict__ x, double* __restrict__ y, long long p_lo,
                                             long long nsteps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x % NXCD;
  const long long wx = (long long)(blockIdx.x / NXCD) * 4 + wave, Wx = (long long)(gridDim.x / NXCD) * 4;
  const long long per = (nsteps + NXCD - 1) / NXCD, s_lo = xcd * per, s_hi = min(nsteps, s_lo + per);
  const bool live = lane < 63;
  auto pair_of = [&](long long s, int u) { return p_lo + (s * U + u) * 63 + lane; };
  long long s = s_lo + wx;
  unsigned pid1[U], pid2[U];                       // ids of the next step / the one after
  PRegs R, Rn;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    pid1[u] = (live && s < s_hi) ? pairid[pair_of(s, u)] : 0xFFFFu;
    pid2[u] = (live && s + Wx < s_hi) ? pairid[pair_of(s + Wx, u)] : 0xFFFFu;
  }
  if (pid1[0] != 0xFFFFu) pload(dict + pid1[0], Rn);
  for (; s < s_hi; s += Wx) {
    unsigned pid0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { pid0[u] = pid1[u]; pid1[u] = pid2[u]; }
    R = Rn;
    if (PIPE) {
#pragma unroll
      for (int u = 0; u < U; ++u) pid2[u] = (live && s + 2 * Wx < s_hi) ? pairid[pair_of(s + 2 * Wx, u)] : 0xFFFFu;
      if (pid1[0] != 0xFFFFu) pload(dict + pid1[0], Rn);          // pattern of the next step, in flight during this one
    }
    dbl2 xv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = pid0[u] != 0xFFFFu;
      const long long r = 2 * pair_of(s, u);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const long long cc = ok ? r + R.dl[i] : 0;
        const dbl2u t = *reinterpret_cast<const dbl2u*>(x + cc);
        xv[u][i].x = t.x; xv[u][i].y = t.y;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (pid0[u] != 0xFFFFu) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int i = 0; i < 7; ++i) { sa += R.va[i] * xv[u][i].x; sb += R.vb[i] * xv[u][i].y; }
        dbl2 o; o.x = sa; o.y = sb;
        __builtin_nontemporal_store(o, reinterpret_cast<dbl2*>(y + 2 * pair_of(s, u)));
      }
    if (!PIPE) {
#pragma unroll
      for (int u = 0; u < U; ++u) pid2[u] = (live && s + 2 * Wx < s_hi) ? pairid[pair_of(s + 2 * Wx, u)] : 0xFFFFu;
      if (pid1[0] != 0xFFFFu) pload(dict + pid1[0], Rn);
    }
  }
}

// v6: the tile product of mxg_pat.hip on synthetic tiles that cover every row: descriptor {start, rows, pattern} and the
// pattern through scalar loads, two rows per lane, 63 lanes per tile.  MODE 0: as in the library; 1: descriptor
// computed instead of loaded (start = lo + t * 126); 2: also the pattern from kernel arguments (no dependent scalar load).
struct TileDesc { int start, nrows, pid, pad; };
struct TileEntry { int len, pad0[7]; int delta[8]; double v[3][8]; };
template <int MODE, int WAVES, int TROWS = 126, bool NT = false>
__global__ __launch_bounds__(64 * WAVES) void k_v6(const TileDesc* __restrict__ desc, const TileEntry* __restrict__ dict, long long ntiles,
                                                    long long ntb, const double* __restrict__ x, double* __restrict__ y, long long lo, Legs L) {
  const long long tb = xcd_block(blockIdx.x, ntb);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const long long t = tb * WAVES + wave;
  if (tb >= ntb || t >= ntiles) return;
  TileDesc d;
  if (MODE == 0) d = desc[t]; else { d.start = (int)(lo + t * TROWS); d.nrows = TROWS; d.pid = (int)(t % 3); }
  const TileEntry* e = dict + d.pid;
  const int len = MODE == 2 ? 7 : e->len;
  if (2 * lane >= d.nrows) return;
  const int ca = (2 * lane) % 3, cb = (2 * lane + 1) % 3;
  dbl2 xv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < len) {
      const long long off = MODE == 2 ? L.off[i] : (long long)e->delta[i];
      const double* xp = x + ((long long)d.start + off);
      const dbl2u tt = *reinterpret_cast<const dbl2u*>(xp + 2 * lane);
      xv[i].x = tt.x; xv[i].y = tt.y;
    }
  double sa = 0.0, sb = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < len) {
      double v0, v1, v2;
      if (MODE == 2) { v0 = v1 = v2 = L.c[i]; } else { v0 = e->v[0][i]; v1 = e->v[1][i]; v2 = e->v[2][i]; }
      const double va = ca == 0 ? v0 : (ca == 1 ? v1 : v2), vb = cb == 0 ? v0 : (cb == 1 ? v1 : v2);
      sa += va * xv[i].x; sb += vb * xv[i].y;
    }
  if (NT) { dbl2 o; o.x = sa; o.y = sb; __builtin_nontemporal_store(o, reinterpret_cast<dbl2*>(y + (long long)d.start + 2 * lane)); }
  else { dbl2u o; o.x = sa; o.y = sb; *reinterpret_cast<dbl2u*>(y + (long long)d.start + 2 * lane) = o; }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 189;                 // 3 * 190^3 = 20.6 M rows
  const long long M = N + 1, sy = 3 * M, sz = 3 * M * M, n = 3 * M * M * M;
  double *x, *y;
  CHECK(hipMalloc(&x, sizeof(double) * (n + 4096)));
  CHECK(hipMalloc(&y, sizeof(double) * (n + 4096)));
  std::vector<double> hx(n);
  for (long long i = 0; i < n; ++i) hx[i] = (double)((i * 2654435761u) & 1023) / 1024.0;
  CHECK(hipMemcpy(x, hx.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  CHECK(hipMemset(y, 0, sizeof(double) * n));
  Legs far{{0, -3, 3, -sy, sy, -sz, sz}, {6.0, -1.0, -1.0, -1.0, -1.0, -1.0, -1.0}};
  Legs near{{0, -3, 3, -2, 2, -1, 1}, {6.0, -1.0, -1.0, -1.0, -1.0, -1.0, -1.0}};
  const long long lo = sz + sy + 256, hi = n - sz - sy - 256;   // interior rows only
  const long long rows = hi - lo, nb = rows / 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch, double rowsdone) {
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    std::vector<double> hy(4);
    printf("%-44s %8.4f ms  %6.1f Mrows  %7.1f GB/s of (x + y) bytes  scaled to 20.6 M rows: %.4f ms\n", name, ms, rowsdone / 1e6,
           rowsdone * 16 / ms / 1e6, ms * 20608881.0 / rowsdone);
    fflush(stdout);
  };
  const unsigned grid = (unsigned)(((nb + NXCD - 1) / NXCD) * NXCD);
  run("v0 7 global gathers per row", [&] { hipLaunchKernelGGL(k_v0, dim3(grid), dim3(256), 0, 0, x, y, n, lo, hi, far, nb); }, nb * 256.0);
  run("v1 7 gathers, all in the own window (L1)", [&] { hipLaunchKernelGGL(k_v0, dim3(grid), dim3(256), 0, 0, x, y, n, lo, hi, near, nb); }, nb * 256.0);
  run("v2 two rows per thread, 16-byte gathers", [&] { hipLaunchKernelGGL(k_v2, dim3(grid), dim3(128), 0, 0, x, y, n, lo, hi, far, nb); }, nb * 256.0);
  run("v3 5 windows of a 256-row block in LDS", [&] { hipLaunchKernelGGL(k_v3, dim3(grid), dim3(256), 0, 0, x, y, n, lo, hi, far, nb); }, nb * 256.0);
  // reference check of v3 against v0 on a few rows
  {
    std::vector<double> a(1024), b(1024);
    hipLaunchKernelGGL(k_v0, dim3(grid), dim3(256), 0, 0, x, y, n, lo, hi, far, nb);
    CHECK(hipMemcpy(a.data(), y + lo + 12345, sizeof(double) * 1024, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_v3, dim3(grid), dim3(256), 0, 0, x, y, n, lo, hi, far, nb);
    CHECK(hipMemcpy(b.data(), y + lo + 12345, sizeof(double) * 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += a[i] != b[i];
    printf("v3 vs v0 mismatches: %d\n", bad);
    hipLaunchKernelGGL(k_v2, dim3(grid), dim3(128), 0, 0, x, y, n, lo, hi, far, nb);
    CHECK(hipMemcpy(b.data(), y + lo + 12345, sizeof(double) * 1024, hipMemcpyDeviceToHost));
    bad = 0; for (int i = 0; i < 1024; ++i) bad += a[i] != b[i];
    printf("v2 vs v0 mismatches: %d\n", bad);
  }
  auto tiles = [&](auto kern, int TJ, int TI, int SEG, const char* name) {
    // interior box of lines: j in [1, M-1), i in [1, M-1); each line 3*M rows, SEG-row segments (the last partial one skipped)
    const int nseg = (int)((3 * M - 16) / SEG), ntj = (int)((M - 2) / TJ), nti = (int)((M - 2) / TI);
    const long long nt = (long long)nseg * ntj * nti;
    const long long tlo = sz + sy + 8;
    const size_t shm = sizeof(double) * (size_t)(TJ + 2) * (TI + 2) * (SEG + 8);
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) {
      printf("%s: %zu bytes of LDS refused\n", name, shm); (void)hipGetLastError(); return;
    }
    const unsigned g = (unsigned)(((nt + NXCD - 1) / NXCD) * NXCD);
    run(name, [&] { hipLaunchKernelGGL(kern, dim3(g), dim3(256), shm, 0, x, y, tlo, sy, sz, nseg, ntj, nti, far, nt); },
        (double)nt * TJ * TI * SEG);
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    // check against v0 on the first tile's first line
    std::vector<double> a(SEG), b(SEG);
    CHECK(hipMemcpy(b.data(), y + tlo, sizeof(double) * SEG, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_v0, dim3(grid), dim3(256), 0, 0, x, y, n, lo, hi, far, nb);
    CHECK(hipMemcpy(a.data(), y + tlo, sizeof(double) * SEG, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < SEG; ++i) bad += a[i] != b[i];
    printf("   mismatches against v0 on %d rows: %d\n", SEG, bad);
  };
  tiles(k_v4<2, 2, 512>, 2, 2, 512, "v4 tile 2 x 2 lines x 512 rows (LDS 66 KB)");
  tiles(k_v4<4, 4, 256>, 4, 4, 256, "v4 tile 4 x 4 lines x 256 rows (LDS 76 KB)");
  tiles(k_v4<4, 2, 256>, 4, 2, 256, "v4 tile 4 x 2 lines x 256 rows (LDS 51 KB)");
  tiles(k_v4<2, 2, 256>, 2, 2, 256, "v4 tile 2 x 2 lines x 256 rows (LDS 34 KB)");
  tiles(k_v4<8, 1, 256>, 8, 1, 256, "v4 tile 8 x 1 lines x 256 rows (LDS 63 KB)");
  // v5
  {
    const long long p_lo = (lo + 1) / 2, npairs = (hi - lo) / 2 - 8;
    std::vector<unsigned short> hp(n / 2 + 4096);
    for (size_t i = 0; i < hp.size(); ++i) hp[i] = (unsigned short)(i % 3);
    std::vector<PairEntry> hd(3);
    for (int k = 0; k < 3; ++k) {
      PairEntry e{};
      e.len = 7;
      for (int i = 0; i < 7; ++i) { e.delta[i] = (int)far.off[i]; e.va[i] = far.c[i]; e.vb[i] = far.c[i]; }
      hd[k] = e;
    }
    unsigned short* dp; PairEntry* dd;
    CHECK(hipMalloc(&dp, hp.size() * 2)); CHECK(hipMalloc(&dd, sizeof(PairEntry) * 3));
    CHECK(hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dd, hd.data(), sizeof(PairEntry) * 3, hipMemcpyHostToDevice));
    auto v5 = [&](auto kern, int U, int wgs_per_cu, const char* name) {
      const long long nsteps = npairs / (63 * U);
      const unsigned g = 256 * wgs_per_cu;
      char buf[128]; snprintf(buf, sizeof buf, "%s, %d WGs/CU", name, wgs_per_cu);
      run(buf, [&] { hipLaunchKernelGGL(kern, dim3(g), dim3(256), 0, 0, dp, dd, x, y, p_lo, nsteps); }, (double)nsteps * 63 * U * 2);
      std::vector<double> a(512), b(512);
      CHECK(hipMemcpy(b.data(), y + 2 * p_lo + 126 * 40, sizeof(double) * 512, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL(k_v0, dim3(grid), dim3(256), 0, 0, x, y, n, lo, hi, far, nb);
      CHECK(hipMemcpy(a.data(), y + 2 * p_lo + 126 * 40, sizeof(double) * 512, hipMemcpyDeviceToHost));
      int bad = 0; for (int i = 0; i < 512; ++i) bad += a[i] != b[i];
      printf("   mismatches against v0 on 512 rows: %d\n", bad);
    };
    for (int w : {2, 3, 4}) {
      v5(k_v5<2, true>, 2, w, "v5 pipelined pair product, U = 2");
      v5(k_v5<2, false>, 2, w, "v5 the same without the pipeline, U = 2");
    }
    for (int w : {2, 3}) v5(k_v5<3, true>, 3, w, "v5 pipelined pair product, U = 3");
    v5(k_v5<1, true>, 1, 4, "v5 pipelined pair product, U = 1");
    v5(k_v5<1, true>, 1, 6, "v5 pipelined pair product, U = 1");
  }
  // v6
  {
    const long long ntiles = (hi - lo - 256) / 126;
    std::vector<TileDesc> ht(ntiles);
    for (long long t = 0; t < ntiles; ++t) { ht[t].start = (int)(lo + t * 126); ht[t].nrows = 126; ht[t].pid = (int)(t % 3); ht[t].pad = 0; }
    std::vector<TileEntry> hd(3);
    for (int k = 0; k < 3; ++k) {
      TileEntry e{}; e.len = 7;
      for (int i = 0; i < 7; ++i) { e.delta[i] = (int)far.off[i]; for (int c = 0; c < 3; ++c) e.v[c][i] = far.c[i]; }
      hd[k] = e;
    }
    TileDesc* dt; TileEntry* dd;
    CHECK(hipMalloc(&dt, sizeof(TileDesc) * ntiles)); CHECK(hipMalloc(&dd, sizeof(TileEntry) * 3));
    CHECK(hipMemcpy(dt, ht.data(), sizeof(TileDesc) * ntiles, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dd, hd.data(), sizeof(TileEntry) * 3, hipMemcpyHostToDevice));
    auto v6 = [&](auto kern, int waves, const char* name) {
      const long long ntb = (ntiles + waves - 1) / waves;
      const unsigned g = (unsigned)(((ntb + NXCD - 1) / NXCD) * NXCD);
      run(name, [&] { hipLaunchKernelGGL(kern, dim3(g), dim3(64 * waves), 0, 0, dt, dd, ntiles, ntb, x, y, lo, far); }, (double)ntiles * 126);
      std::vector<double> a(504), b(504);
      CHECK(hipMemcpy(b.data(), y + lo + 126 * 50, sizeof(double) * 504, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL(k_v0, dim3(grid), dim3(256), 0, 0, x, y, n, lo, hi, far, nb);
      CHECK(hipMemcpy(a.data(), y + lo + 126 * 50, sizeof(double) * 504, hipMemcpyDeviceToHost));
      int bad = 0; for (int i = 0; i < 504; ++i) bad += a[i] != b[i];
      if (bad) printf("   mismatches against v0 on 504 rows: %d\n", bad);
    };
    auto v6b = [&](auto kern, int waves, int trows, const char* name) {
      const long long nt2 = (hi - lo - 256) / trows, ntb = (nt2 + waves - 1) / waves;
      const unsigned g = (unsigned)(((ntb + NXCD - 1) / NXCD) * NXCD);
      run(name, [&] { hipLaunchKernelGGL(kern, dim3(g), dim3(64 * waves), 0, 0, dt, dd, nt2, ntb, x, y, lo, far); }, (double)nt2 * trows);
    };
    v6b(k_v6<2, 4, 128, false>, 4, 128, "v6 computed, 128-row tiles (64 lanes, aligned)");
    v6b(k_v6<2, 4, 128, true>, 4, 128, "v6 computed, 128-row tiles, nontemporal store");
    v6b(k_v6<2, 4, 126, true>, 4, 126, "v6 computed, 126-row tiles, nontemporal store");
    v6b(k_v6<2, 2, 128, true>, 2, 128, "v6 computed, 128-row tiles, nontemporal store, 2 waves/WG");
    v6(k_v6<0, 4>, 4, "v6 tile product, descriptor + pattern loaded, 4 waves/WG");
    v6(k_v6<1, 4>, 4, "v6 descriptor computed, pattern loaded");
    v6(k_v6<2, 4>, 4, "v6 descriptor computed, pattern in kernel args");
    v6(k_v6<0, 2>, 2, "v6 tile product, 2 waves/WG");
    v6(k_v6<0, 8>, 8, "v6 tile product, 8 waves/WG");
    v6(k_v6<0, 1>, 1, "v6 tile product, 1 wave/WG");
  }
  return 0;
}