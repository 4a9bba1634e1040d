// Microbenchmark (development tool, not product code): what the output-store phase of the encoder GEMMs can expect from the
// memory system.  hipcc --offload-arch=gfx950 -O3 store_bw.hip -o store_bw
//   modes: 0 contiguous 16-B stores | 1 tile pattern (256 rows x 512 B at a row pitch) | 2 contiguous loads | 3 half the
//   workgroups load, half store | policy: 0 plain 1 nt 2 sc1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int POL>
__device__ __forceinline__ void st16(char* p, u32x4 v) {
  if (POL == 1) __builtin_nontemporal_store(v, (u32x4*)p);
  else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *(u32x4*)p = v;
}

// every workgroup owns `per_wg` bytes of the footprint (wrapping inside `foot` bytes) and sweeps them `reps` times
template <int POL>
__global__ __launch_bounds__(512) void k_store(char* buf, size_t foot, size_t per_wg, int reps) {
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  for (int r = 0; r < reps; ++r) {
    size_t base = ((size_t)blockIdx.x * per_wg + (size_t)r * per_wg * gridDim.x) % foot;
    for (size_t o = (size_t)threadIdx.x * 16; o < per_wg; o += 512 * 16) st16<POL>(buf + (base + o) % foot, v);
  }
}
// tile pattern: a workgroup writes tiles of 256 rows x 512 B, row pitch `pitch`, tile t of the footprint
template <int POL>
__global__ __launch_bounds__(512) void k_tile(char* buf, size_t rows_total, int pitch, int tiles_n, int n_tiles, int reps) {
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  const int piece = threadIdx.x & 31, r0 = threadIdx.x >> 5;   // 16 rows per sweep of the 512 threads
  for (int r = 0; r < reps; ++r) {
    const int t = (blockIdx.x + r * gridDim.x) % n_tiles;
    const int tm = t / tiles_n, tn = t % tiles_n;
    char* base = buf + (size_t)tm * 256 * pitch + (size_t)tn * 512;
    for (int i = 0; i < 16; ++i) st16<POL>(base + (size_t)(r0 + 16 * i) * pitch + piece * 16, v);
  }
}
// only the workgroups of `nx` XCDs store (blockIdx % 8 < nx), `per_x` of each: per-XCD or chip-wide limit?
template <int POL>
__global__ __launch_bounds__(512) void k_tile_x(char* buf, int pitch, int tiles_n, int n_tiles, int reps, int nx, int per_x) {
  if ((int)(blockIdx.x & 7) >= nx || (int)(blockIdx.x >> 3) >= per_x) return;
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  const int piece = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  for (int r = 0; r < reps; ++r) {
    const int t = (blockIdx.x + r * gridDim.x) % n_tiles;
    const int tm = t / tiles_n, tn = t % tiles_n;
    char* base = buf + (size_t)tm * 256 * pitch + (size_t)tn * 512;
    for (int i = 0; i < 16; ++i) st16<POL>(base + (size_t)(r0 + 16 * i) * pitch + piece * 16, v);
  }
}
__global__ __launch_bounds__(512) void k_load(const char* buf, size_t foot, size_t per_wg, int reps, unsigned* sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    size_t base = ((size_t)blockIdx.x * per_wg + (size_t)r * per_wg * gridDim.x) % foot;
    for (size_t o = (size_t)threadIdx.x * 16; o < per_wg; o += 512 * 16) acc += __builtin_nontemporal_load((const u32x4*)(buf + (base + o) % foot));
  }
  if (acc[0] == 0x12345u) sink[0] = acc[1];
}
template <int POL>
__global__ __launch_bounds__(512) void k_mixed(char* wbuf, const char* rbuf, size_t foot, size_t per_wg, int reps, unsigned* sink) {
  const int half = gridDim.x / 2;
  if ((int)blockIdx.x < half) {
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
      size_t base = ((size_t)blockIdx.x * per_wg + (size_t)r * per_wg * half) % foot;
      for (size_t o = (size_t)threadIdx.x * 16; o < per_wg; o += 512 * 16) acc += __builtin_nontemporal_load((const u32x4*)(rbuf + (base + o) % foot));
    }
    if (acc[0] == 0x12345u) sink[0] = acc[1];
  } else {
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int r = 0; r < reps; ++r) {
      size_t base = ((size_t)(blockIdx.x - half) * per_wg + (size_t)r * per_wg * half) % foot;
      for (size_t o = (size_t)threadIdx.x * 16; o < per_wg; o += 512 * 16) st16<POL>(wbuf + (base + o) % foot, v);
    }
  }
}

template <typename F>
static double time_ms(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}

int main() {
  const size_t big = (size_t)4 << 30;
  char *w, *rd; unsigned* sink;
  CK(hipMalloc(&w, big)); CK(hipMalloc(&rd, big)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(w, 1, big)); CK(hipMemset(rd, 2, big));
  const int wg = 256;
  const size_t per_wg = 128 << 10;   // one output tile
  const char* pol[3] = {"plain", "nt", "sc1"};
  for (size_t foot : {(size_t)16 << 20, (size_t)128 << 20, (size_t)4 << 30}) {
    const int reps = 64;
    const double bytes = (double)per_wg * wg * reps;
    double t0 = time_ms([&] { hipLaunchKernelGGL(k_store<0>, dim3(wg), dim3(512), 0, 0, w, foot, per_wg, reps); });
    double t1 = time_ms([&] { hipLaunchKernelGGL(k_store<1>, dim3(wg), dim3(512), 0, 0, w, foot, per_wg, reps); });
    double t2 = time_ms([&] { hipLaunchKernelGGL(k_store<2>, dim3(wg), dim3(512), 0, 0, w, foot, per_wg, reps); });
    double tl = time_ms([&] { hipLaunchKernelGGL(k_load, dim3(wg), dim3(512), 0, 0, rd, foot, per_wg, reps, sink); });
    printf("footprint %5zu MiB: contiguous stores plain %.2f nt %.2f sc1 %.2f TB/s | loads %.2f TB/s\n", foot >> 20, bytes / t0 / 1e9, bytes / t1 / 1e9,
           bytes / t2 / 1e9, bytes / tl / 1e9);
    double m0 = time_ms([&] { hipLaunchKernelGGL(k_mixed<1>, dim3(wg), dim3(512), 0, 0, w, rd, foot, per_wg, 2 * reps, sink); });
    printf("                    half the workgroups load, half store (nt): %.2f TB/s each way\n", bytes / m0 / 1e9);
  }
  // the GEMM epilogue's pattern: 256-row x 512-B tiles of a [262144][N] fp16 tensor
  for (int N : {768, 2304, 3072}) {
    const int pitch = N * 2, tiles_n = N / 256, tiles_m = 1024, n_tiles = tiles_m * tiles_n;
    const int reps = n_tiles / wg;
    const double bytes = (double)n_tiles * per_wg;
    for (int p = 0; p < 3; ++p) {
      double t = time_ms([&] {
        if (p == 0) hipLaunchKernelGGL(k_tile<0>, dim3(wg), dim3(512), 0, 0, w, (size_t)262144, pitch, tiles_n, n_tiles, reps);
        if (p == 1) hipLaunchKernelGGL(k_tile<1>, dim3(wg), dim3(512), 0, 0, w, (size_t)262144, pitch, tiles_n, n_tiles, reps);
        if (p == 2) hipLaunchKernelGGL(k_tile<2>, dim3(wg), dim3(512), 0, 0, w, (size_t)262144, pitch, tiles_n, n_tiles, reps);
      });
      printf("tile pattern N = %4d (%s): %.3f ms, %.2f TB/s\n", N, pol[p], t, bytes / t / 1e9);
    }
  }
  // how fast can FEW workgroups store (is the store rate a per-CU limit or a chip-wide one)?
  for (int nwg : {8, 16, 32, 64, 128, 256}) {
    const int N = 2304, pitch = N * 2, tiles_n = N / 256, n_tiles = 1024 * tiles_n;
    const int reps = 16;
    const double bytes = (double)nwg * reps * per_wg;
    double t = time_ms([&] { hipLaunchKernelGGL(k_tile<1>, dim3(nwg), dim3(512), 0, 0, w, (size_t)262144, pitch, tiles_n, n_tiles, reps); });
    double tp = time_ms([&] { hipLaunchKernelGGL(k_tile<0>, dim3(nwg), dim3(512), 0, 0, w, (size_t)262144, pitch, tiles_n, n_tiles, reps); });
    printf("%3d workgroups, tile pattern N = 2304: nt %.1f GB/s per workgroup (%.2f TB/s), plain %.1f GB/s per workgroup\n", nwg, bytes / t / 1e6 / nwg,
           bytes / t / 1e9, bytes / tp / 1e6 / nwg);
  }
  for (int nx : {1, 2, 4, 8})
    for (int per_x : {4, 8, 16, 32}) {
      const int N = 2304, pitch = N * 2, tiles_n = N / 256, n_tiles = 1024 * tiles_n;
      const int reps = 16;
      const double bytes = (double)nx * per_x * reps * per_wg;
      double t = time_ms([&] { hipLaunchKernelGGL(k_tile_x<1>, dim3(256), dim3(512), 0, 0, w, pitch, tiles_n, n_tiles, reps, nx, per_x); });
      printf("%d XCDs x %2d workgroups storing (nt): %.1f GB/s per workgroup, %.2f TB/s per XCD, %.2f TB/s\n", nx, per_x, bytes / t / 1e6 / (nx * per_x),
             bytes / t / 1e9 / nx, bytes / t / 1e9);
    }
  return 0;
}
