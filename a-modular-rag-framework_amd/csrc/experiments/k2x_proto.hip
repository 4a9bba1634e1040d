// EXPERIMENT (not part of libmrag_hip.so): main loop of a 4-wave / 512-register similarity GEMM, 384 corpus rows x 256
// queries per workgroup with the accumulators spilling into AGPRs (384 per lane), BK = 32, three LDS stages.
// Question it answers: does a larger tile (-17 % L2->LDS bytes per flop) with ONE wave per SIMD beat the shipped
// 8-wave 256 x 256 x 64 loop (main loop alone: 11.85 ms at 10 000 x 1 M x 768)?  No top-k epilogue here:
// the accumulators are only folded into one value per workgroup so that nothing is optimised away.
//   build: hipcc -O3 --offload-arch=gfx950 -o k2x_proto k2x_proto.hip ;  run: ./k2x_proto [nq n d iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <type_traits>
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

constexpr int TM = 384, TQ = 256, BK = 32, NS = 3;
constexpr int A_BYTES = TM * BK * 2;            // 24 KiB
constexpr int B_BYTES = TQ * BK * 2;            // 16 KiB
constexpr int STAGE = A_BYTES + B_BYTES;        // 40 KiB
constexpr int MF = 12, NF = 8;                  // per wave: 192 rows x 128 queries
#ifndef SPREAD
#define SPREAD 0   // 1: the 10 DMA issues of a stage between the MFMA rows instead of at the top of the step (measured worse: 19.4 vs 15.0 ms)
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k2x_kernel(const uint16_t* __restrict__ corpus, const uint16_t* __restrict__ queries, int ld, int n_ctiles, int T, int S,
                float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  // XCD-aware (t, s) map as in the shipped kernel
  const int per = (int)gridDim.x >> 3;
  const int lin = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  const int t = lin / S, s = lin - t * S;
  const int tile_lo = (int)(((long long)s * n_ctiles) / S), tile_hi = (int)(((long long)(s + 1) * n_ctiles) / S);
  const int ksteps = ld / BK;
  // DMA pieces: one instruction = 16 rows x 64 B = 1 KiB contiguous in LDS; lane -> row (lane >> 2), 16-byte chunk (lane & 3).
  // A part: 24 pieces, B part: 16 pieces per stage; wave w takes pieces w, w + 4, ...
  const uint32_t voff = (uint32_t)(lane >> 2) * (uint32_t)ld * 2u + (uint32_t)(lane & 3) * 16u;
  const char* qbase = (const char*)(queries + (size_t)t * TQ * ld);
  // piece i (0..9) of this wave's share of a stage: 0..5 corpus, 6..9 queries
  auto stage_piece = [&](int tile, int kk, int buf, int i) {
    char* la = sm + buf * STAGE;
    if (i < 6) {
      const char* abase = (const char*)(corpus + (size_t)tile * TM * ld) + (size_t)kk * BK * 2;
      const int piece = w + 4 * i;
      __builtin_amdgcn_global_load_lds((glb_vptr)(abase + (size_t)piece * 16 * ld * 2 + voff), (lds_vptr)(la + piece * 1024), 16, 0, 0);
    } else {
      const char* bbase = qbase + (size_t)kk * BK * 2;
      const int piece = w + 4 * (i - 6);
      __builtin_amdgcn_global_load_lds((glb_vptr)(bbase + (size_t)piece * 16 * ld * 2 + voff), (lds_vptr)(la + A_BYTES + piece * 1024), 16, 0, 0);
    }
  };
  auto stage = [&](int tile, int kk, int buf) {
#pragma unroll
    for (int i = 0; i < 10; ++i) stage_piece(tile, kk, buf, i);
  };
  // fragment reads: 16 rows x 64 B = one contiguous KiB; lane -> row (lane & 15), 16-byte chunk (lane >> 4)
  const int frd = (lane & 15) * 64 + (lane >> 4) * 16;
  const int a_rd = (wm * 192) * 64 + frd, b_rd = A_BYTES + (wn * 128) * 64 + frd;
  f32x4 acc[MF][NF];
  const int n_tiles = tile_hi - tile_lo;
  const int n_steps = n_tiles * ksteps;
  int st_tile = tile_lo, st_kk = 0, st_buf = 0, staged = 0;
  auto stage_next = [&]() {
    stage(st_tile, st_kk, st_buf);
    if (++st_kk == ksteps) { st_kk = 0; ++st_tile; }
    if (++st_buf == NS) st_buf = 0;
    ++staged;
  };
  stage_next();
  if (n_steps > 1) stage_next();
  int buf = 0, step = 0;
  float sink = 0.f;
  // one K step: wait for its stage, barrier, issue the stage two ahead, 20 fragment reads + 96 MFMAs.
  // FIRST selects the zero-C form (first K step of a tile).  Accumulators of the first 8 row fragments are pinned in AGPRs,
  // the last 4 in VGPRs (hipcc spills a 384-register accumulator array it is free to place: 577 spilled VGPRs).
  auto kstep = [&](auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    if (staged - step - 1 >= 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool do_stage = staged < n_steps;     // its 10 DMA issues go between the MFMA rows below (SPREAD) or all here
    const int s_tile = st_tile, s_kk = st_kk, s_buf = st_buf;
    if (do_stage) {
      if (!SPREAD) stage(s_tile, s_kk, s_buf);
      if (++st_kk == ksteps) { st_kk = 0; ++st_tile; }
      if (++st_buf == NS) st_buf = 0;
      ++staged;
    }
    const char* sb = sm + buf * STAGE;
    f16x8 bf[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) bf[nf] = *(const f16x8*)(sb + b_rd + nf * 1024);
    f16x8 af[2];
    af[0] = *(const f16x8*)(sb + a_rd);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      if (mf + 1 < MF) af[(mf + 1) & 1] = *(const f16x8*)(sb + a_rd + (mf + 1) * 1024);
      if (SPREAD && do_stage && mf < 10) stage_piece(s_tile, s_kk, s_buf, mf);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        if (FIRST) {
          if (mf < 8) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc[mf][nf]) : "v"(af[mf & 1]), "v"(bf[nf]));
          else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(acc[mf][nf]) : "v"(af[mf & 1]), "v"(bf[nf]));
        } else {
          if (mf < 8) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[mf][nf]) : "v"(af[mf & 1]), "v"(bf[nf]));
          else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[mf][nf]) : "v"(af[mf & 1]), "v"(bf[nf]));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (++buf == NS) buf = 0;
    ++step;
  };
  for (int tile = 0; tile < n_tiles; ++tile) {
    kstep(std::true_type{});
    for (int kk = 1; kk < ksteps; ++kk) kstep(std::false_type{});
    // stand-in for the top-k epilogue: fold the tile into one number (keeps the accumulators live)
    float x = 0.f;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) x = fmaxf(x, fmaxf(fmaxf(acc[mf][nf][0], acc[mf][nf][1]), fmaxf(acc[mf][nf][2], acc[mf][nf][3])));
    sink = fmaxf(sink, x);
  }
  if (sink == 123456.789f) out[blockIdx.x] = sink;
  if (tid == 0 && sink > 1e30f) out[0] = sink;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int nq = argc > 1 ? atoi(argv[1]) : 10000, n = argc > 2 ? atoi(argv[2]) : 1000000, d = argc > 3 ? atoi(argv[3]) : 768;
  const int iters = argc > 4 ? atoi(argv[4]) : 10;
  const int T = (nq + TQ - 1) / TQ, n_ctiles = (n + TM - 1) / TM;
  int S = 256 / T; if (S < 1) S = 1;
  while ((T * S) % 8) --S;
  const size_t crow = (size_t)n_ctiles * TM, qrow = (size_t)T * TQ;
  std::vector<uint16_t> h(1 << 20);
  for (auto& v : h) { const float f = ((rand() % 2001) - 1000) / 30000.f; _Float16 x = (_Float16)f; v = *(uint16_t*)&x; }
  uint16_t *dc, *dq; float* dout;
  CK(hipMalloc(&dc, crow * d * 2)); CK(hipMalloc(&dq, qrow * d * 2)); CK(hipMalloc(&dout, 4096));
  for (size_t off = 0; off < crow * d; off += h.size()) CK(hipMemcpy(dc + off, h.data(), std::min(h.size(), crow * d - off) * 2, hipMemcpyHostToDevice));
  for (size_t off = 0; off < qrow * d; off += h.size()) CK(hipMemcpy(dq + off, h.data() + 12345, std::min(h.size() - 12345, qrow * d - off) * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)k2x_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k2x_kernel, dim3(T * S), dim3(256), NS * STAGE, 0, dc, dq, d, n_ctiles, T, S, dout);
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0.f;
  for (int i = 0; i < iters; ++i) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k2x_kernel, dim3(T * S), dim3(256), NS * STAGE, 0, dc, dq, d, n_ctiles, T, S, dout);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
  }
  const double flop = 2.0 * nq * (double)n * d;
  printf("k2x proto %d x %d x %d: T=%d S=%d (%d workgroups, %d-row corpus tiles): mean %.3f ms, min %.3f ms = %.1f TFLOP/s (algorithmic)\n",
         nq, n, d, T, S, T * S, TM, sum / iters, best, flop / (sum / iters) / 1e9);
  return 0;
}
