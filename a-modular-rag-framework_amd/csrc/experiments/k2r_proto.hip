// EXPERIMENT (not part of libmrag_hip.so): K2's main loop (8 waves, 256 corpus rows x 256 queries per workgroup, same
// XCD-aware (query tile, corpus split) map, no top-k epilogue) with the operand stages as a RING of finer stages behind
// COUNTED vmcnt waits, against the shipped structure (two 64-KiB stages, the next one issued during a K step and waited
// for with vmcnt(0) at its end).
// Question (round 3): the shipped loop's L2->LDS stream runs at 47-57 GB/s per CU with <= 64 KiB in flight that must
// all land before every barrier.  Does a ring that keeps 64-96 KiB in flight CONTINUOUSLY (4 x 32 KiB stages of
// K = 32, three of them in flight) move the same bytes faster -- and do 64-byte row segments cost L2->L1 bandwidth?
//   RB  = bytes of a row per stage (128: K = 64 per stage, 64: K = 32 per stage)
//   NS  = ring depth (stages of 256 x RB x 2 operands bytes); NS - 1 stages are in flight behind the one being read
//   MODE 0 = loads only, 1 = loads + fragment reads + MFMA (the main loop)
//   build: hipcc -O3 --offload-arch=gfx950 -DRB=64 -DNS=4 -DMODE=1 -o k2r_proto k2r_proto.hip ; run: ./k2r_proto [nq n d iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <type_traits>

#ifndef RB
#define RB 64
#endif
#ifndef NS
#define NS 4
#endif
#ifndef MODE
#define MODE 1
#endif
#ifndef SPREAD
#define SPREAD 1   // 1: the DMA issues of a step go between its MFMAs (one per 8 MFMAs), 0: all at the top of the step
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 256, TQ = 256;
constexpr int OP_BYTES = TM * RB;          // one operand's part of a stage (16 or 32 KiB)
constexpr int STAGE = 2 * OP_BYTES;
constexpr int KSUB = RB / 64;              // 32-wide MFMA k sub-steps per stage (1 or 2)
constexpr int PCS = OP_BYTES / 1024 / 8;   // 1-KiB DMA pieces per wave per operand and stage (2 or 4)
constexpr int DMA_PER_STAGE = 2 * PCS;     // per wave

#define DMA2(voff0, voff1, p0, p1, la)                                                              \
  do { uint32_t keep_m0;                                                                            \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"  \
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\ts_mov_b32 m0, %0" \
               : "=&s"(keep_m0) : "v"(voff0), "v"(voff1), "s"(p0), "s"(p1), "s"(la) : "memory", "scc"); } while (0)

__global__ __launch_bounds__(512, 2) void k2r_kernel(const uint16_t* __restrict__ corpus, const uint16_t* __restrict__ queries,
                                                     int ld, int n_ctiles, int T, int S, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int per = (int)gridDim.x >> 3;
  const int lin = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  const int t = lin / S, s = lin - t * S;
  const int tile_lo = (int)(((long long)s * n_ctiles) / S), tile_hi = (int)(((long long)(s + 1) * n_ctiles) / S);
  const int ksteps = ld * 2 / RB;          // stages per corpus tile
  const uint32_t row_b = (uint32_t)ld * 2u;

  // ---- DMA source offsets (swizzle on the SOURCE, LDS image lane-linear) ----
  // RB = 128: piece = 8 rows x 128 B, lane -> row l>>3, physical chunk l&7 holds logical chunk (l&7) ^ ((row>>1)&7)
  //           pieces c and c+2 share the per-lane part (shipped kernel): even / odd piece offsets
  // RB = 64:  piece = 16 rows x 64 B, lane -> row l>>2, physical chunk l&3 holds logical chunk (l&3) ^ ((-(row>>2))&3);
  //           row>>2 within the piece = l>>4: every piece has the same per-lane part
  uint32_t voff_e, voff_o;
  if (RB == 128) {
    voff_e = (uint32_t)(lane >> 3) * row_b + (uint32_t)(((lane & 7) ^ (lane >> 4)) * 16);
    voff_o = (uint32_t)(lane >> 3) * row_b + (uint32_t)(((lane & 7) ^ (4 + (lane >> 4))) * 16);
  } else {
    voff_e = voff_o = (uint32_t)(lane >> 2) * row_b + (uint32_t)(((lane & 3) ^ ((0 - (lane >> 4)) & 3)) * 16);
  }
  constexpr int ROWS_PER_PIECE = 1024 / RB;
  const size_t wave_rows = (size_t)(PCS * w) * ROWS_PER_PIECE;          // this wave's first row of an operand
  const char* q_ptr = (const char*)(queries + (size_t)t * TQ * ld) + wave_rows * row_b;
  const size_t tile_bytes = (size_t)TM * row_b;
  const uint32_t piece_b = (uint32_t)ROWS_PER_PIECE * row_b;

  // two DMA issues (1 KiB each) of stage slot `slot`: part p of this wave's share (p < PCS: p/2 selects corpus/queries)
  auto dma_pair = [&](const char* a, const char* b, int slot, int p) {   // p = 0 .. PCS-1; pairs (corpus 2i,2i+1) then queries
    const int op = p / (PCS / 2), i = (p % (PCS / 2)) * 2;
    const char* base = (op ? b : a) + (size_t)i * piece_b;
    const uint32_t la = (uint32_t)(slot * STAGE + op * OP_BYTES + (PCS * w + i) * 1024);
    DMA2(voff_e, voff_o, base, base + piece_b, la);
  };

  // ---- fragment read offsets ----
  const int frow = lane & 15;
  int a_rd, b_rd, ph0;
  if (RB == 128) {
    a_rd = (wm * 128 + frow) * 128; b_rd = OP_BYTES + (wn * 64 + frow) * 128;
    ph0 = ((lane >> 4) ^ (frow >> 1)) * 16;                       // k sub-step 1: ph0 ^ 64
  } else {
    a_rd = (wm * 128 + frow) * 64; b_rd = OP_BYTES + (wn * 64 + frow) * 64;
    ph0 = ((lane >> 4) ^ ((0 - (frow >> 2)) & 3)) * 16;
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int mf = 0; mf < 8; ++mf)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int n_tiles = tile_hi - tile_lo;
  const int n_steps = n_tiles * ksteps;
  const char* a_tile = (const char*)corpus + (size_t)tile_lo * tile_bytes + wave_rows * row_b;
  int pf_kk = 0, pf_left = n_steps, pf_slot = 0;
  auto pf_advance = [&]() {
    --pf_left;
    if (++pf_kk == ksteps) { pf_kk = 0; a_tile += tile_bytes; }
    if (++pf_slot == NS) pf_slot = 0;
  };
  // prologue: NS - 1 stages in flight
  for (int i = 0; i < NS - 1 && pf_left > 0; ++i) {
    const char* sa = a_tile + pf_kk * RB; const char* sb = q_ptr + pf_kk * RB;
#pragma unroll
    for (int p = 0; p < PCS; ++p) dma_pair(sa, sb, pf_slot, p);
    pf_advance();
  }
  int slot = 0;
  f16x8 fa[2][8], fb[2][4];   // two fragment register sets (RB = 128: k sub-steps 0 / 1 of a stage; RB = 64: alternate stages)

  // wait until this wave's loads of the stage about to be read have landed: all but the DMA_PER_STAGE * (NS - 2)
  // youngest (the stages behind it in the ring), then the barrier publishes every wave's part
  auto wait_stage = [&](int younger_stages) {
    if (younger_stages >= 3 && NS >= 5) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMA_PER_STAGE * 3) : "memory");
    else if (younger_stages >= 2 && NS >= 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMA_PER_STAGE * 2) : "memory");
    else if (younger_stages >= 1 && NS >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMA_PER_STAGE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  int step = 0;
  // top of a step: wait for its stage, barrier, then the issue of the stage NS - 1 ahead (into the slot read last step)
  const char* sa = nullptr; const char* sb = nullptr; int st_slot = 0; bool do_stage = false;
  auto step_top = [&]() {
    wait_stage(std::min(NS - 2, n_steps - 1 - step));   // stages issued, younger than the one needed now
    do_stage = pf_left > 0;
    sa = a_tile + pf_kk * RB; sb = q_ptr + pf_kk * RB;
    st_slot = pf_slot;
    if (do_stage) pf_advance();
    if ((!SPREAD || MODE == 0) && do_stage) {
#pragma unroll
      for (int p = 0; p < PCS; ++p) dma_pair(sa, sb, st_slot, p);
    }
  };
  auto step_end = [&]() { ++step; if (++slot == NS) slot = 0; };
  auto mfma_set = [&](auto set_tag, auto zero_tag, bool spread) {
    constexpr int SET = decltype(set_tag)::value;
    constexpr bool ZERO = decltype(zero_tag)::value;
#pragma unroll
    for (int part = 0; part < 4; ++part) {
#pragma unroll
      for (int mf = 2 * part; mf < 2 * part + 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[SET][mf], fb[SET][nf], ZERO ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mf][nf], 0, 0, 0);
      if (SPREAD && spread && part < PCS) {
        __builtin_amdgcn_sched_barrier(0);
        if (do_stage) dma_pair(sa, sb, st_slot, part);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

  if (MODE == 0) {
    for (; step < n_steps;) { step_top(); step_end(); }
  } else if (RB == 64) {
    // a stage = one 32-wide k step: its 12 fragment reads go into set (stage & 1) while the 32 MFMAs of the previous
    // stage run from the other set; the tile's last stage is drained before the (absent) epilogue
    auto rstep = [&](auto cur_tag, auto pend_tag, auto zero_tag) {
      constexpr int CUR = decltype(cur_tag)::value;
      constexpr bool PEND = decltype(pend_tag)::value;
      step_top();
      const char* sbuf = sm + slot * STAGE;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) fb[CUR][nf] = *(const f16x8*)(sbuf + b_rd + nf * 1024 + ph0);
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) fa[CUR][mf] = *(const f16x8*)(sbuf + a_rd + mf * 1024 + ph0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PEND) {
        mfma_set(std::integral_constant<int, CUR ^ 1>{}, zero_tag, true);
      } else if (SPREAD && do_stage) {
#pragma unroll
        for (int p = 0; p < PCS; ++p) dma_pair(sa, sb, st_slot, p);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      step_end();
    };
    for (int ti = 0; ti < n_tiles; ++ti) {
      rstep(I0{}, std::false_type{}, std::false_type{});
      rstep(I1{}, std::true_type{}, std::true_type{});
      for (int j = 2; j < ksteps; j += 2) {
        rstep(I0{}, std::true_type{}, std::false_type{});
        rstep(I1{}, std::true_type{}, std::false_type{});
      }
      mfma_set(I1{}, std::false_type{}, false);     // drain
    }
  } else {
    // RB = 128: the shipped schedule -- reads of k sub-step 0, pending MFMAs of sub-step 1 of the previous stage,
    // then the MFMAs of sub-step 0 with the reads of sub-step 1 between them
    auto kstep = [&](auto pend_tag) {
      constexpr bool PEND = decltype(pend_tag)::value;
      step_top();
      const char* sbuf = sm + slot * STAGE;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) fb[0][nf] = *(const f16x8*)(sbuf + b_rd + nf * 2048 + ph0);
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) fa[0][mf] = *(const f16x8*)(sbuf + a_rd + mf * 2048 + ph0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PEND) {
        mfma_set(I1{}, std::false_type{}, true);
      } else if (SPREAD && do_stage) {
#pragma unroll
        for (int p = 0; p < PCS; ++p) dma_pair(sa, sb, st_slot, p);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) fb[1][nf] = *(const f16x8*)(sbuf + b_rd + nf * 2048 + (ph0 ^ 64));
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) fa[1][mf] = *(const f16x8*)(sbuf + a_rd + mf * 2048 + (ph0 ^ 64));
#pragma unroll
      for (int mf = 0; mf < 8; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][mf], fb[0][nf], PEND ? acc[mf][nf] : (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#define SGB(n_rd) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, n_rd, 0);
      SGB(2) SGB(1) SGB(2) SGB(1) SGB(2) SGB(1) SGB(2) SGB(1)
#undef SGB
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      step_end();
    };
    for (int ti = 0; ti < n_tiles; ++ti) {
      kstep(std::false_type{});
      for (int kk = 1; kk < ksteps; ++kk) kstep(std::true_type{});
      mfma_set(I1{}, std::false_type{}, false);     // drain k sub-step 1 of the tile's last stage
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float x = 0.f;
#pragma unroll
  for (int mf = 0; mf < 8; ++mf)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) x += acc[mf][nf][0] + acc[mf][nf][1] + acc[mf][nf][2] + acc[mf][nf][3];
  // (the accumulators of every tile but the split's last are simply overwritten: timing is what this is for)
  out[(size_t)blockIdx.x * 512 + tid] = x;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int nq = argc > 1 ? atoi(argv[1]) : 10000, n = argc > 2 ? atoi(argv[2]) : 1000000, d = argc > 3 ? atoi(argv[3]) : 768;
  const int iters = argc > 4 ? atoi(argv[4]) : 10;
  const int T = (nq + 255) / 256, n_ctiles = (n + 255) / 256;
  const int S = std::max(1, 256 / T) >= 8 ? (256 / T) / 8 * 8 : 256 / T;   // as choose_split: one wave of workgroups
  const int grid = T * S;
  if (grid % 8) { printf("grid %d not a multiple of 8\n", grid); return 1; }
  const size_t cb = (size_t)n_ctiles * 256 * d * 2, qb = (size_t)T * 256 * d * 2;
  uint16_t *dc, *dq; float* dout;
  CK(hipMalloc(&dc, cb)); CK(hipMalloc(&dq, qb)); CK(hipMalloc(&dout, (size_t)grid * 512 * 4));
  // random fp16 in [-1, 1): the clock an MFMA loop holds depends on the data
  std::vector<uint16_t> h(std::max(cb, qb) / 2);
  uint64_t st = 88172645463325252ull;
  auto fill = [&](uint16_t* dst, size_t bytes) {
    for (size_t i = 0; i < bytes / 2; ++i) {
      st ^= st << 13; st ^= st >> 7; st ^= st << 17;
      const float f = (float)((st >> 40) & 0xFFFF) / 32768.f - 1.f;
      _Float16 hf = (_Float16)f; h[i] = *(uint16_t*)&hf;
    }
    CK(hipMemcpy(dst, h.data(), bytes, hipMemcpyHostToDevice));
  };
  fill(dc, cb); fill(dq, qb);
  const int lds = NS * STAGE;
  CK(hipFuncSetAttribute((const void*)k2r_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int it = 0; it < iters + 2; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k2r_kernel, dim3(grid), dim3(512), lds, 0, dc, dq, d, n_ctiles, T, S, dout);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float m; CK(hipEventElapsedTime(&m, e0, e1));
    if (it >= 2) ms.push_back(m);
  }
  CK(hipGetLastError());
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2];
  const double flop = 2.0 * T * 256.0 * n_ctiles * 256.0 * d, bytes = (double)grid * ((double)n_ctiles / S) * (d * 2.0 / RB) * STAGE;
  // spot check (MODE 1): workgroup 0's sum over its last tile against the host
  std::vector<float> ho((size_t)grid * 512);
  CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
  double dev_sum = 0; for (int i = 0; i < 512; ++i) dev_sum += ho[i];
  // host: block 0 = (query tile 0, split 0); the sum over its last corpus tile's 256 x 256 scores = <sum of rows, sum of queries>
  double ref_sum = 0;
  {
    const int last_tile = (int)(((long long)1 * n_ctiles) / S) - 1;
    std::vector<uint16_t> hc((size_t)256 * d), hq((size_t)256 * d);
    CK(hipMemcpy(hc.data(), dc + (size_t)last_tile * 256 * d, hc.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hq.data(), dq, hq.size() * 2, hipMemcpyDeviceToHost));
    std::vector<double> sr(d, 0.0), sq(d, 0.0);
    for (int r = 0; r < 256; ++r) for (int j = 0; j < d; ++j) { sr[j] += (double)*(_Float16*)&hc[(size_t)r * d + j]; sq[j] += (double)*(_Float16*)&hq[(size_t)r * d + j]; }
    for (int j = 0; j < d; ++j) ref_sum += sr[j] * sq[j];
  }
  printf("RB=%d NS=%d MODE=%d SPREAD=%d  T=%d S=%d grid=%d lds=%d: median %.3f ms (min %.3f)  %.1f TFLOP/s  L2->LDS %.2f TB/s (%.1f GB/s per CU)  wg0 sum %.6g (host %.6g)\n",
         RB, NS, MODE, SPREAD, T, S, grid, lds, med, ms[0], flop / med / 1e9, bytes / med / 1e9, bytes / med / 1e6 / grid, dev_sum, ref_sum);
  return 0;
}
