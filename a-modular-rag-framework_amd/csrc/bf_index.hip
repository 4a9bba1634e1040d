// Brute-force cosine / inner-product top-k over an HBM-resident fp16/bf16 corpus.
//
// Replaces, at corpus scale, the reference's "cosine every candidate, sort descending,
// truncate" (app/modules/retrieval/retrieval_backend.py:192-197,245,371-372).
//
//   K1  prep_rows_kernel (common.hip)  normalise + round queries/rows to storage type
//   K2  bf_gemm_topk_kernel            S = C . Q^T on MFMA 16x16x32 with the top-k filter
//                                      fused into the epilogue; the score matrix never
//                                      reaches HBM
//   K4  bf_merge_kernel                per query: select the k best of the candidates the
//                                      K2 workgroups kept, order (score desc, row asc)
//
// K2 geometry (MI355X: 256 CUs, 160 KiB LDS, 8 XCDs x 4 MiB L2)
//   workgroup = 512 threads = 8 waves, tile = 256 corpus rows x 256 queries, K step 64.
//   wave (wm, wn) owns 128 corpus rows x 64 queries = 8 x 4 MFMA 16x16 accumulators.
//   The CORPUS is the MFMA A operand, so an accumulator column (lane & 15) is ONE query:
//   a lane needs only 4 running thresholds in registers to filter its 128 scores.
//   LDS: 2 stages x (32 KiB corpus + 32 KiB query) filled by global_load_lds (16 B/lane),
//   rows of 128 B XOR-swizzled on the 16-B chunk index by (row>>1)&7 so every
//   ds_read_b128 lane group touches 16 distinct slots of the 256-B bank row.  The swizzle
//   is applied to the per-lane SOURCE address (LDS-DMA writes lane-linear).
//   A workgroup = (query tile t, corpus split s): it walks the tiles of its split with the
//   running per-query candidate lists (global memory, L2 resident) and thresholds (LDS).
//   One wave of workgroups: S = floor(CUs / T) splits per query tile (T <= 64 per launch); blocks go
//   to XCDs round-robin, so XCD x gets a contiguous run of the tile-major (t, s) order -- at T = 40
//   five query tiles (1.9 MiB, L2-resident across corpus tiles) x all six corpus streams, each
//   shared by those tiles' workgroups.
//
// Top-k filter (exact)
//   Per query a threshold thr certified by ">= k EARLIER rows score >= thr"; a score is listed
//   only if it beats thr (strict: a later row that ties loses the (score desc, row asc)
//   tie-break).  Certificates: (a) k <= 16: every lane tracks the two largest per-tile maxima it
//   has seen for each of its 4 query columns; the minimum over the 8 lanes that share a query
//   (2 waves x 4 lane groups) of their 2nd-largest is a score that >= 16 earlier rows reach
//   (LDS ds_min, double-buffered per tile); the first tile bootstraps from its own scores and
//   passes on '>='.  (b) any k: the k-th best of a compacted list.
//   (c) k < 16: after tiles 0-7, 15, 31, 63, ... of a split the k-th LARGEST of the 16 certified maxima
//   replaces their minimum (one thread per query, bitonic network in registers): ~2x fewer listings.
//   (d) 16 < k <= 64: no row-count certificate; every query is compacted after tiles 0, 1, 3, 7, ...
//   Lists: each of the 8 lanes that share a query owns a private 64-entry segment of the
//   query's list (count kept in a register during the push, committed to LDS after the tile's
//   barrier): a push is a compare and one predicated 8-byte store, no atomics, no waits.
//   A full segment raises a flag: the tile is replayed in 4 sub-rounds (<= 8 pushes per
//   segment each) with the affected queries compacted in between.  Compaction = one wave per
//   query: keys staged in LDS, MSB-first radix select of the k-th key (skipped when <= 64 keys are
//   staged), rank by counting among the selected, sorted write-back into a 64-entry "kept" area,
//   segments emptied, thr raised to the k-th.  Adversarial inputs (every tile beats the last) take
//   the replay path every tile: slower, never wrong.
//
// Main loop
//   One barrier per K step.  The next stage's 8 LDS-DMA loads per wave are issued (inline asm, scalar
//   base + 32-bit voffset) at the top of a step; fragment reads are software-pipelined over two
//   register sets (the MFMAs of k-half 1 of the previous step run while the reads of this step's
//   half 0 land); the first K step of a tile uses a zero C operand instead of cleared accumulators.
#include "common.h"
#include <type_traits>

#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace mrag {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

constexpr int TM = 256;   // corpus rows per tile
constexpr int TQ = 256;   // queries per tile
constexpr int BK = 64;    // K elements per stage
constexpr int NTHR = 512;
constexpr int A_BYTES = TM * BK * 2;            // 32 KiB
constexpr int STAGE_BYTES = (TM + TQ) * BK * 2; // 64 KiB
constexpr int GEMM_LDS = 2 * STAGE_BYTES;       // 128 KiB

constexpr int NGRP = 8;                   // lanes (2 waves x 4 lane groups) that share a query
constexpr int SEG = 64;                   // private list segment per (query, lane group)
constexpr int KEPT = 64;                  // compacted entries per query (= largest k)
constexpr int QCAP = KEPT + NGRP * SEG;   // 576 list entries per (workgroup, query)
constexpr int KMAX = KEPT;
#ifndef MRAG_DENSE_TILES
#define MRAG_DENSE_TILES 8
#endif
constexpr int DENSE_TILES = MRAG_DENSE_TILES;             // tiles of a split filtered by the mask + LDS path (see the epilogue)
constexpr int K_CERT = 16;                // k served by the 16-row threshold certificate

constexpr int OFF_TAU = GEMM_LDS;               // float[256]      k-th best after a compaction
constexpr int OFF_KCNT = OFF_TAU + 1024;        // int[256]        kept entries
constexpr int OFF_SCNT = OFF_KCNT + 1024;       // int[256][8]     committed segment counts
constexpr int OFF_STAT = OFF_SCNT + 8192;       // uint[2][256]    certificates, orderable bits
constexpr int OFF_FLAGS = OFF_STAT + 2048;      // int[4]
constexpr int OFF_RM = OFF_FLAGS + 16;          // float[4 columns][2][512 threads]: each lane's two largest
                                                // per-tile maxima per query column (kept out of the VGPR budget)
constexpr int LDS_TOTAL = OFF_RM + 4 * 2 * NTHR * 4;

struct BfParams {
  const uint16_t* corpus;   // [n_cap][ld]
  const uint16_t* queries;  // [T*256][ld]
  int ld;                   // padded dim (elements), multiple of 64
  int ksteps;               // ld / 64
  int n_rows;               // valid corpus rows
  int n_ctiles;             // ceil(n_rows / 256)
  int nq;                   // valid queries (rows >= nq of the last query tile are padding)
  int T;                    // query tiles
  int S;                    // corpus splits
  int k;
  int xcd_map;              // 0: plain map; else S % 8 == 0 and this many query tiles share an XCD at a time
  const int* wg_desc;       // descriptor mode (IVF list scan): per workgroup {query row0, valid queries,
                            // first tile, end tile, end row, 0,0,0}; NULL = dense (t, s) decomposition
  int dbg;                  // development ablations (MRAG_DEBUG_FLAGS); 0 in production
  long long* stamps;        // dbg & 16: block 0 / wave 0 writes s_memtime stamps here
  long long* clock;         // 4 words or NULL: workgroup 0's s_memtime / s_memrealtime at entry and exit
  uint2* list;              // [T*S][256][QCAP]  {score bits, local row}: one 8-byte store per push
  int* counts;              // [T*S][256] entries left in the kept area
};

__device__ __forceinline__ uint32_t f32_ord(float f) {
  uint32_t b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord_f32(uint32_t o) {
  uint32_t b = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  return __uint_as_float(b);
}
// largest float strictly below x in comparison order (x finite or -inf; -inf stays -inf)
__device__ __forceinline__ float next_below(float x) {
  if (x == -INFINITY) return x;
  float nb = ord_f32(f32_ord(x) - 1u);
  if (nb == x) nb = ord_f32(f32_ord(nb) - 1u);   // +0.0 -> -0.0 compares equal: step once more
  return nb;
}
// larger key = better: higher score first, then LOWER row
__device__ __forceinline__ uint64_t make_key(uint32_t score_bits, uint32_t row) {
  return ((uint64_t)f32_ord(__uint_as_float(score_bits)) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}

template <int DT> struct Mfma;
template <> struct Mfma<MRAG_F16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma<MRAG_BF16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

extern __shared__ __attribute__((aligned(16))) char smem[];

// Note on waits: the LDS-DMA stage loads are issued from inline asm, so hipcc does not know an
// LDS write is pending on the VM counter and ordinary LDS accesses in the epilogue get lgkmcnt
// waits only (with the __builtin_amdgcn_global_load_lds form every LDS access was guarded by
// s_waitcnt vmcnt(0), draining the prefetch and every outstanding list store per access).
__device__ __forceinline__ uint32_t lds_off(const void* p) {
  return (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ uint32_t lds_load_u32(const void* p) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_off(p)) : "memory");
  return v;
}
__device__ __forceinline__ void lds_min_u32(void* p, uint32_t x) {
  asm volatile("ds_min_u32 %0, %1" : : "v"(lds_off(p)), "v"(x) : "memory");
}
__device__ __forceinline__ void lds_store_u32(void* p, uint32_t x) {
  asm volatile("ds_write_b32 %0, %1" : : "v"(lds_off(p)), "v"(x) : "memory");
}

// Cache policy of the corpus-stream LDS-DMA loads ("" = default, " nt" = non-temporal), measured on one box:
//   batch kernel (K2): the five workgroups of an XCD that share a corpus stream rely on each other's lines staying in
//     L2 for a moment -- nt made it 13 % SLOWER (13.5 -> 15.2 ms at 10 k x 1M): default policy.
//   streaming kernel (K2s): every corpus byte is read ONCE per launch by ONE workgroup -- nt: 0.270 -> 0.257 ms
//     (5.69 -> 5.98 TB/s at 1 x 1M x 768).
#ifndef MRAG_A_POLICY
#define MRAG_A_POLICY ""
#endif
#ifndef MRAG_S_POLICY
#define MRAG_S_POLICY " nt"
#endif
// K walk of the batch kernel.  1 / 2 = corpus tiles with an odd global index / odd index inside the split walk their K
// steps last-to-first, so that the query K slices a workgroup has just used are the ones it asks for next (LRU-friendly:
// the VERDICT r2 #1a experiment).  Measured round 3, same box, interleaved (tools/ab.sh): 10 000 x 1 M x 768
// 13.83 / 13.88 / 13.90 ms forward vs 14.05 / 14.01 / 14.02 (by global parity) and 14.02 / 14.06 / 14.02 (by split
// parity): 1.2 % SLOWER; 1 000 x 1 M 1.593-1.600 vs 1.593-1.613; C2 unchanged.  It also makes a row's fp32 accumulation
// order depend on its tile parity (sharded != unsharded in the last bit).  Not taken: 0.
#ifndef MRAG_ZZ
#define MRAG_ZZ 0
#endif

// Diagnostics (ablations, s_memtime stamps) exist only in a -DMRAG_DIAG=<flags> build (make DIAG=<flags>):
// the flags are COMPILE-TIME constants, so an ablated build carries no extra branches or registers
// (run-time flags cost the K loop its register budget and made every ablation read 15-30 % slow).
// the production kernel carries none of them.
#ifdef MRAG_DIAG
#define MRAG_DBG(bit) ((MRAG_DIAG) & (bit))
#define MRAG_STAMP(i)                                                                     \
  do {                                                                                    \
    if (((MRAG_DIAG) & 16) && stamp_on && blockIdx.x == 0 && tid == 0 && stamp_n < 64) {   \
      p.stamps[stamp_n * 2] = (i);                                                        \
      p.stamps[stamp_n * 2 + 1] = (long long)__builtin_readcyclecounter();                \
      ++stamp_n;                                                                          \
    }                                                                                     \
  } while (0)
#else
#define MRAG_DBG(bit) 0
#define MRAG_STAMP(i) do { } while (0)
#endif

// One wave compacts query q: the k best of its kept area + 8 segments go (sorted, best first)
// into the kept area, the segments are emptied, tau_c rises to the k-th.  The keys are staged
// through `scratch` (>= QCAP u64 of LDS: the stage buffer the K loop has just consumed), then
// ranked by counting with broadcast LDS reads.  Register-light on purpose (it shares the
// kernel's 256-VGPR budget with the 128 accumulators); slow, and rare by construction.

// Select the k best of the `total` unique keys staged in `scratch` (one wave) and write them, sorted
// best first, into the query's kept area; tau_c rises to the k-th.  MSB-first radix select, 8 bits
// per pass over an LDS histogram: each pass narrows the bucket that holds the k-th key; it stops as
// soon as that bucket is needed whole (typically after 3-5 passes).  The <= 64 selected keys are
// then ranked by counting.  ~600 instructions whatever `total` is (rank-by-counting over all the
// entries was ~20 000 at 576 entries, which made k > 16 -- no row-count certificate, thresholds
// driven by compaction alone -- 3-5x slower than k <= 16).
// scratch layout (8 KiB per wave): keys[QCAP] u64 | hist[256] u32 | sel[KMAX] u64
template <bool WIDE = false>   // WIDE: eight candidate reads in flight in the ranking loop (16 more VGPRs: end-of-split tail only)
__device__ __forceinline__ void rank_and_keep(uint2* __restrict__ lst, int q, int k, int lane,
                                              uint64_t* __restrict__ scratch, int total) {
  float* tau_c = (float*)(smem + OFF_TAU);
  int* kcnt = (int*)(smem + OFF_KCNT);
  int* scnt = (int*)(smem + OFF_SCNT);
  uint32_t* hist = (uint32_t*)(scratch + QCAP);
  uint64_t* sel = scratch + QCAP + 128;
  int shift = 0;               // keys with (key >> shift) >= prefix are selected once the loop ends
  uint64_t prefix = 0ull;
  // <= 64 staged keys (the usual case at the end of a split, after pruning): one key per lane, ranked
  // directly below -- no selection pass at all
  if (total > 64) {
    int need = k;
    for (shift = 56; shift >= 0; shift -= 8) {
      *(uint4*)&hist[lane * 4] = make_uint4(0u, 0u, 0u, 0u);
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < total; i += 64) {
        const uint64_t key = scratch[i];
        const bool active = shift == 56 || (key >> (shift + 8)) == prefix;
        if (active) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
      }
      __builtin_amdgcn_wave_barrier();
      const uint4 h = *(const uint4*)&hist[lane * 4];      // bins 4*lane .. 4*lane+3
      const int mine = (int)(h.x + h.y + h.z + h.w);
      int suf = mine;                                       // inclusive suffix sum over lanes >= this one
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_down(suf, off);
        if (lane + off < 64) suf += v;
      }
      int cum = suf - mine;                                 // keys in bins above this lane's
      int found = -1, above = 0, inb = 0;
      const int hb[4] = {(int)h.x, (int)h.y, (int)h.z, (int)h.w};
#pragma unroll
      for (int b = 3; b >= 0; --b) {
        if (found < 0 && cum < need && cum + hb[b] >= need) { found = lane * 4 + b; above = cum; inb = hb[b]; }
        cum += hb[b];
      }
      const unsigned long long who = __ballot(found >= 0);
      const int src = __builtin_ctzll(who);
      const int digit = __shfl(found, src);
      above = __shfl(above, src);
      inb = __shfl(inb, src);
      need -= above;
      prefix = (prefix << 8) | (uint64_t)digit;
      if (inb == need) break;                               // the whole bucket is selected
    }
    if (shift < 0) shift = 0;
  }
  int nsel = total;
  const uint64_t* cand = scratch;
  if (total > 64) {
    // gather the selected keys (exactly k of them)
    nsel = 0;
    for (int base = 0; base < total; base += 64) {
      const int i = base + lane;
      const uint64_t key = i < total ? scratch[i] : 0ull;
      const bool pick = i < total && (key >> shift) >= prefix;
      const unsigned long long bal = __ballot(pick);
      if (pick) sel[nsel + __popcll(bal & ((1ull << lane) - 1ull))] = key;
      nsel += __popcll(bal);
    }
    cand = sel;
  }
  __builtin_amdgcn_wave_barrier();
  const uint64_t my = lane < nsel ? cand[lane] : 0ull;
  int rank = 0;
  // eight broadcast reads in flight per step (one at a time, every step waits out an LDS round trip);
  // slots past nsel hold stale keys: masked by the index test
  if constexpr (WIDE) {
    for (int j0 = 0; j0 < nsel; j0 += 8) {
      uint64_t c8[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) c8[t] = cand[(j0 + t) & 63];
#pragma unroll
      for (int t = 0; t < 8; ++t) rank += (j0 + t < nsel && c8[t] > my) ? 1 : 0;
    }
  } else {
    for (int j = 0; j < nsel; ++j) rank += (cand[j] > my) ? 1 : 0;
  }
  if (lane < nsel && rank < k) {
    const uint32_t sb = __float_as_uint(ord_f32((uint32_t)(my >> 32)));
    lst[rank] = make_uint2(sb, 0xFFFFFFFFu - (uint32_t)my);
    if (rank == k - 1) tau_c[q] = __uint_as_float(sb);
  }
  if (lane == 0) kcnt[q] = min(nsel, k);
  if (lane < NGRP) scnt[q * NGRP + lane] = 0;
}

// End of the split: cut the lists of queries q0..q0+3 to their k best.  A query's entries sit in up to
// nine places (kept area + 8 segments, 576 slots) but number only a few dozen, so they are enumerated
// DENSELY: entry e of query q lives in the place whose running count covers e.  One 8-byte load per lane
// fetches the first 64 entries of a query (usually all of them); the loads of the four queries are issued
// before any is used (the accumulators are dead here); entries below the query's final certified
// threshold are dropped before ranking.  (Walking all nine 64-slot chunks per query made this tail 15 % of
// the kernel at 125 k rows per GPU.)
__device__ __forceinline__ int tail_slot(const int (&pre)[NGRP + 1], int e) {
  // pre[0] = kept entries, pre[g+1] = pre[g] + entries of segment g; e < pre[NGRP]
  int idx = e;                                    // inside the kept area
#pragma unroll
  for (int g = 0; g < NGRP; ++g)
    if (e >= pre[g]) idx = KEPT + g * SEG + (e - pre[g]);
  return idx;
}

// NB queries per call.  Phase A (unrolled, small): one load per query into registers, then parked in
// LDS (`park`, NB x 64 entries).  Phase B (a run-time loop, so the selection code exists once): prune,
// stage keys, rank.  `scratch` = the 8-KiB rank_and_keep work area of this wave.
template <int NB, bool WIDE = false>
__device__ __forceinline__ void tail_compact(uint2* __restrict__ wg_list, int q0, int k, int lane,
                                             uint64_t* __restrict__ scratch, uint2* __restrict__ park, const float* thr_lds,
                                             long long* dbg_t = nullptr) {
  const int* kcnt = (const int*)(smem + OFF_KCNT);
  if (dbg_t) dbg_t[0] = (long long)__builtin_readcyclecounter();
  const int* scnt = (const int*)(smem + OFF_SCNT);
  {
    uint2 first[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int q = q0 + u;
      int pre[NGRP + 1];
      pre[0] = kcnt[q];
#pragma unroll
      for (int g = 0; g < NGRP; ++g) pre[g + 1] = pre[g] + min(scnt[q * NGRP + g], SEG);
      first[u] = make_uint2(0u, 0u);
      if (lane < pre[NGRP]) first[u] = wg_list[(size_t)q * QCAP + tail_slot(pre, lane)];
    }
    if (dbg_t) dbg_t[1] = (long long)__builtin_readcyclecounter();
#pragma unroll
    for (int u = 0; u < NB; ++u) park[u * 64 + lane] = first[u];
    if (dbg_t) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); dbg_t[2] = (long long)__builtin_readcyclecounter(); }
  }
#pragma unroll 1
  for (int u = 0; u < NB; ++u) {
    const int q = q0 + u;
    const uint2* lst = wg_list + (size_t)q * QCAP;
    int pre_u[NGRP + 1];
    pre_u[0] = kcnt[q];
#pragma unroll
    for (int g = 0; g < NGRP; ++g) pre_u[g + 1] = pre_u[g] + min(scnt[q * NGRP + g], SEG);
    const int n_ent = pre_u[NGRP];
    if (n_ent == 0) continue;   // nothing listed (padding query of a partial tile / IVF group): counts are already zero
    const float th = thr_lds[q];
    int total = 0;
    {   // entries 0..63: parked in LDS by phase A.  (Kept apart from the global loads below: merged into one
        // loop hipcc selects between the two POINTERS and issues a flat load, whose wait also covers the
        // previous query's stores -- 2 000 cycles per query.)
      const uint2 en = park[u * 64 + lane];
      const bool valid = lane < n_ent && __uint_as_float(en.x) >= th;
      const unsigned long long bal = __ballot(valid);
      if (valid) scratch[__popcll(bal & ((1ull << lane) - 1ull))] = make_key(en.x, en.y);
      total = __popcll(bal);
    }
    for (int base = 64; base < n_ent; base += 64) {
      const int e = base + lane;
      uint2 en = make_uint2(0u, 0u);
      if (e < n_ent) en = lst[tail_slot(pre_u, e)];
      const bool valid = e < n_ent && __uint_as_float(en.x) >= th;
      const unsigned long long bal = __ballot(valid);
      if (valid) scratch[total + __popcll(bal & ((1ull << lane) - 1ull))] = make_key(en.x, en.y);
      total += __popcll(bal);
    }
    if (dbg_t && u < 4) dbg_t[3 + 2 * u] = (long long)__builtin_readcyclecounter();
    rank_and_keep<WIDE>(wg_list + (size_t)q * QCAP, q, k, lane, scratch, total);
    if (dbg_t && u < 4) dbg_t[4 + 2 * u] = (long long)__builtin_readcyclecounter();
  }
  if (dbg_t) dbg_t[11] = (long long)__builtin_readcyclecounter();
}

__device__ __forceinline__ void compact_query(uint2* __restrict__ lst, int q, int k,
                                              int lane, uint64_t* __restrict__ scratch, float keep_thr) {
  // replay-path compaction of ONE query (register-light: the 128 accumulators are live here)
  int* kcnt = (int*)(smem + OFF_KCNT);
  int* scnt = (int*)(smem + OFF_SCNT);
  const int kc = kcnt[q];
  int total = 0;
  {
#pragma unroll 1
    for (int r = 0; r < QCAP / 64; ++r) {
      const int idx = r * 64 + lane;
      bool valid;
      if (idx < KEPT) valid = idx < kc;
      else valid = ((idx - KEPT) & (SEG - 1)) < min(scnt[q * NGRP + ((idx - KEPT) / SEG)], SEG);
      uint64_t key = 0ull;
      if (valid) {
        const uint2 e = lst[idx];
        valid = __uint_as_float(e.x) >= keep_thr;       // entries below a certified threshold cannot be in the top k
        if (valid) key = make_key(e.x, e.y);            // never 0 for a valid entry (row < 2^31)
      }
      const unsigned long long bal = __ballot(valid);
      if (valid) scratch[total + __popcll(bal & ((1ull << lane) - 1ull))] = key;
      total += __popcll(bal);
    }
  }
  // (one wave: program order + the LDS queue order make the staged keys visible to every lane)
  rank_and_keep(lst, q, k, lane, scratch, total);
}

// Owner wave w (queries 32w .. 32w+31) compacts every query that has a segment above `limit`.
template <int QPW>
__device__ __forceinline__ void compact_owned(uint2* __restrict__ wg_list, int w, int lane, int k, int limit,
                                              uint64_t* __restrict__ scratch) {
  const int* scnt = (const int*)(smem + OFF_SCNT);
  const int q = w * QPW + (lane & (QPW - 1));
  bool need = false;
  if (lane < QPW) {
#pragma unroll
    for (int g = 0; g < NGRP; ++g) need |= scnt[q * NGRP + g] > limit;
  }
  unsigned long long m = __ballot(need);
  while (m) {
    const int b = __builtin_ctzll(m);
    m &= m - 1;
    const int qq = w * QPW + b;
    compact_query(wg_list + (size_t)qq * QCAP, qq, k, lane, scratch, -INFINITY);
  }
}

// NF = 16-query column groups per wave: 4 -> 8 waves (2 x 4) of 128 x 64, two per SIMD, 256 VGPRs each;
//                                        8 -> 4 waves (2 x 2) of 128 x 128, ONE per SIMD, 512 VGPRs each
// DESC: descriptor mode (IVF list scan) -- its block decoding and its partial-query-tile loads exist only in that
// instance, the dense (t, s) instance carries neither.
template <int DT, int NF, bool DESC>
__global__ __launch_bounds__(64 * 2 * (16 / NF), NF == 4 ? 2 : 1) void bf_gemm_topk_kernel(BfParams p) {
  constexpr int WN = 16 / NF;            // waves along the query dimension
  constexpr int NW = 2 * WN;             // waves per workgroup
  constexpr int NT = 64 * NW;            // threads per workgroup
  // Wave roles (MRAG_K2_ROLES=1): only waves 0-3 (one per SIMD) issue the LDS-DMA loads of a stage, 16 each instead of 8 from
  // every wave: the SIMD partner of a loading wave goes straight to its fragment reads and MFMAs, so the two waves of a SIMD
  // stop hitting their DMA issues, their read bursts and the barrier together.
#ifndef MRAG_K2_ROLES
#define MRAG_K2_ROLES 0
#endif
  constexpr bool ROLES = MRAG_K2_ROLES != 0 && NW == 8;
  constexpr int CPW = ROLES ? 8 : 32 / NW;           // 1-KiB DMA chunks per (loading) wave per operand and stage
  constexpr int QPW = TQ / NW;           // queries whose lists a wave owns (compaction)
  typedef typename Mfma<DT>::frag frag;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WN, wn = w % WN;
#ifdef MRAG_DIAG
  int stamp_n = 0;
  bool stamp_on = true;   // narrowed to tiles 60..65 of the split inside the tile loop (the sparse regime)
#endif
  MRAG_STAMP(0);
  // held clock = d(s_memtime) / d(s_memrealtime) x 100 MHz over workgroup 0's lifetime (MI355X_MICROARCH.md, DVFS
  // item 6): two stamps at entry, two at exit, written by one lane to a buffer nothing else reads; p.clock is
  // NULL unless the caller asked for the reading (bench.py's "held_clock_ghz")
  if (!DESC && p.clock && blockIdx.x == 0 && tid == 0) {
    p.clock[0] = (long long)__builtin_amdgcn_s_memtime();
    p.clock[1] = (long long)__builtin_amdgcn_s_memrealtime();
  }

  // ---- block -> (query tile, corpus tile range) ----------------------------------------------
  int wg, q_row0, nq_local, tile_lo, tile_hi, rows_end;
  if (DESC) {
    // descriptor mode: one workgroup = one IVF list x up to 256 of the queries that probe it
    const int* d = p.wg_desc + (size_t)blockIdx.x * 8;
    wg = blockIdx.x;
    q_row0 = d[0]; nq_local = d[1]; tile_lo = d[2]; tile_hi = d[3]; rows_end = d[4];
  } else {
    int t, s;
    const int b = blockIdx.x;
    if (p.xcd_map) {
      // XCD-aware: blocks go to XCDs round-robin (b & 7); give each XCD a CONTIGUOUS run of the tile-major
      // (t, s) order: an XCD then works on grid/8/S query tiles (5 at T = 40: 1.9 MB of queries, L2-resident
      // across corpus tiles) against all S corpus streams, each shared by those query tiles' workgroups
      const int per = (int)gridDim.x >> 3;
      const int lin = (b & 7) * per + (b >> 3);
      t = lin / p.S;
      s = lin - t * p.S;
    } else {
      s = b % p.S;
      t = b / p.S;
    }
    wg = t * p.S + s;
    q_row0 = t * TQ;
    nq_local = p.nq - q_row0;
    tile_lo = (int)(((long long)s * p.n_ctiles) / p.S);
    tile_hi = (int)(((long long)(s + 1) * p.n_ctiles) / p.S);
    rows_end = p.n_rows;
  }
  uint2* wg_list = p.list + (MRAG_DBG(64) ? (size_t)(wg & 255) : (size_t)wg) * TQ * QCAP;   // diag 64: alias list memory (timing only)
  const int ksteps = p.ksteps;

  float* tau_c = (float*)(smem + OFF_TAU);
  int* kcnt = (int*)(smem + OFF_KCNT);
  int* scnt = (int*)(smem + OFF_SCNT);
  uint32_t* stat = (uint32_t*)(smem + OFF_STAT);   // [2][256] orderable bits, ds_min target
  int* flags = (int*)(smem + OFF_FLAGS);
  uint32_t* rm = (uint32_t*)(smem + OFF_RM);
  if (tid < TQ) {
    tau_c[tid] = -INFINITY;
    kcnt[tid] = 0;
    stat[tid] = 0xFFFFFFFFu;
    stat[TQ + tid] = 0xFFFFFFFFu;
  }
  for (int i = tid; i < TQ * NGRP; i += NT) scnt[i] = 0;
  if (tid == 0) flags[0] = 0;

  // ---- LDS-DMA source offsets: wave w fills 1-KiB chunks 4w..4w+3 of each operand ---------
  // chunk c = 8 rows x 128 B; lane l -> row r = 8c + (l>>3), physical 16-B chunk l&7,
  // logical chunk (l&7) ^ ((r>>1)&7) = (l&7) ^ ((4c + (l>>4)) & 7).  Chunks c and c+2 share the
  // per-lane part, so ONE 32-bit voffset per chunk parity is enough; the chunk's row block goes
  // into the scalar base (scalar-base + 32-bit-voffset form, no 64-bit per-lane addresses).
  const uint32_t row_b = (uint32_t)p.ld * 2u;                       // bytes per row
  const uint32_t voff_e = (uint32_t)(lane >> 3) * row_b + (uint32_t)(((lane & 7) ^ (lane >> 4)) * 16);        // even chunks
  const uint32_t voff_o = (uint32_t)(lane >> 3) * row_b + (uint32_t)(((lane & 7) ^ (4 + (lane >> 4))) * 16);  // odd chunks
  const int lw = ROLES ? (w & 3) : w;          // loading wave's slot
  const bool loader = !ROLES || w < 4;
  const char* q_ptr = (const char*)(p.queries + (size_t)q_row0 * p.ld) + (size_t)(CPW * lw) * 8 * row_b;
  const size_t tile_bytes = (size_t)TM * p.ld * 2;
  const uint32_t chunk_b = 8u * row_b;                               // 8 rows

  // One stage = 2*CPW LDS-DMA loads per wave (CPW corpus chunks, CPW query chunks), issued from inline
  // asm two at a time.  hipcc does not count them: every wait for them is an explicit s_waitcnt vmcnt
  // below.  M0 (LDS destination base) is written inside the statement that uses it and restored after.
  f32x4 dbg_sink0 = {0.f, 0.f, 0.f, 0.f}, dbg_sink1 = {0.f, 0.f, 0.f, 0.f};   // diag 16384 only
  constexpr int NPART = CPW;   // two chunks per part, two operands
  auto stage_part = [&](const char* a, const char* b, int buf, int part) {   // first half of the parts: corpus chunks; second half: query chunks
    if (!loader) return;
    const int op = part / (NPART / 2), i = (part % (NPART / 2)) * 2;
    if (op == 1 && (CPW * lw + i) * 8 >= nq_local && DESC) return;
    const char* base = op ? b : a;
    const uint32_t la = (uint32_t)(buf * STAGE_BYTES + op * A_BYTES + (CPW * lw + i) * 1024);   // wave-uniform
    const char* c0 = base + (size_t)i * chunk_b;
    const char* c1 = c0 + chunk_b;
    uint32_t keep;
    if (op == 0) {   // corpus chunks: streamed once per workgroup (cache policy: MRAG_A_POLICY)
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3" MRAG_A_POLICY "\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4" MRAG_A_POLICY "\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(voff_e), "v"(voff_o), "s"(c0), "s"(c1), "s"(la)
          : "memory", "scc");
    } else {         // query chunks: re-read for every corpus tile, default policy (L2-resident)
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(voff_e), "v"(voff_o), "s"(c0), "s"(c1), "s"(la)
          : "memory", "scc");
    }
  };
  auto stage = [&](const char* a, const char* b, int buf) {
    static_assert(CPW == 4 || CPW == 8, "stage_part covers 4 or 8 chunks per operand and wave");
#pragma unroll
    for (int part = 0; part < NPART; ++part) stage_part(a, b, buf, part);
  };

  // ---- fragment read offsets ------------------------------------------------------------
  const int frow = lane & 15;
  const int fsw = frow >> 1;
  const int a_rd = (wm * 128 + frow) * 128;
  const int b_rd = A_BYTES + (wn * (16 * NF) + frow) * 128;
  const int ph0 = (((lane >> 4)) ^ fsw) * 16;      // k sub-step 0: logical chunks 0..3
  // k sub-step 1 reads logical chunks 4..7: ((4 + x) ^ fsw) * 16 == ph0 ^ 64

  f32x4 acc[8][NF];
#pragma unroll
  for (int mf = 0; mf < 8; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prefetch cursor: (tile, k step) of the NEXT stage to issue, advanced with scalar adds only
  const int n_tiles = tile_hi - tile_lo;
  const char* a_tile = (const char*)p.corpus + (size_t)tile_lo * tile_bytes + (size_t)(CPW * lw) * 8 * row_b;   // tile being prefetched (+ this wave's row block)
  int pf_kk = 0, pf_left = n_tiles * ksteps;
  // K walk direction (MRAG_ZZ, see the top of the file): 0 = every corpus tile first-to-last (shipped)
  int pf_tile = (MRAG_ZZ == 2) ? 0 : tile_lo;
  auto pf_koff = [&]() -> int {
    if (DESC || MRAG_ZZ == 0) return pf_kk * (BK * 2);
    return ((pf_tile & 1) ? (ksteps - 1 - pf_kk) : pf_kk) * (BK * 2);
  };
  int buf = 0;
  if (pf_left > 0) {
    const int ko = pf_koff();
    stage(a_tile + ko, q_ptr + ko, 0);
    --pf_left;
    if (++pf_kk == ksteps) { pf_kk = 0; a_tile += tile_bytes; ++pf_tile; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MRAG_STAMP(1);

  // One K step: prefetch the next stage, then this stage's fragment reads + 64 MFMAs per wave.
  // PENDING: the previous K step of this corpus tile left its k-half-1 fragments in f1a/f1b;
  // !PENDING = first K step of a corpus tile: its MFMAs start from a zero C operand (an inline
  // constant), so the accumulators are never cleared by separate instructions.
  frag f1a[8], f1b[NF];
  auto kstep = [&](auto pending_tag) {
    constexpr bool PENDING = decltype(pending_tag)::value;
    // Two of the next stage's eight DMA loads go out at the top of the step; in a PENDING step the other six
    // follow two at a time between the pending MFMAs (corpus chunks first: they can miss L2), so that the wave
    // feeds the MFMA pipe again after 2 DMA issues instead of 8 (all at the top: 13.38 ms; 4 + 4: 13.21; 2 + 6: 13.22
    // vs 13.38 in a second pairing).
    const bool do_stage = pf_left > 0 && !MRAG_DBG(4);   // diag 4: ablate the loads
    const int st_ko = pf_koff();
    const char* st_a = a_tile + st_ko;
    const char* st_b = q_ptr + st_ko;
    const int st_buf = buf ^ 1;
    if (do_stage) {
      --pf_left;
      if (++pf_kk == ksteps) { pf_kk = 0; a_tile += tile_bytes; ++pf_tile; }
#pragma unroll
      for (int pp = 0; pp < NPART / 4; ++pp) stage_part(st_a, st_b, st_buf, pp);
      if (!PENDING || MRAG_DBG(8)) {
#pragma unroll
        for (int pp = NPART / 4; pp < NPART; ++pp) stage_part(st_a, st_b, st_buf, pp);
      }
    }
    const char* sb = smem + buf * STAGE_BYTES;
    buf ^= 1;
    if (MRAG_DBG(8)) return;   // diag 8: ablate LDS reads + MFMA
    {
      // Software-pipelined fragment reads, two register sets.  The 12 reads of k half 0 are issued
      // first; the 32 MFMAs of k half 1 of the PREVIOUS step (operands already in registers) run
      // while they land; then the 32 MFMAs of half 0 run with the 12 reads of half 1 interleaved.
      // The MFMA pipe never waits on a read it has just issued (hipcc's own order re-reads two
      // fragments per 4-8 MFMAs and waits on each: 65 % MFMA duty at 2.15 GHz in the reads + MFMA
      // ablation).  Accumulation order is unchanged: half 0, half 1 of every K step in sequence.
      const int ph1 = ph0 ^ 64;
      frag f0a[8], f0b[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) f0b[nf] = *(const frag*)(sb + b_rd + nf * 2048 + ph0);
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) f0a[mf] = (MRAG_DBG(8192) && mf >= 4) ? f0a[mf - 4] : *(const frag*)(sb + a_rd + mf * 2048 + ph0);   // diag 8192: a third fewer LDS reads (timing only)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PENDING) {
#pragma unroll
        for (int part = 0; part < 4; ++part) {
#pragma unroll
          for (int mf = 2 * part; mf < 2 * part + 2; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = Mfma<DT>::run(f1a[mf], f1b[nf], acc[mf][nf]);
          if (part < 3) {
            __builtin_amdgcn_sched_barrier(0);
            if (do_stage) {
#pragma unroll
              for (int pp = 0; pp < NPART / 4; ++pp) stage_part(st_a, st_b, st_buf, (1 + part) * (NPART / 4) + pp);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) f1b[nf] = *(const frag*)(sb + b_rd + nf * 2048 + ph1);
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) f1a[mf] = (MRAG_DBG(8192) && mf >= 4) ? f1a[mf - 4] : *(const frag*)(sb + a_rd + mf * 2048 + ph1);
#pragma unroll
      for (int mf = 0; mf < 8; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          acc[mf][nf] = Mfma<DT>::run(f0a[mf], f0b[nf], PENDING ? acc[mf][nf] : (f32x4){0.f, 0.f, 0.f, 0.f});
      // pin the interleave: 4 MFMAs, then 2 / 1 LDS reads, eight times (12 reads under 32 MFMAs)
#define MRAG_SGB(n_rd) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, n_rd, 0);
      MRAG_SGB(2) MRAG_SGB(1) MRAG_SGB(2) MRAG_SGB(1) MRAG_SGB(2) MRAG_SGB(1) MRAG_SGB(2) MRAG_SGB(1)
#undef MRAG_SGB
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment read of this stage buffer has returned before the next barrier lets it be restaged
    }
  };
  auto kstep_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stage prefetched during this step has landed
    __syncthreads();
  };

  for (int ti = 0; ti < n_tiles; ++ti) {
#ifdef MRAG_DIAG
    stamp_on = DESC || (ti >= 60 && ti < 66);   // descriptor mode (IVF): one or two tiles per workgroup, stamp them all
    MRAG_STAMP(2);
#endif
    kstep(std::false_type{});
    for (int kk = 1; kk < ksteps; ++kk) {
      kstep_sync();
      kstep(std::true_type{});
    }
    if (!MRAG_DBG(8)) {   // drain: k half 1 of the last K step
#pragma unroll
      for (int mf = 0; mf < 8; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = Mfma<DT>::run(f1a[mf], f1b[nf], acc[mf][nf]);
    }
    MRAG_STAMP(3);
    if (MRAG_DBG(1)) {
      // ablation: no top-k filter; keep the accumulators observable, then clear them
      float x = 0.f;
#pragma unroll
      for (int mf = 0; mf < 8; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          x += acc[mf][nf][0] + acc[mf][nf][1] + acc[mf][nf][2] + acc[mf][nf][3];
        }
      if (x == 123456.789f) kcnt[0] = 1;
    } else {
      // =========================== fused top-k epilogue ===================================
      const int tile = tile_lo + ti;   // ti = tile index inside the split
      // lane-derived values are recomputed from an opaque copy so that nothing the epilogue
      // needs is hoisted out of the K loop (the loop body is near the 256-VGPR budget)
      int elane = lane;
      asm volatile("" : "+v"(elane));
      const int row0 = tile * TM + wm * 128 + (elane >> 4) * 4;  // + mf*16 + j
      const int q0 = wn * (16 * NF) + (elane & 15);              // + nf*16
      const int grp = wm * 4 + (elane >> 4);                     // this lane's segment of each of its queries
      const bool certify = p.k <= K_CERT;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stage prefetched during the last K step has landed (issued a K step ago: no stall)
      MRAG_STAMP(10 + (ti == 0 ? 0 : 100));
      if ((tile + 1) * TM > rows_end) {
#pragma unroll
        for (int mf = 0; mf < 8; ++mf)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (row0 + mf * 16 + j >= rows_end) {
#pragma unroll
              for (int nf = 0; nf < NF; ++nf) acc[mf][nf][j] = -INFINITY;
            }
      }
      float thr[NF];
      uint32_t* stat_cur = stat + (ti & 1) * TQ;        // certificate built from tiles <= ti
      uint32_t* stat_prev = stat + ((ti & 1) ^ 1) * TQ;  // certificate from tiles < ti
      if (ti == 0) {
        // ---- first tile: tau0 = min over the query's 8 lanes of the lane's 2nd largest score
        // IN this tile; >= 16 rows are >= tau0, so pass on '>=' (as '> next_below').
        if (certify) {
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
            for (int mf = 0; mf < 8; ++mf)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float v = acc[mf][nf][j];
                m2 = fmaxf(m2, fminf(m1, v));
                m1 = fmaxf(m1, v);
              }
            rm[(nf * 2 + 0) * NT + tid] = __float_as_uint(m1);
            rm[(nf * 2 + 1) * NT + tid] = __float_as_uint(m2);
            float x = m2;
            x = fminf(x, __shfl_xor(x, 16));
            x = fminf(x, __shfl_xor(x, 32));
            if (elane < 16) lds_min_u32(&stat_cur[q0 + nf * 16], f32_ord(x));
          }
          __syncthreads();
          if constexpr (DESC) if (p.k < K_CERT) {   // descriptor instance only: in the dense instance the extra code costs the K loop registers (+2 % measured)
            // Sharper bootstrap: the k-th LARGEST of the query's 16 certified maxima (8 lanes x top 2, distinct rows
            // of this tile) -- k rows reach it, so it certifies like tau0 (which is their 16th largest) and sits
            // near the true k-th best.  It matters most where a workgroup sees ONE tile (IVF list scan: a 150-row
            // list leaves lanes with fewer than two valid rows, tau0 = -inf, every row listed and every query sent
            // through the radix select at the end: 70 % of that kernel).  Rank by counting, 16 work items per
            // query, register-light (the accumulators are live here).
#pragma unroll 1
            for (int wi = tid; wi < min(nq_local, TQ) * 16; wi += NT) {      // (padding queries list nothing: skipped)
              const int ql = wi >> 4, vi = wi & 15;
              const int cq = ql & 15, cnf = (ql >> 4) & (NF - 1), cwn = ql / (16 * NF);
              const uint32_t* col = rm + (cnf * 2) * NT + cwn * 64 + cq;    // + t*NT + ((o>>2)*WN)*64 + (o&3)*16
              float v16[16];
#pragma unroll
              for (int j = 0; j < 16; ++j) v16[j] = __uint_as_float(col[(j & 1) * NT + ((j >> 3) * WN) * 64 + ((j >> 1) & 3) * 16]);   // 16 reads in flight
              float mine = v16[0];
#pragma unroll
              for (int j = 1; j < 16; ++j) mine = (vi == j) ? v16[j] : mine;
              int rank = 0;
#pragma unroll
              for (int j = 0; j < 16; ++j) rank += (v16[j] > mine || (v16[j] == mine && j < vi)) ? 1 : 0;
              if (rank == p.k - 1 && mine > -INFINITY) {
                stat_cur[ql] = f32_ord(mine);
                tau_c[ql] = mine;
              }
            }
            __syncthreads();
          }
          MRAG_STAMP(11);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) thr[nf] = next_below(ord_f32(stat_cur[q0 + nf * 16]));
        } else {
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) thr[nf] = -INFINITY;
        }
      } else {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const float sc = certify ? ord_f32(stat_prev[q0 + nf * 16]) : -INFINITY;
          thr[nf] = fmaxf(sc, tau_c[q0 + nf * 16]);
        }
      }
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        if (q0 + nf * 16 >= nq_local) thr[nf] = INFINITY;   // padding queries never list anything
      int cseg[NF];   // this lane's segment fill, per query column
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) cseg[nf] = scnt[(q0 + nf * 16) * NGRP + grp];
      if (ti < DENSE_TILES) __builtin_amdgcn_s_barrier();   // every wave is done reading the stage buffer the dense push pass reuses as scratch
      MRAG_STAMP(21);

      // ---- push: attempt 0 = whole tile at once; a full segment -> replay in 4 sub-rounds ------
      int attempt = 0, round = 0;
      float tmax[NF];
      while (true) {
        int rbase = row0;
        asm volatile("" : "+v"(rbase));   // keep the 128 store operands from being pre-formed outside this loop
        if (ti < DENSE_TILES) {
          // (a) first tiles of a split: some lane passes in almost every accumulator
          // Filter, per query column of this lane (32 scores): lane maximum first; when some lane of
          // the wave beats its threshold, every lane builds a 32-bit pass mask (straight-line code,
          // no branches), parks its 32 scores in LDS ([accumulator][thread] x 16 B: the stage buffer
          // the K loop has just consumed) and a short wave-uniform loop stores one passing score per
          // lane per trip, fetched back from LDS by bit index (registers cannot be indexed at run time).
          // ~140 instructions per column whatever the pass rate -- the early tiles of a split pass
          // somewhere in almost every accumulator, which made per-accumulator branching 4x dearer.
          char* stg = smem + (buf ^ 1) * STAGE_BYTES + tid * 16;
  #pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            float tm = acc[0][nf][0];
  #pragma unroll
            for (int mf = 0; mf < 8; ++mf) tm = fmaxf(tm, fmaxf(fmaxf(acc[mf][nf][0], acc[mf][nf][1]), fmaxf(acc[mf][nf][2], acc[mf][nf][3])));
            tmax[nf] = tm;
            if (!__any(tm > thr[nf]) || MRAG_DBG(32 | 512)) continue;   // dbg 32: ablate list pushes (512: of the dense path only)
            uint32_t pm = 0u;
  #pragma unroll
            for (int mf = 7; mf >= 0; --mf)
  #pragma unroll
              for (int j = 3; j >= 0; --j) pm = (pm << 1) | (acc[mf][nf][j] > thr[nf] ? 1u : 0u);   // bit = mf*4 + j
            if (attempt) pm &= 0xFFu << (8 * round);            // replay: 2 accumulators (<= 8 pushes/segment) per round
  #pragma unroll
            for (int mf = 0; mf < 8; ++mf) *(f32x4*)(stg + mf * (NT * 16)) = acc[mf][nf];
            const int q = q0 + nf * 16;
            uint2* lst = wg_list + (size_t)q * QCAP + KEPT + grp * SEG;
            int c = cseg[nf];
            while (__any(pm != 0u)) {
              if (pm) {
                const int b = __builtin_ctz(pm);
                pm &= pm - 1u;
                const float v = *(const float*)(stg + (b >> 2) * (NT * 16) + (b & 3) * 4);
                if (c < SEG) lst[c] = make_uint2(__float_as_uint(v), (uint32_t)(rbase + (b >> 2) * 16 + (b & 3)));
                ++c;
              }
            }
            cseg[nf] = c;
          }
        } else {
          // (b) later tiles: passes are sparse.  Two-level filter: max of each 4-score group (one MFMA
          // accumulator) and a wave-uniform test against the column's threshold; only a group in which
          // some lane beats its threshold runs the store path.
  #pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            const int q = q0 + nf * 16;
            uint2* lst = wg_list + (size_t)q * QCAP + KEPT + grp * SEG;
            float tm = -INFINITY;
            int c = cseg[nf];
            // all eight group tests first (8 v_cmp into 8 SGPR pairs, back to back), then scalar-only
            // branches on the saved masks: a v_cmp -> s_cbranch_vcc pair per group serialises the vector
            // and scalar pipes (~33 cycles per test measured)
            unsigned long long hit[8];
  #pragma unroll
            for (int mf = 0; mf < 8; ++mf) {
              const f32x4 a = acc[mf][nf];
              const float m4 = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
              tm = fmaxf(tm, m4);
              hit[mf] = __ballot(m4 > thr[nf]);
            }
  #pragma unroll
            for (int mf = 0; mf < 8; ++mf) {
              const f32x4 a = acc[mf][nf];
              if (attempt && (mf >> 1) != round) continue;      // replay: 2 accumulators (<= 8 pushes/segment) per round
              if (hit[mf] == 0ull || MRAG_DBG(32 | 1024)) continue;   // dbg 32: ablate list pushes (1024: of the sparse path only)
  #pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (a[j] > thr[nf]) {
                  if (c < SEG) {
                    uint32_t sv = __float_as_uint(a[j]);
                    asm volatile("" : "+v"(sv));   // form the {score,row} pair HERE, not hoisted out of the tile loop (spills)
                    lst[c] = make_uint2(sv, (uint32_t)(rbase + mf * 16 + j));
                  }
                  ++c;
                }
              }
            }
            cseg[nf] = c;
            tmax[nf] = tm;
          }
        }
        {
          bool over = false;
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) over |= cseg[nf] > SEG;
          if (over) lds_store_u32(&flags[0], 1u);
        }
        MRAG_STAMP(22);
        if (attempt == 0 && certify && ti != 0) {
          // fold this tile's maxima into the running certificate used from the NEXT tile on
          // (the four columns as four independent chains: all loads, then all stores, then the shuffles --
          // written column by column the LDS read -> write -> shuffle -> atomic chain ran serially, 3 % of a tile)
          float o1[NF], o2[NF], x[NF];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            o1[nf] = __uint_as_float(rm[(nf * 2 + 0) * NT + tid]);
            o2[nf] = __uint_as_float(rm[(nf * 2 + 1) * NT + tid]);
          }
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            const float n2 = fmaxf(o2[nf], fminf(o1[nf], tmax[nf])), n1 = fmaxf(o1[nf], tmax[nf]);
            rm[(nf * 2 + 0) * NT + tid] = __float_as_uint(n1);
            rm[(nf * 2 + 1) * NT + tid] = __float_as_uint(n2);
            x[nf] = n2;
          }
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) x[nf] = fminf(x[nf], __shfl_xor(x[nf], 16));
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) x[nf] = fminf(x[nf], __shfl_xor(x[nf], 32));
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            if (elane < 16) lds_min_u32(&stat_cur[q0 + nf * 16], f32_ord(x[nf]));
        }
        MRAG_STAMP(12 + (ti == 0 ? 0 : 100));
        // This barrier also publishes the stage prefetched during the last K step (every wave waited
        // for its own LDS-DMA loads at the top of the epilogue), so the common path needs no second
        // barrier per tile.  It is a RAW barrier behind an LDS-only wait: the list stores of this tile
        // stay in flight across it (nobody reads the lists before one of the full __syncthreads() of the
        // replay / compaction / end-of-split paths) -- __syncthreads() here made every wave sit out the
        // completion latency of its last 8-byte store on every tile.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        MRAG_STAMP(13 + (ti == 0 ? 0 : 100));
        if (attempt == 0) {
          if (!flags[0]) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) scnt[(q0 + nf * 16) * NGRP + grp] = cseg[nf];   // commit
            if (tid < TQ) stat_prev[tid] = 0xFFFFFFFFu;   // becomes stat_cur of tile ti+1
            break;
          }
          __syncthreads();               // everyone has seen the flag; committed counts are still pre-tile
          if (tid == 0) flags[0] = 0;
          attempt = 1;
          round = 0;
        } else {
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) scnt[(q0 + nf * 16) * NGRP + grp] = cseg[nf];     // commit (cannot overflow)
          if (round == 3) {
            if (tid < TQ) stat_prev[tid] = 0xFFFFFFFFu;
            break;
          }
          ++round;
          __syncthreads();
        }
        // replay step: make room (every segment <= SEG - 8), then reload counts and thresholds
        // LDS scratch: the stage buffer this K step has just consumed (free until the next stage() call)
        compact_owned<QPW>(wg_list, w, elane, p.k, SEG - 8, (uint64_t*)(smem + (buf ^ 1) * STAGE_BYTES + w * 8192));
        __syncthreads();
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          cseg[nf] = scnt[(q0 + nf * 16) * NGRP + grp];
          float th = tau_c[q0 + nf * 16];
          if (q0 + nf * 16 >= nq_local) th = INFINITY;
          thr[nf] = fmaxf(thr[nf], th);
        }
      }
      // Sharper certificate, refreshed after tiles 0..7, 15, 31, 63, ... of the split: the min over the
      // 8 lanes of their 2nd largest maxima (stat) sits near the 38th best score so far; the k-th
      // LARGEST of the 16 certified maxima (8 lanes x top 2, all distinct earlier rows) is just as valid
      // -- k rows reach it -- and close to the k-th best.  One thread per query sorts the 16 values
      // (bitonic network in registers) and raises tau_c.  Simulated list pushes per (query, split) at
      // k = 10: 231 -> 112 (an exact k-th-best threshold: 88).
      if (certify && p.k < K_CERT && (!DESC || ti != 0) && (ti < 8 || ((ti + 1) & ti) == 0) && ti + 1 < n_tiles) {   // (descriptor mode, tile 0: done in its bootstrap)
        if (tid < TQ) {
          const int cq = tid & 15, cnf = (tid >> 4) & 3, cwn = tid >> 6;
          float v[16];
#pragma unroll
          for (int o = 0; o < 8; ++o) {   // owner lanes of query `tid`: wave row o>>2, lane group o&3
            const int owner = ((o >> 2) * WN + cwn) * 64 + (o & 3) * 16 + cq;
            v[2 * o] = __uint_as_float(rm[(cnf * 2 + 0) * NT + owner]);
            v[2 * o + 1] = __uint_as_float(rm[(cnf * 2 + 1) * NT + owner]);
          }
#pragma unroll
          for (int kk2 = 2; kk2 <= 16; kk2 <<= 1)
#pragma unroll
            for (int j = kk2 >> 1; j > 0; j >>= 1)
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                const int l = i ^ j;
                if (l > i) {
                  const float lo = fminf(v[i], v[l]), hi = fmaxf(v[i], v[l]);
                  const bool desc = (i & kk2) == 0;          // descending overall: v[0] largest
                  v[i] = desc ? hi : lo;
                  v[l] = desc ? lo : hi;
                }
              }
          float sel = v[15];
#pragma unroll
          for (int i = 14; i >= 0; --i) sel = (i == p.k - 1) ? v[i] : sel;
          tau_c[tid] = fmaxf(tau_c[tid], sel);
        }
      }
      // With one K step per tile nothing else separates this tile's flag / certificate words from the
      // next tile's epilogue; the forced compaction below uses the scratch buffer after the barrier.
      bool resync = ksteps == 1;
      if (!certify && ((ti + 1) & ti) == 0 && ti + 1 < n_tiles) {
        resync = true;
        // k > K_CERT has no row-count certificate: thresholds come from compaction alone (tau_c =
        // the k-th best listed so far).  Compact every query after tiles 0, 1, 3, 7, 15, ... of the
        // split: each interval then lists ~k entries per query (the rows seen double, the rate of
        // rows beating a k-th-best threshold halves), so segments practically never fill up.
        __syncthreads();   // all pushes of this tile are in the lists; nobody still parks scores in the stage buffer
        // (the accumulators are dead here: the batched-load form used at the end of the split fits)
        uint64_t* scr = (uint64_t*)(smem + (buf ^ 1) * STAGE_BYTES + w * 8192);
#pragma unroll 1
        // (work area + parking: this wave's 8 KiB of the consumed stage buffer; 4 queries per pass keep the parking at 2 KiB)
        for (int qq = 0; qq < QPW; qq += 4)
          tail_compact<4>(wg_list, w * QPW + qq, p.k, elane, scr, (uint2*)(scr + 768), tau_c);
      }
      MRAG_STAMP(14 + (ti == 0 ? 0 : 100));
      if (resync) __syncthreads();
      // (the accumulators are not cleared: the first K step of the next tile starts from a zero C operand)
      // The K loop's lane constants may have been spilled around the epilogue.  Touch them HERE so
      // that the compiler's reload (and its s_waitcnt vmcnt, which would also drain the LDS-DMA
      // prefetch it cannot see) sits at the end of the epilogue, not inside the next K step.
      asm volatile("" :: "v"(a_rd), "v"(b_rd), "v"(ph0), "v"(voff_e), "v"(voff_o));
    }
    if (MRAG_DBG(1)) kstep_sync();
  }

  if (MRAG_DBG(16384)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); asm volatile("" :: "v"(dbg_sink0), "v"(dbg_sink1)); }
  __syncthreads();
#ifdef MRAG_DIAG
  stamp_on = true;
#endif
  MRAG_STAMP(98);
  // ---- end of the split: every query's list is cut to its k best (sorted) so that K4 only has to
  // merge S x k entries.  Entries below the final certified threshold are dropped before ranking
  // (typically ~25 of ~170 survive), which is what keeps this tail at ~1 % of the workgroup's time.
  if (n_tiles > 0 && !MRAG_DBG(2048)) {   // dbg 2048: ablate the end-of-split compaction
    const uint32_t* stat_last = stat + ((n_tiles - 1) & 1) * TQ;
    uint64_t* scratch = (uint64_t*)(smem + w * 8192);
    // 16 queries per pass: two memory latencies per wave for its 32 lists (they are spread over GBs of
    // list space, i.e. they come from HBM).  Both stage buffers are free here: per wave 8 KiB work area
    // + 8 KiB of parking.  The final thresholds go through the (now idle) tau_c words.
    if (p.k <= K_CERT) {
      for (int i = lane; i < QPW; i += 64) { const int q = w * QPW + i; tau_c[q] = fmaxf(tau_c[q], ord_f32(stat_last[q])); }
    }
    constexpr int TB = 16;
    uint2* park = (uint2*)(smem + NW * 8192 + w * 8192);
#ifdef MRAG_DIAG
    long long tt[12] = {0};
    for (int qq = 0; qq < QPW; qq += TB)
      tail_compact<TB, true>(wg_list, w * QPW + qq, p.k, lane, scratch, park, tau_c, (((MRAG_DIAG) & 16) && qq == 0) ? tt : nullptr);
    if (((MRAG_DIAG) & 16) && blockIdx.x == 0 && tid == 0 && p.stamps)
      for (int i = 0; i < 12; ++i) p.stamps[100 + i] = tt[i] - tt[0];
#else
    for (int qq = 0; qq < QPW; qq += TB) tail_compact<TB, true>(wg_list, w * QPW + qq, p.k, lane, scratch, park, tau_c);
#endif
  }
  __syncthreads();
  MRAG_STAMP(99);
  if (!DESC && p.clock && blockIdx.x == 0 && tid == 0) {
    p.clock[2] = (long long)__builtin_amdgcn_s_memtime();
    p.clock[3] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  if (tid < TQ) p.counts[(size_t)wg * TQ + tid] = kcnt[tid];
}


// ------------------------------------------------------------------------------------------
// K2s: the online regime (<= 16 queries per call, e.g. the reference's one question at a time).
// HBM-bound: every corpus byte is read once and the query block rides along, so the structure is
// a pure stream -- 256-row corpus tiles through a 3-stage LDS ring (2 stages = 68 KiB per CU in
// flight behind counted vmcnt), scored on MFMA against the 16 queries, and filtered per query by
// ONE wave per query pair (no atomics): a wave appends the scores that beat the query's running
// k-th best to a 128-entry LDS list, 64 rows at a time, and compacts the list (rank by counting)
// when it passes 64.  Algorithmic bytes per launch = N*D*2 (+ 16*D*2).
// ------------------------------------------------------------------------------------------
constexpr int SQ = 16;                               // query columns per block (one MFMA B fragment)
constexpr int S_NST = 3;
constexpr int S_STAGE = A_BYTES + SQ * BK * 2;       // 32 KiB corpus + 2 KiB queries
constexpr int S_OFF_SC = S_NST * S_STAGE;            // float [16][260] score tile
constexpr int S_SC_LD = 260;
constexpr int S_OFF_LIST = S_OFF_SC + SQ * S_SC_LD * 4;   // uint2 [queries][LCAP]
// Two instances: QPW = 2 queries per wave (16 per block) with 128-entry lists serves k <= 64;
// QPW = 1 (8 per block) with 512-entry lists serves k <= KMAX_WIDE = 256 -- the reference's dense pool
// is 200 candidates per question (config/settings.yaml:101-102, retrieval_backend.py:218).
constexpr int KMAX_WIDE = 256;
template <int QPW, int LCAP> struct StreamLds {
  static constexpr int NQB = 8 * QPW;                                  // queries per block
  static constexpr int OFF_CNT = S_OFF_LIST + NQB * LCAP * 8;          // int [16]
  static constexpr int OFF_TAU = OFF_CNT + 64;                         // float [16]
  static constexpr int TOTAL = OFF_TAU + 64;
};

struct StreamParams {
  const uint16_t* corpus;
  const uint16_t* queries;   // [>= 16 rows][ld], rows >= nq are zero
  int ld, ksteps, n_rows, n_ctiles, nq, S, k;
  uint2* list;               // K4 layout: [(wg*256 + q)*QCAP + i]
  int* counts;               // [wg*256 + q]
};

// one wave: cut query q's LDS list (<= 64*R entries) to its k best (sorted), raise tau
template <int R>
__device__ __forceinline__ void stream_compact(uint2* __restrict__ lst, int* __restrict__ cnt, float* __restrict__ tau, int q, int k, int lane) {
  const int c = cnt[q];
  uint2 e[R];
  uint64_t key[R];
  int rank[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = r * 64 + lane;
    e[r] = i < c ? lst[i] : make_uint2(0u, 0u);
    key[r] = i < c ? make_key(e[r].x, e[r].y) : 0ull;
    rank[r] = 0;
  }
  for (int j = 0; j < c; ++j) {
    const uint2 o = lst[j];                       // broadcast read
    const uint64_t kj = make_key(o.x, o.y);
#pragma unroll
    for (int r = 0; r < R; ++r) rank[r] += kj > key[r] ? 1 : 0;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = r * 64 + lane;
    if (i < c && rank[r] < k) {
      lst[rank[r]] = e[r];
      if (rank[r] == k - 1) tau[q] = __uint_as_float(e[r].x);
    }
  }
  if (lane == 0) cnt[q] = min(c, k);
}

template <int DT, int QPW, int LCAP>
__global__ __launch_bounds__(NTHR, 2) void bf_stream_topk_kernel(StreamParams p) {
  typedef typename Mfma<DT>::frag frag;
  typedef StreamLds<QPW, LCAP> L;
  static_assert(LCAP % 64 == 0 && LCAP >= 128, "list capacity: whole 64-entry appends + compaction head room");
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  const int tile_lo = (int)(((long long)wg * p.n_ctiles) / p.S);
  const int tile_hi = (int)(((long long)(wg + 1) * p.n_ctiles) / p.S);
  const int ksteps = p.ksteps, n_tiles = tile_hi - tile_lo;
  float* sc = (float*)(smem + S_OFF_SC);
  uint2* lists = (uint2*)(smem + S_OFF_LIST);
  int* cnt = (int*)(smem + L::OFF_CNT);
  float* tau = (float*)(smem + L::OFF_TAU);
  if (tid < SQ) { cnt[tid] = 0; tau[tid] = -INFINITY; }

  const uint32_t row_b = (uint32_t)p.ld * 2u;
  const uint32_t voff_e = (uint32_t)(lane >> 3) * row_b + (uint32_t)(((lane & 7) ^ (lane >> 4)) * 16);
  const uint32_t voff_o = (uint32_t)(lane >> 3) * row_b + (uint32_t)(((lane & 7) ^ (4 + (lane >> 4))) * 16);
  const uint32_t chunk_b = 8u * row_b;
  const size_t tile_bytes = (size_t)TM * p.ld * 2;
  const char* q_ptr = (const char*)p.queries + (size_t)(w & 1) * chunk_b;     // every wave (re)loads query chunk w&1
  const uint32_t voff_q = (w & 1) ? voff_o : voff_e;
  auto stage = [&](const char* a, const char* b, int slot) {
    const uint32_t la = (uint32_t)(slot * S_STAGE + (4 * w) * 1024);
    const uint32_t lb = (uint32_t)(slot * S_STAGE + A_BYTES + (w & 1) * 1024);
    const char *a1 = a + chunk_b, *a2 = a1 + chunk_b, *a3 = a2 + chunk_b;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4" MRAG_S_POLICY "\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5" MRAG_S_POLICY "\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %6" MRAG_S_POLICY "\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7" MRAG_S_POLICY "\n\t"
        "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %8\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff_e), "v"(voff_o), "v"(voff_q), "s"(a), "s"(a1), "s"(a2), "s"(a3), "s"(b), "s"(la), "s"(lb)
        : "memory", "scc");
  };
  const int frow = lane & 15, fsw = frow >> 1;
  const int a_rd = (w * 32 + frow) * 128;
  const int b_rd = A_BYTES + frow * 128;
  const int ph0 = ((lane >> 4) ^ fsw) * 16;
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

  const char* a_tile = (const char*)p.corpus + (size_t)tile_lo * tile_bytes + (size_t)(4 * w) * 8 * row_b;
  int pf_kk = 0, pf_left = n_tiles * ksteps, pf_slot = 0, slot = 0;
#pragma unroll 1
  for (int i = 0; i < S_NST - 1; ++i)
    if (pf_left > 0) {
      stage(a_tile + pf_kk * (BK * 2), q_ptr + pf_kk * (BK * 2), pf_slot);
      --pf_left;
      pf_slot = pf_slot == S_NST - 1 ? 0 : pf_slot + 1;
      if (++pf_kk == ksteps) { pf_kk = 0; a_tile += tile_bytes; }
    }
  __syncthreads();
  const int total = n_tiles * ksteps;
  int step = 0;
  for (int ti = 0; ti < n_tiles; ++ti) {
    for (int kk = 0; kk < ksteps; ++kk, ++step) {
      if (step + 1 < total) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // the next stage may stay in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (pf_left > 0) {
        stage(a_tile + pf_kk * (BK * 2), q_ptr + pf_kk * (BK * 2), pf_slot);
        --pf_left;
        pf_slot = pf_slot == S_NST - 1 ? 0 : pf_slot + 1;
        if (++pf_kk == ksteps) { pf_kk = 0; a_tile += tile_bytes; }
      }
      const char* sb = smem + slot * S_STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ph = ks ? (ph0 ^ 64) : ph0;
        const frag b = *(const frag*)(sb + b_rd + ph);
        const frag a0 = *(const frag*)(sb + a_rd + ph);
        const frag a1 = *(const frag*)(sb + a_rd + 2048 + ph);
        acc[0] = Mfma<DT>::run(a0, b, acc[0]);
        acc[1] = Mfma<DT>::run(a1, b, acc[1]);
      }
      slot = slot == S_NST - 1 ? 0 : slot + 1;
    }
    // ---- tile done: scores -> LDS [query][row], then one wave per query pair filters them -------
    const int tile = tile_lo + ti;
    const int q = lane & 15;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = w * 32 + mf * 16 + (lane >> 4) * 4 + j;
        sc[q * S_SC_LD + r] = (tile * TM + r < p.n_rows) ? acc[mf][j] : -INFINITY;
      }
    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll 1
    for (int qq = 0; qq < QPW; ++qq) {
      const int qi = QPW * w + qq;
      if (qi >= p.nq) continue;
      uint2* lst = lists + qi * LCAP;
#pragma unroll 1
      for (int c0 = 0; c0 < TM; c0 += 64) {
        const float v = sc[qi * S_SC_LD + c0 + lane];
        const bool pass = v > tau[qi];                 // strict: a later tie loses to the rows already listed
        const unsigned long long bal = __ballot(pass);
        if (bal) {
          const int base = cnt[qi];
          if (pass) lst[base + __popcll(bal & ((1ull << lane) - 1ull))] = make_uint2(__float_as_uint(v), (uint32_t)(tile * TM + c0 + lane));
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) cnt[qi] = base + __popcll(bal);
          __builtin_amdgcn_wave_barrier();
          if (base + __popcll(bal) > LCAP - 64) stream_compact<LCAP / 64>(lst, cnt, tau, qi, p.k, lane);
        }
      }
    }
    __syncthreads();   // score tile free again
  }
  // ---- end of the split: k best per query (sorted) into the K4 layout ------------------------
  for (int qq = 0; qq < QPW; ++qq) {
    const int qi = QPW * w + qq;
    if (qi >= p.nq) { if (lane == 0 && qi < SQ) p.counts[(size_t)wg * TQ + qi] = 0; continue; }
    uint2* lst = lists + qi * LCAP;
    stream_compact<LCAP / 64>(lst, cnt, tau, qi, p.k, lane);
    __builtin_amdgcn_wave_barrier();
    const int c = cnt[qi];
    uint2* out = p.list + ((size_t)wg * TQ + qi) * QCAP;
    for (int i = lane; i < c; i += 64) out[i] = lst[i];
    if (lane == 0) p.counts[(size_t)wg * TQ + qi] = c;
  }
}

// ------------------------------------------------------------------------------------------
// K4: one wave per query selects the k best of the S workgroups' lists (kept + 8 segments each).
// ------------------------------------------------------------------------------------------
// LDS the merge is launched with: one batch of up to 64 regions (k entries each) on top of the carried best --
// 1 KiB at S = 6, k = 10 instead of a fixed 33 KiB, i.e. 32 instead of 4 resident waves per CU (K4 over 10 000
// queries: 164 -> ~60 us)
static inline int merge_lds_entries(int n_regions, int k) { return std::min(n_regions, 64) * k + KMAX; }

struct MergeParams {
  const uint2* list;
  const int* counts;
  int T, S, k;
  int64_t nq;
  int64_t id_base;
  const int2* pair_loc;     // descriptor mode: [nq][nprobe] (workgroup, slot) of each probe, wg < 0 = none
  int nprobe;
  const int64_t* row_ids;   // descriptor mode: stored position -> original row (ties break on it)
  int cap;                  // LDS key slots of this launch (merge_lds_entries)
  float* out_scores;   // [nq][k]
  int64_t* out_ids;    // [nq][k]
};

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off), lo = __shfl_xor((uint32_t)v, off);
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    v = o > v ? o : v;
  }
  return v;
}

// k best keys of keys[0..fill) -> best[0..nsel), best first (keys are unique: the row is in the key)
__device__ __forceinline__ int merge_extract(const uint64_t* keys, int fill, int k, uint64_t* best, int lane) {
  uint64_t last = ~0ull;
  const int nsel = min(k, fill);
  for (int i = 0; i < nsel; ++i) {
    uint64_t m = 0ull;
    for (int e = lane; e < fill; e += 64) {
      const uint64_t key = keys[e];
      if (key < last && key > m) m = key;
    }
    m = wave_max_u64(m);
    if (lane == 0) best[i] = m;
    last = m;
  }
  return nsel;
}

// One wave per query.  Every K2 workgroup left the query's k best (sorted) in its kept area; 64
// of them are gathered at a time, one per lane, into LDS at offsets from a wave prefix sum, then
// the k best are extracted by k wave-wide max passes.
__global__ __launch_bounds__(64) void bf_merge_kernel(MergeParams p) {
  uint64_t* keys = (uint64_t*)smem;          // [cap] dynamic (merge_lds_entries)
  __shared__ uint64_t best[KMAX];
  const int lane = threadIdx.x;
  const int64_t q = blockIdx.x;
  if (q >= p.nq) return;
  const int t = (int)(q / TQ), ql = (int)(q % TQ);
  const int k = p.k;
  const int n_regions = p.pair_loc ? p.nprobe : p.S;   // one region (the kept area, <= k entries) per K2 workgroup
  int fill = 0;
  for (int r0 = 0; r0 < n_regions; r0 += 64) {
    const int r = r0 + lane;
    int c = 0;
    size_t base = 0;
    if (r < n_regions) {
      size_t wq;
      bool ok = true;
      if (p.pair_loc) {
        const int2 loc = p.pair_loc[q * p.nprobe + r];
        ok = loc.x >= 0;
        wq = (size_t)(ok ? loc.x : 0) * TQ + (size_t)loc.y;
      } else {
        wq = (size_t)(t * p.S + r) * TQ + ql;
      }
      if (ok) c = p.counts[wq];
      base = wq * QCAP;
    }
    int incl = c;   // inclusive wave prefix sum
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    const int total = __shfl(incl, 63);
    if (fill + total > p.cap) {   // make room: fold what is staged into the running best
      __syncthreads();
      const int nb = merge_extract(keys, fill, k, best, lane);
      __syncthreads();
      for (int i = lane; i < nb; i += 64) keys[i] = best[i];
      fill = nb;
    }
    const int off0 = fill + incl - c;
    // eight entries of the lane's region in flight at a time (written one load per trip, every trip sat out a full
    // memory round trip: the lists were written by other CUs, so they come from beyond L2 -- 41 us for ONE query's
    // 256 regions in the online regime)
    int cmax = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
    for (int i0 = 0; i0 < cmax; i0 += 8) {
      uint2 e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) e[u] = (i0 + u < c) ? p.list[base + i0 + u] : make_uint2(0u, 0u);
      if (p.row_ids) {
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + u < c) e[u].y = (uint32_t)p.row_ids[e[u].y];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) if (i0 + u < c) keys[off0 + i0 + u] = make_key(e[u].x, e[u].y);
    }
    fill += total;
  }
  __syncthreads();
  const int nbest = merge_extract(keys, fill, k, best, lane);
  __syncthreads();
  for (int i = lane; i < k; i += 64) {
    float sc = -INFINITY;
    int64_t id = -1;
    if (i < nbest) {
      const uint64_t key = best[i];
      sc = ord_f32((uint32_t)(key >> 32));
      id = p.id_base + (int64_t)(0xFFFFFFFFu - (uint32_t)key);
    }
    p.out_scores[q * k + i] = sc;
    p.out_ids[q * k + i] = id;
  }
}

// K4s, the merge of the online regime (a handful of queries x up to 256 splits, k <= 256): every split left its k best
// SORTED, and the global top k draws on average k/S entries from each, so only the heads of the lists are read.  One
// 256-thread workgroup per query, thread = region: round 1 loads 8 entries of every region into LDS and radix-selects
// the k-th largest of what is loaded (T); a region whose last loaded entry still reaches T may hold more of the top k
// and loads 8 more; repeat until no region does.  Then every unloaded entry is below its region's last loaded one,
// hence below T: the k largest loaded keys ARE the global top k.  Typically one round, 2 048 entries instead of S * k
// (the one-wave K4 walked 256 regions in four dependent batches: 41 us of a 0.30 ms search; the global-memory radix
// select K4w took 0.27 ms at k = 200).  A query whose loads would not fit the LDS budget is flagged and left to K4w.
constexpr int MS_CAP = 12288;          // keys staged in LDS (96 KiB)

// k-th largest of keys[0..n) (n > k, keys unique) as a (shift, prefix) pair: selected <=> (key >> shift) >= prefix.
// All 256 threads; hist[256] and pick[3] are LDS scratch.
__device__ __forceinline__ void block_radix_kth(const uint64_t* keys, int n, int k, uint32_t* hist, int* pick, int tid,
                                                int& shift_out, uint64_t& prefix_out) {
  const int lane = tid & 63;
  int need = k, shift = 56;
  uint64_t prefix = 0ull;
  for (; shift >= 0; shift -= 8) {
    hist[tid] = 0u;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      const uint64_t key = keys[i];
      if (shift == 56 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      const int hb[4] = {(int)hist[lane * 4], (int)hist[lane * 4 + 1], (int)hist[lane * 4 + 2], (int)hist[lane * 4 + 3]};
      const int mine = hb[0] + hb[1] + hb[2] + hb[3];
      int suf = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_down(suf, off);
        if (lane + off < 64) suf += v;
      }
      int cum = suf - mine;
#pragma unroll
      for (int b = 3; b >= 0; --b) {
        if (cum < need && cum + hb[b] >= need) { pick[0] = lane * 4 + b; pick[1] = cum; pick[2] = hb[b]; }
        cum += hb[b];
      }
    }
    __syncthreads();
    const int digit = pick[0], above = pick[1], inb = pick[2];
    need -= above;
    prefix = (prefix << 8) | (uint64_t)digit;
    if (inb == need) break;
  }
  if (shift < 0) shift = 0;
  shift_out = shift;
  prefix_out = prefix;
}

__global__ __launch_bounds__(256) void bf_merge_sorted_kernel(MergeParams p, int* __restrict__ fallback, int cap) {
  uint64_t* keys = (uint64_t*)smem;        // [MS_CAP] dynamic
  __shared__ uint32_t hist[256];
  __shared__ uint64_t sel[KMAX_WIDE];
  __shared__ int pick[3];
  __shared__ int s_total, s_more, s_nsel;
  const int tid = threadIdx.x;
  const int64_t q = blockIdx.x;
  if (q >= p.nq) return;
  const int t = (int)(q / TQ), ql = (int)(q % TQ);
  const int k = p.k, S = p.S;
  int c = 0;
  size_t base = 0;
  if (tid < S) {
    const size_t wq = (size_t)(t * S + tid) * TQ + ql;
    c = min(p.counts[wq], k);
    base = wq * QCAP;
  }
  if (tid == 0) { s_total = 0; s_nsel = 0; }
  int loaded = 0, shift = 0;
  uint64_t last = ~0ull, prefix = 0ull;     // T = (shift, prefix); (0, 0) selects everything
  bool overflow = false;
  for (int round = 0; round < 64; ++round) {
    if (tid == 0) s_more = 0;
    __syncthreads();
    const bool want = loaded < c && (last >> shift) >= prefix;
    const int n_new = want ? min(8, c - loaded) : 0;
    int slot = 0;
    if (n_new) { slot = atomicAdd(&s_total, n_new); atomicOr(&s_more, 1); }
    __syncthreads();
    if (!s_more) break;
    if (s_total > cap) { overflow = true; break; }        // (uniform: s_total is read after the barrier)
    if (n_new) {
      uint2 e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) e[u] = u < n_new ? p.list[base + loaded + u] : make_uint2(0u, 0u);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < n_new) { last = make_key(e[u].x, e[u].y); keys[slot + u] = last; }    // sorted: the last one loaded is the region's smallest so far
      loaded += n_new;
    }
    __syncthreads();
    const int n = s_total;
    if (n > k) block_radix_kth(keys, n, k, hist, pick, tid, shift, prefix);
    else { shift = 0; prefix = 0ull; }
  }
  if (tid == 0) fallback[q] = overflow ? 1 : 0;
  if (overflow) return;                      // K4w redoes this query from global memory
  const int n = s_total;
  for (int i = tid; i < n; i += 256) {
    const uint64_t key = keys[i];
    if ((key >> shift) >= prefix) sel[atomicAdd(&s_nsel, 1)] = key;
  }
  __syncthreads();
  const int nsel = s_nsel;                   // = min(n, k)
  if (tid < nsel) {
    const uint64_t my = sel[tid];
    int rank = 0;
    for (int j = 0; j < nsel; ++j) rank += sel[j] > my ? 1 : 0;
    p.out_scores[q * k + rank] = ord_f32((uint32_t)(my >> 32));
    p.out_ids[q * k + rank] = p.id_base + (int64_t)(0xFFFFFFFFu - (uint32_t)my);
  }
  for (int i = nsel + tid; i < k; i += 256) { p.out_scores[q * k + i] = -INFINITY; p.out_ids[q * k + i] = -1; }
}

// K4w (fallback of K4s; 64 < k <= 256, the online regime with the reference's 200-candidate pool): one 256-thread workgroup
// per query selects the k best of the S regions' sorted lists straight from global memory (S*k <= 65 536
// keys, L2-resident): MSB-first radix select of the k-th key (8 bits per pass over an LDS histogram, stops
// as soon as the bucket holding the k-th key is needed whole), the k selected keys gathered into LDS and
// ranked by counting.  Same order as K4: (score desc, row asc); slots past the corpus are (-inf, -1).
__global__ __launch_bounds__(256) void bf_merge_wide_kernel(MergeParams p, const int* __restrict__ only_flagged) {
  if (only_flagged && !only_flagged[blockIdx.x]) return;   // K4s handled this query
  __shared__ uint32_t hist[256];
  __shared__ uint64_t sel[KMAX_WIDE];
  __shared__ int cnts[256];
  __shared__ int s_pick[3];     // digit, keys above it, keys in its bin
  __shared__ int s_nsel;
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t q = blockIdx.x;
  if (q >= p.nq) return;
  const int t = (int)(q / TQ), ql = (int)(q % TQ);
  const int k = p.k, S = p.S;
  cnts[tid] = tid < S ? min(p.counts[(size_t)(t * S + tid) * TQ + ql], k) : 0;
  if (tid == 0) s_nsel = 0;
  __syncthreads();
  int total = 0;
  for (int r = 0; r < S; ++r) total += cnts[r];
  const int span = S * k;       // entry idx -> (region idx / k, slot idx % k)
  auto fetch = [&](int idx, uint64_t& key) -> bool {
    const int r = idx / k, i = idx - r * k;
    if (i >= cnts[r]) return false;
    const uint2 e = p.list[((size_t)(t * S + r) * TQ + ql) * QCAP + i];
    key = make_key(e.x, e.y);
    return true;
  };
  int shift = 0;
  uint64_t prefix = 0ull;
  if (total > k) {
    int need = k;
    for (shift = 56; shift >= 0; shift -= 8) {
      hist[tid] = 0u;
      __syncthreads();
      for (int idx = tid; idx < span; idx += 256) {
        uint64_t key;
        if (fetch(idx, key) && (shift == 56 || (key >> (shift + 8)) == prefix))
          atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid < 64) {           // wave 0: bins 4*lane .. 4*lane+3, suffix sums over the lanes above
        const int hb[4] = {(int)hist[lane * 4], (int)hist[lane * 4 + 1], (int)hist[lane * 4 + 2], (int)hist[lane * 4 + 3]};
        const int mine = hb[0] + hb[1] + hb[2] + hb[3];
        int suf = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int v = __shfl_down(suf, off);
          if (lane + off < 64) suf += v;
        }
        int cum = suf - mine;
#pragma unroll
        for (int b = 3; b >= 0; --b) {
          if (cum < need && cum + hb[b] >= need) { s_pick[0] = lane * 4 + b; s_pick[1] = cum; s_pick[2] = hb[b]; }
          cum += hb[b];
        }
      }
      __syncthreads();
      const int digit = s_pick[0], above = s_pick[1], inb = s_pick[2];
      need -= above;
      prefix = (prefix << 8) | (uint64_t)digit;
      if (inb == need) break;   // the whole bucket is selected
    }
    if (shift < 0) shift = 0;
  }
  for (int idx = tid; idx < span; idx += 256) {
    uint64_t key;
    if (fetch(idx, key) && (key >> shift) >= prefix) sel[atomicAdd(&s_nsel, 1)] = key;
  }
  __syncthreads();
  const int nsel = s_nsel;      // = min(total, k): keys are unique (the row is in the key)
  if (tid < nsel) {
    const uint64_t my = sel[tid];
    int rank = 0;
    for (int j = 0; j < nsel; ++j) rank += sel[j] > my ? 1 : 0;
    p.out_scores[q * k + rank] = ord_f32((uint32_t)(my >> 32));
    p.out_ids[q * k + rank] = p.id_base + (int64_t)(0xFFFFFFFFu - (uint32_t)my);
  }
  for (int i = nsel + tid; i < k; i += 256) { p.out_scores[q * k + i] = -INFINITY; p.out_ids[q * k + i] = -1; }
}

__global__ void fill_empty_kernel(float* sc, int64_t* ids, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { sc[i] = -INFINITY; ids[i] = -1; }
}

__global__ void unpack_rows_kernel(const uint16_t* __restrict__ src, int ld, int dim, int is_bf16, int64_t n,
                                   float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * dim) return;
  const int64_t r = i / dim;
  const int c = (int)(i % dim);
  const uint16_t b = src[r * ld + c];
  float f;
  if (is_bf16) f = __uint_as_float((uint32_t)b << 16);
  else { _Float16 h; __builtin_memcpy(&h, &b, 2); f = (float)h; }
  out[i] = f;
}

// one wave per listed row: score = <prepared query, stored row> (fp64 accumulation of the exact 16-bit products)
__global__ __launch_bounds__(256) void score_rows_kernel(const uint16_t* __restrict__ rows, const uint16_t* __restrict__ q, int ld,
                                                         int is_bf16, const int64_t* __restrict__ ids, int64_t n, int64_t n_rows,
                                                         float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int64_t r = ids[i];
  if (r < 0 || r >= n_rows) { if (lane == 0) out[i] = 0.0f; return; }   // unknown row: the reference's 0.0 for an unusable vector
  const uint16_t* row = rows + (size_t)r * ld;
  double acc = 0.0;
  for (int c = lane; c < ld; c += 64) {
    float a, b;
    if (is_bf16) { a = __uint_as_float((uint32_t)row[c] << 16); b = __uint_as_float((uint32_t)q[c] << 16); }
    else { _Float16 ha, hb; __builtin_memcpy(&ha, &row[c], 2); __builtin_memcpy(&hb, &q[c], 2); a = (float)ha; b = (float)hb; }
    acc += (double)a * (double)b;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) out[i] = (float)acc;
}

// ------------------------------------------------------------------------------------------
// Wide batches (64 < k <= 256, more than a handful of queries -- batch evaluation with the reference's 200-candidate pool,
// config/settings.yaml:101-102): the batch kernel keeps at most KMAX = 64 entries per (query, corpus split), so it runs
// with k' = 64 over S >= 4k / 64 splits and the merge takes the k best of the S x 64 candidates.  That answer is exact
// unless some split held MORE than 64 of the query's true top k; this kernel checks it: a split whose list is full and
// whose worst kept entry still beats the merged k-th entry may have dropped rows that belong in the answer.  Flagged
// queries (a handful per 10 000 on i.i.d. rows; many when a query's neighbours sit in adjacent rows) are redone exactly
// by the streaming kernel.  One thread per query.
__global__ void bf_wide_verify_kernel(const uint2* __restrict__ list, const int* __restrict__ counts, int S, int k_kept, int k,
                                      int64_t nq, const float* __restrict__ out_scores, const int64_t* __restrict__ out_ids,
                                      int64_t id_base, int* __restrict__ need) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int t = (int)(q / TQ), ql = (int)(q % TQ);
  const int64_t tid_ = out_ids[q * k + k - 1];
  const uint64_t key_tau = tid_ < 0 ? 0ull : make_key(__float_as_uint(out_scores[q * k + k - 1]), (uint32_t)(tid_ - id_base));
  int bad = 0;
  for (int s = 0; s < S; ++s) {
    const size_t wq = (size_t)(t * S + s) * TQ + ql;
    if (counts[wq] >= k_kept) {
      const uint2 e = list[wq * QCAP + (k_kept - 1)];
      if (make_key(e.x, e.y) > key_tau) bad = 1;
    }
  }
  need[q] = bad;
}

struct BfIndex : Object {
  int dim = 0, ld = 0, metric = 0, dtype = MRAG_F16;
  int64_t n = 0, cap_rows = 0, id_base = 0;
  uint16_t* rows = nullptr;
  DevBuf qbuf, lists, counts, stage_in, out_sc, out_id, clock;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool timed = false, want_clock = false;
  void* q_cleared_buf = nullptr;            // query buffer bookkeeping: rows [q_dirty_rows, q_cleared_rows) are known to be zero
  int64_t q_cleared_rows = 0, q_dirty_rows = 0;
  ~BfIndex() override {
    if (rows) (void)hipFree(rows);
    qbuf.release(); lists.release(); counts.release(); stage_in.release(); out_sc.release(); out_id.release(); clock.release();
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
  }
};

#ifdef MRAG_DIAG
static long long* g_stamps = nullptr;  // diagnostic s_memtime stamps (MRAG_DEBUG_FLAGS & 16)
#endif

#ifdef MRAG_DIAG
static int dump_stamps(int dbg, hipStream_t stream) {
  if (!g_stamps || !(dbg & 16)) return MRAG_OK;
  long long hs[128];
  MRAG_HIP(hipStreamSynchronize(stream));
  MRAG_HIP(hipMemcpy(hs, g_stamps, sizeof(hs), hipMemcpyDeviceToHost));
  fprintf(stderr, "[mrag tail] first pass of 16 queries, cycles from its start: loads issued %lld, parked %lld; query 0 staged %lld ranked %lld; q1 %lld %lld; q2 %lld %lld; q3 %lld %lld; pass done %lld\n",
          hs[101], hs[102], hs[103], hs[104], hs[105], hs[106], hs[107], hs[108], hs[109], hs[110], hs[111]);
  fprintf(stderr, "[mrag stamps]");
  for (int i = 0; i < 64 && (i == 0 || hs[2 * i + 1]); ++i) fprintf(stderr, " %lld:%lld", hs[2 * i], hs[2 * i + 1] - hs[1]);
  fprintf(stderr, "\n");
  return MRAG_OK;
}
#endif

static size_t dtype_size(int dt) {
  switch (dt) { case MRAG_F32: return 4; case MRAG_F16: case MRAG_BF16: return 2; case MRAG_F64: return 8; default: return 0; }
}

static int grow_rows(BfIndex* ix, int64_t need_rows, hipStream_t stream) {
  if (need_rows <= ix->cap_rows) return MRAG_OK;
  int64_t cap = std::max<int64_t>(need_rows, ix->cap_rows + ix->cap_rows / 2);
  cap = round_up(std::max<int64_t>(cap, TM), TM);
  uint16_t* nr = nullptr;
  const size_t bytes = (size_t)cap * ix->ld * 2;
  hipError_t e = hipMalloc((void**)&nr, bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(MRAG_ERR_OOM, "hipMalloc(%zu) for corpus rows failed: %s", bytes, hipGetErrorString(e)); }
  MRAG_HIP(hipMemsetAsync(nr, 0, bytes, stream));
  if (ix->rows && ix->n > 0) MRAG_HIP(hipMemcpyAsync(nr, ix->rows, (size_t)ix->n * ix->ld * 2, hipMemcpyDeviceToDevice, stream));
  if (ix->rows) { MRAG_HIP(hipStreamSynchronize(stream)); (void)hipFree(ix->rows); }
  ix->rows = nr;
  ix->cap_rows = cap;
  return MRAG_OK;
}

// choose the corpus split count for T query tiles
static void choose_split(int T, int n_ctiles, int* S_out, int* xcd_out) {
  // ONE wave of workgroups: S = floor(CUs / T) corpus splits per query tile (T <= 64 per launch).  Every
  // workgroup pays a first-tile bootstrap, loose thresholds in its early tiles and an end-of-split
  // compaction (~4.4 tiles' worth, measured), so few long splits beat the 2.5-5 waves of shorter ones a
  // CU-multiple grid needs -- even with T*S a little short of the CU count: 10 k queries (T = 40, S = 6,
  // 240 of 256 CUs) against S = 32: 2.43 -> 2.02 ms at 125 k rows, 4.22 -> 3.79 at 250 k, 7.60 -> 7.27 at
  // 500 k, 14.41 -> 14.32 at 1M.
  static int n_cu = 0;
  if (!n_cu) {
    hipDeviceProp_t pr; int dev = 0;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
               ? pr.multiProcessorCount : 256;
  }
  int S = std::max(1, std::min(n_ctiles, n_cu / std::max(1, T)));
  { const char* fs = getenv("MRAG_S");   // experiment: force the split count
    if (fs && atoi(fs) > 0) S = std::min(atoi(fs), n_ctiles); }
  *S_out = S;
  // the XCD-aware block map needs a grid that splits evenly over the 8 XCDs, and pays when an XCD then holds
  // >= 2 whole query tiles (their workgroups share the corpus streams); with fewer it measured 3-6 % slower
  *xcd_out = ((T * S) % 8 == 0 && (T * S) / 8 >= 2 * S && n_ctiles >= 64) ? 1 : 0;
}


static int64_t g_wide_redone = 0;   // queries the last wide-batch search redid through the streaming kernel (tests / tools)
int64_t bf_last_wide_redone() { return g_wide_redone; }
int bf_max_k() { return KMAX; }            // fused batch kernel / IVF list scan
int bf_max_k_wide() { return KMAX_WIDE; }  // streaming kernel, 8 queries per launch
int64_t bf_round_rows(int64_t n) { return round_up(n, TM); }

// Query batches are cut so that one launch keeps its candidate lists (4.6 KB per workgroup and
// query) within a few GB: at most 64 query tiles (16 384 queries) per launch.
constexpr int MAX_QTILES_PER_LAUNCH = 64;

// K2 launcher.  The kernel is templated on the wave tiling; only the 8-wave form (NF = 4) is built:
// the 4-wave / one-wave-per-SIMD form (NF = 8, 256 accumulator registers per lane) is correct but
// hipcc's allocation of the 256-AGPR tile spills into the K loop (98 ms vs 13.6 ms, round-1 measurement).
static int launch_k2(int dtype, size_t grid, hipStream_t stream, const BfParams& p) {
  typedef void (*K2Fn)(BfParams);
  static const K2Fn fns[2][2] = {{bf_gemm_topk_kernel<MRAG_F16, 4, false>, bf_gemm_topk_kernel<MRAG_F16, 4, true>},
                                 {bf_gemm_topk_kernel<MRAG_BF16, 4, false>, bf_gemm_topk_kernel<MRAG_BF16, 4, true>}};
  static bool attr_done[2][2] = {};
  const int di = dtype == MRAG_F16 ? 0 : 1, de = p.wg_desc ? 1 : 0;
  if (!attr_done[di][de]) {
    MRAG_HIP(hipFuncSetAttribute((const void*)fns[di][de], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    attr_done[di][de] = true;
  }
  if (!grid) return MRAG_OK;
  hipLaunchKernelGGL(fns[di][de], dim3((unsigned)grid), dim3(NTHR), LDS_TOTAL, stream, p);
  MRAG_HIP(hipGetLastError());
  return MRAG_OK;
}

int bf_launch(const BfLaunch& a) {
  if (a.k <= 0 || a.k > (a.wg_desc ? KMAX : KMAX_WIDE)) return fail(MRAG_ERR_UNSUPPORTED, "k = %d outside 1..%d", a.k, a.wg_desc ? KMAX : KMAX_WIDE);
  hipStream_t stream = a.stream;
  const int di = a.dtype == MRAG_F16 ? 0 : 1;
  BfParams p;
  p.corpus = a.corpus;
  p.ld = a.ld;
  p.ksteps = a.ld / BK;
  p.k = a.k;
  p.dbg = 0;
  p.stamps = nullptr;
  p.clock = a.clock;
#ifdef MRAG_DIAG
  {
    p.dbg = MRAG_DIAG;
    if (p.dbg & 16) {
      if (!g_stamps) MRAG_HIP(hipMalloc((void**)&g_stamps, 128 * 8));
      MRAG_HIP(hipMemsetAsync(g_stamps, 0, 128 * 8, stream));
      p.stamps = g_stamps;
    }
  }
#endif
  if (a.ev_k2_begin) MRAG_HIP(hipEventRecord(a.ev_k2_begin, stream));
  if (a.wg_desc) {
    // ---- descriptor mode: one launch, one merge -------------------------------------------
    const size_t grid = (size_t)a.n_wg;
    MRAG_TRY(a.lists->ensure(grid * TQ * QCAP * 8));
    MRAG_TRY(a.counts->ensure(grid * TQ * sizeof(int)));
    p.queries = a.queries;
    p.n_rows = 0; p.n_ctiles = 0; p.nq = 0; p.T = (int)grid; p.S = 1; p.xcd_map = 0;
    p.wg_desc = a.wg_desc;
    p.list = (uint2*)a.lists->p;
    p.counts = (int*)a.counts->p;
    MRAG_TRY(launch_k2(a.dtype, grid, stream, p));
    if (a.ev_k2_end) MRAG_HIP(hipEventRecord(a.ev_k2_end, stream));
#ifdef MRAG_DIAG
    MRAG_TRY(dump_stamps(p.dbg, stream));
#endif
    MergeParams mp;
    mp.list = p.list; mp.counts = p.counts;
    mp.T = 0; mp.S = 0; mp.k = a.k; mp.nq = a.nq; mp.id_base = a.id_base;
    mp.pair_loc = (const int2*)a.pair_loc; mp.nprobe = a.nprobe; mp.row_ids = a.row_ids;
    mp.out_scores = a.out_scores; mp.out_ids = a.out_ids;
    if (a.nq) {
      mp.cap = merge_lds_entries(a.nprobe, a.k);
      hipLaunchKernelGGL(bf_merge_kernel, dim3((unsigned)a.nq), dim3(64), (size_t)mp.cap * 8, stream, mp);
      MRAG_HIP(hipGetLastError());
    }
    return MRAG_OK;
  }
  const int n_ctiles = (int)((a.n_rows + TM - 1) / TM);
  // ---- online regime: HBM-bound streaming kernel (K2s), one workgroup per CU-sized corpus split ----
  // k <= 64: up to 16 queries per launch; 64 < k <= 256: 8 per launch with 512-entry lists, any number of queries as
  // consecutive launches (exact, one HBM pass each).  `extra_counts`: ints the caller keeps behind the per-region counts.
  auto run_stream = [&](const uint16_t* queries, int64_t nq_s, float* out_scores, int64_t* out_ids, bool record_end,
                        size_t extra_counts) -> int {
    typedef void (*SFn)(StreamParams);
    static const SFn sfns[2][2] = {{bf_stream_topk_kernel<MRAG_F16, 2, 128>, bf_stream_topk_kernel<MRAG_F16, 1, 512>},
                                   {bf_stream_topk_kernel<MRAG_BF16, 2, 128>, bf_stream_topk_kernel<MRAG_BF16, 1, 512>}};
    static const int slds[2] = {StreamLds<2, 128>::TOTAL, StreamLds<1, 512>::TOTAL};
    static bool s_attr[2][2] = {};
    // the 512-entry-list instance also wins for 32 < k <= 64: with 128-entry lists a list of k = 64 is compacted after every
    // 64-row chunk (1 x 1M x 768: 0.431 ms at k = 64 against 0.332 ms at k = 100 on the wide instance)
    const int wi = (a.k > 32) ? 1 : 0;
    const int grp = wi ? 8 : SQ;
    if (!s_attr[di][wi]) {
      MRAG_HIP(hipFuncSetAttribute((const void*)sfns[di][wi], hipFuncAttributeMaxDynamicSharedMemorySize, slds[wi]));
      s_attr[di][wi] = true;
    }
    const int S = std::max(1, std::min(n_ctiles, 256));
    if (!extra_counts) {    // (a caller with extra_counts has sized both workspaces itself: ensure() does not keep contents)
      MRAG_TRY(a.lists->ensure((size_t)S * TQ * QCAP * 8));
      MRAG_TRY(a.counts->ensure(((size_t)S * TQ + 16) * sizeof(int)));   // + the K4s fallback flags of one query group
    }
    for (int64_t g0 = 0; g0 < nq_s; g0 += grp) {
      const int64_t gq = std::min<int64_t>(grp, nq_s - g0);
      StreamParams sp;
      sp.corpus = a.corpus; sp.queries = queries + (size_t)g0 * a.ld; sp.ld = a.ld; sp.ksteps = a.ld / BK;
      sp.n_rows = (int)a.n_rows; sp.n_ctiles = n_ctiles; sp.nq = (int)gq; sp.S = S; sp.k = a.k;
      sp.list = (uint2*)a.lists->p; sp.counts = (int*)a.counts->p;
      hipLaunchKernelGGL(sfns[di][wi], dim3((unsigned)S), dim3(NTHR), slds[wi], stream, sp);
      MRAG_HIP(hipGetLastError());
      if (record_end && g0 + grp >= nq_s && a.ev_k2_end) MRAG_HIP(hipEventRecord(a.ev_k2_end, stream));
      MergeParams mp;
      mp.list = sp.list; mp.counts = sp.counts;
      mp.T = 1; mp.S = S; mp.k = a.k; mp.nq = gq; mp.id_base = a.id_base;
      mp.pair_loc = nullptr; mp.nprobe = 0; mp.row_ids = nullptr;
      mp.out_scores = out_scores + (size_t)g0 * a.k; mp.out_ids = out_ids + (size_t)g0 * a.k;
      if (wi || S > 64) {
        // K4s (heads of the sorted lists through LDS) + K4w for the queries it flags (adversarial layouts only)
        static bool s_attr2 = false;
        if (!s_attr2) {
          MRAG_HIP(hipFuncSetAttribute((const void*)bf_merge_sorted_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MS_CAP * 8));
          s_attr2 = true;
        }
        int* flags = (int*)a.counts->p + (size_t)S * TQ;       // 16 ints behind the per-region counts
        static const int ms_cap = [] { const char* e = getenv("MRAG_K4S_CAP"); return e ? std::max(0, std::min(atoi(e), MS_CAP)) : MS_CAP; }();   // test knob: 0 sends every query to K4w
        hipLaunchKernelGGL(bf_merge_sorted_kernel, dim3((unsigned)gq), dim3(256), (size_t)MS_CAP * 8, stream, mp, flags, ms_cap);
        hipLaunchKernelGGL(bf_merge_wide_kernel, dim3((unsigned)gq), dim3(256), 0, stream, mp, (const int*)flags);
      } else {
        mp.cap = merge_lds_entries(S, a.k);
        hipLaunchKernelGGL(bf_merge_kernel, dim3((unsigned)gq), dim3(64), (size_t)mp.cap * 8, stream, mp);
      }
      MRAG_HIP(hipGetLastError());
    }
    return MRAG_OK;
  };
  static const bool wide_off = [] { const char* e = getenv("MRAG_NO_WIDE_BATCH"); return e && atoi(e) != 0; }();   // test knob: every k > 64 search through the streaming kernel
  const bool wide_batch = a.k > KMAX && a.nq > 32 && n_ctiles >= 64 && !wide_off;
  if (!a.wg_desc) g_wide_redone = 0;
  if (!wide_batch && (a.k > KMAX || (a.nq <= (a.k > 32 ? 8 : SQ) && n_ctiles >= 8))) return run_stream(a.queries, a.nq, a.out_scores, a.out_ids, true, 0);
  if (wide_batch) {
    // ---- wide batch: the batch kernel at k' = 64 over S >= 4k/64 splits + merge to k + verification (bf_wide_verify_kernel),
    // the flagged queries redone by the streaming kernel.  10 000 queries at the reference's pool of 200 were 1 250 streaming
    // launches (one HBM pass each, ~0.5 s over 1M rows); now ceil(T / 16..19) batch-kernel launches.
    int n_cu = 256;
    { int S0, x0; choose_split(1, 1 << 30, &S0, &x0); n_cu = S0; }                       // (choose_split(1, inf) = CU count)
    const int S_min = std::max(4, (4 * a.k + KMAX - 1) / KMAX);
    const int T_max = std::max(1, std::min(MAX_QTILES_PER_LAUNCH, n_cu / S_min));
    const int T_all = (int)((a.nq + TQ - 1) / TQ);
    const int S_stream = std::max(1, std::min(n_ctiles, 256));
    // one sizing of both workspaces for every launch below (ensure() drops contents)
    const size_t grid_max = (size_t)std::max(n_cu, S_stream);          // T * S <= CUs for every launch below; the streaming redo uses S_stream regions
    MRAG_TRY(a.lists->ensure(grid_max * TQ * QCAP * 8));
    const size_t cnt_ints = grid_max * TQ + 16;
    const size_t flag_off = cnt_ints, need_off = flag_off + (size_t)T_max * TQ;                    // ints
    const size_t stage_off_b = round_up((int64_t)((need_off + (size_t)a.nq) * sizeof(int)), 256);     // bytes: 8 staged query rows, then results
    const size_t res_off_b = stage_off_b + (size_t)16 * a.ld * 2;
    MRAG_TRY(a.counts->ensure(res_off_b + (size_t)8 * a.k * 12 + 256));
    int* need = (int*)a.counts->p + need_off;
    static bool s_attr3 = false;
    if (!s_attr3) {
      MRAG_HIP(hipFuncSetAttribute((const void*)bf_merge_sorted_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MS_CAP * 8));
      s_attr3 = true;
    }
    for (int t0 = 0; t0 < T_all; t0 += T_max) {
      const int T = std::min(T_max, T_all - t0);
      const int64_t q0 = (int64_t)t0 * TQ;
      const int64_t nq = std::min<int64_t>(a.nq - q0, (int64_t)T * TQ);
      int S, xcd;
      choose_split(T, n_ctiles, &S, &xcd);
      const size_t grid = (size_t)T * S;
      p.queries = a.queries + (size_t)q0 * a.ld;
      p.n_rows = (int)a.n_rows; p.n_ctiles = n_ctiles; p.nq = (int)nq;
      p.T = T; p.S = S; p.xcd_map = xcd; p.k = KMAX;
      p.wg_desc = nullptr;
      p.list = (uint2*)a.lists->p;
      p.counts = (int*)a.counts->p;
      MRAG_TRY(launch_k2(a.dtype, grid, stream, p));
      if (t0 + T_max >= T_all && a.ev_k2_end) MRAG_HIP(hipEventRecord(a.ev_k2_end, stream));
      MergeParams mp;
      mp.list = p.list; mp.counts = p.counts;
      mp.T = T; mp.S = S; mp.k = a.k; mp.nq = nq; mp.id_base = a.id_base;
      mp.pair_loc = nullptr; mp.nprobe = 0; mp.row_ids = nullptr;
      mp.out_scores = a.out_scores + (size_t)q0 * a.k;
      mp.out_ids = a.out_ids + (size_t)q0 * a.k;
      int* flags = (int*)a.counts->p + flag_off;
      hipLaunchKernelGGL(bf_merge_sorted_kernel, dim3((unsigned)nq), dim3(256), (size_t)MS_CAP * 8, stream, mp, flags, MS_CAP);
      hipLaunchKernelGGL(bf_merge_wide_kernel, dim3((unsigned)nq), dim3(256), 0, stream, mp, (const int*)flags);
      hipLaunchKernelGGL(bf_wide_verify_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, (const uint2*)p.list, (const int*)p.counts,
                         S, KMAX, a.k, nq, (const float*)mp.out_scores, (const int64_t*)mp.out_ids, a.id_base, need + q0);
      MRAG_HIP(hipGetLastError());
    }
    // the flagged queries, exactly, eight per streaming launch (their rows staged contiguously)
    std::vector<int> h_need((size_t)a.nq);
    MRAG_HIP(hipMemcpyAsync(h_need.data(), need, (size_t)a.nq * sizeof(int), hipMemcpyDeviceToHost, stream));
    MRAG_HIP(hipStreamSynchronize(stream));
    std::vector<int64_t> redo;
    for (int64_t q = 0; q < a.nq; ++q) if (h_need[(size_t)q]) redo.push_back(q);
    g_wide_redone = (int64_t)redo.size();
    uint16_t* stage = (uint16_t*)((char*)a.counts->p + stage_off_b);
    float* r_sc = (float*)((char*)a.counts->p + res_off_b);
    int64_t* r_id = (int64_t*)((char*)a.counts->p + res_off_b + (size_t)8 * a.k * 4);
    for (size_t g0 = 0; g0 < redo.size(); g0 += 8) {
      const size_t m = std::min<size_t>(8, redo.size() - g0);
      MRAG_HIP(hipMemsetAsync(stage, 0, (size_t)16 * a.ld * 2, stream));
      for (size_t i = 0; i < m; ++i)
        MRAG_HIP(hipMemcpyAsync(stage + i * a.ld, a.queries + (size_t)redo[g0 + i] * a.ld, (size_t)a.ld * 2, hipMemcpyDeviceToDevice, stream));
      MRAG_TRY(run_stream(stage, (int64_t)m, r_sc, r_id, false, 1));
      for (size_t i = 0; i < m; ++i) {
        MRAG_HIP(hipMemcpyAsync(a.out_scores + (size_t)redo[g0 + i] * a.k, r_sc + i * a.k, (size_t)a.k * 4, hipMemcpyDeviceToDevice, stream));
        MRAG_HIP(hipMemcpyAsync(a.out_ids + (size_t)redo[g0 + i] * a.k, r_id + i * a.k, (size_t)a.k * 8, hipMemcpyDeviceToDevice, stream));
      }
    }
    return MRAG_OK;
  }
  // ---- dense mode: (query tile, corpus split) grid, query batches of <= 64 tiles ----------------
  const int T_all = (int)((a.nq + TQ - 1) / TQ);
  for (int t0 = 0; t0 < T_all; t0 += MAX_QTILES_PER_LAUNCH) {
    const int T = std::min(MAX_QTILES_PER_LAUNCH, T_all - t0);
    const int64_t q0 = (int64_t)t0 * TQ;
    const int64_t nq = std::min<int64_t>(a.nq - q0, (int64_t)T * TQ);
    int S, xcd;
    choose_split(T, n_ctiles, &S, &xcd);
    const size_t grid = (size_t)T * S;
    MRAG_TRY(a.lists->ensure(grid * TQ * QCAP * 8));
    MRAG_TRY(a.counts->ensure(grid * TQ * sizeof(int)));
    p.queries = a.queries + (size_t)q0 * a.ld;
    p.n_rows = (int)a.n_rows;
    p.n_ctiles = n_ctiles;
    p.nq = (int)nq;
    p.T = T; p.S = S; p.xcd_map = xcd;
    p.wg_desc = nullptr;
    p.list = (uint2*)a.lists->p;
    p.counts = (int*)a.counts->p;
    MRAG_TRY(launch_k2(a.dtype, grid, stream, p));
    if (t0 + MAX_QTILES_PER_LAUNCH >= T_all && a.ev_k2_end) MRAG_HIP(hipEventRecord(a.ev_k2_end, stream));
    MergeParams mp;
    mp.list = p.list; mp.counts = p.counts;
    mp.T = T; mp.S = S; mp.k = a.k; mp.nq = nq; mp.id_base = a.id_base;
    mp.pair_loc = nullptr; mp.nprobe = 0; mp.row_ids = nullptr;
    mp.out_scores = a.out_scores + (size_t)q0 * a.k;
    mp.out_ids = a.out_ids + (size_t)q0 * a.k;
    mp.cap = merge_lds_entries(S, a.k);
    hipLaunchKernelGGL(bf_merge_kernel, dim3((unsigned)nq), dim3(64), (size_t)mp.cap * 8, stream, mp);
    MRAG_HIP(hipGetLastError());
#ifdef MRAG_DIAG
    MRAG_TRY(dump_stamps(p.dbg, stream));
#endif
  }
  return MRAG_OK;
}

}  // namespace mrag

using namespace mrag;

extern "C" {

int mrag_index_create(int dim, int metric, int storage_dtype, int device, mrag_handle* out) {
  if (!out) return fail(MRAG_ERR_INVALID, "out is NULL");
  if (dim <= 0 || dim > 8192) return fail(MRAG_ERR_INVALID, "dim %d out of range (1..8192)", dim);
  if (metric != MRAG_METRIC_COSINE && metric != MRAG_METRIC_IP) return fail(MRAG_ERR_INVALID, "unknown metric %d", metric);
  if (storage_dtype != MRAG_F16 && storage_dtype != MRAG_BF16) return fail(MRAG_ERR_INVALID, "storage dtype must be fp16 or bf16");
  MRAG_TRY(use_device(device));
  BfIndex* ix = new BfIndex();
  ix->kind = KIND_BF_INDEX;
  ix->device = device;
  ix->dim = dim;
  ix->ld = (int)round_up(dim, BK);
  ix->metric = metric;
  ix->dtype = storage_dtype;
  for (auto& e : ix->ev) {
    if (hipEventCreate(&e) != hipSuccess) { delete ix; return fail(MRAG_ERR_HIP, "hipEventCreate failed"); }
  }
  *out = register_object(ix);
  return MRAG_OK;
}

int mrag_index_destroy(mrag_handle h) {
  Object* o = take(h, KIND_BF_INDEX);
  if (!o) return MRAG_ERR_INVALID;
  (void)hipSetDevice(o->device);
  (void)hipDeviceSynchronize();
  delete o;
  return MRAG_OK;
}

int mrag_index_reserve(mrag_handle h, int64_t n_rows) {
  BfIndex* ix = (BfIndex*)lookup(h, KIND_BF_INDEX);
  if (!ix) return MRAG_ERR_INVALID;
  if (n_rows < 0 || n_rows > 0x7FFFFF00ll) return fail(MRAG_ERR_INVALID, "n_rows out of range");
  MRAG_TRY(use_device(ix->device));
  MRAG_TRY(grow_rows(ix, n_rows, nullptr));
  // the zero fill / copy above ran on the null stream; callers add and search on their own (non-blocking)
  // streams, which are not ordered behind it: finish it here (reserve is a one-off)
  MRAG_HIP(hipStreamSynchronize(nullptr));
  return MRAG_OK;
}

int mrag_index_add(mrag_handle h, const void* rows, int64_t n, int src_dtype, int normalize, int rows_is_device,
                   void* stream_) {
  BfIndex* ix = (BfIndex*)lookup(h, KIND_BF_INDEX);
  if (!ix) return MRAG_ERR_INVALID;
  if (n < 0) return fail(MRAG_ERR_INVALID, "n < 0");
  if (n == 0) return MRAG_OK;
  if (!rows) return fail(MRAG_ERR_INVALID, "rows is NULL");
  const size_t esz = dtype_size(src_dtype);
  if (!esz) return fail(MRAG_ERR_INVALID, "unknown source dtype %d", src_dtype);
  if (ix->n + n > 0x7FFFFF00ll) return fail(MRAG_ERR_UNSUPPORTED, "more than 2^31 rows per index; shard the corpus");
  MRAG_TRY(use_device(ix->device));
  hipStream_t stream = (hipStream_t)stream_;
  MRAG_TRY(grow_rows(ix, ix->n + n, stream));
  const void* src = rows;
  if (!rows_is_device) {
    const size_t bytes = (size_t)n * ix->dim * esz;
    MRAG_TRY(ix->stage_in.ensure(bytes));
    MRAG_HIP(hipMemcpyAsync(ix->stage_in.p, rows, bytes, hipMemcpyHostToDevice, stream));
    src = ix->stage_in.p;
  }
  MRAG_TRY(launch_prep_rows(src, src_dtype, n, ix->dim, ix->rows + (size_t)ix->n * ix->ld, ix->ld, ix->dtype,
                            normalize, stream));
  if (!rows_is_device) MRAG_HIP(hipStreamSynchronize(stream));
  ix->n += n;
  return MRAG_OK;
}

int mrag_index_size(mrag_handle h, int64_t* out_rows) {
  BfIndex* ix = (BfIndex*)lookup(h, KIND_BF_INDEX);
  if (!ix) return MRAG_ERR_INVALID;
  if (!out_rows) return fail(MRAG_ERR_INVALID, "out_rows is NULL");
  *out_rows = ix->n;
  return MRAG_OK;
}

int mrag_index_dim(mrag_handle h, int* out_dim) {
  BfIndex* ix = (BfIndex*)lookup(h, KIND_BF_INDEX);
  if (!ix) return MRAG_ERR_INVALID;
  if (!out_dim) return fail(MRAG_ERR_INVALID, "out_dim is NULL");
  *out_dim = ix->dim;
  return MRAG_OK;
}

int mrag_index_set_id_base(mrag_handle h, int64_t id_base) {
  BfIndex* ix = (BfIndex*)lookup(h, KIND_BF_INDEX);
  if (!ix) return MRAG_ERR_INVALID;
  ix->id_base = id_base;
  return MRAG_OK;
}

int mrag_index_max_k(int64_t nq, int* out_k) {
  if (!out_k) return fail(MRAG_ERR_INVALID, "out_k is NULL");
  (void)nq;   // every batch size is served up to the wide limit (k > 64: streaming kernel, or the wide-batch path of bf_launch)
  *out_k = KMAX_WIDE;
  return MRAG_OK;
}

int mrag_index_last_wide_redone(int64_t* out_queries) {
  if (!out_queries) return fail(MRAG_ERR_INVALID, "out_queries is NULL");
  *out_queries = bf_last_wide_redone();
  return MRAG_OK;
}

int mrag_index_get_rows(mrag_handle h, int64_t row0, int64_t n, float* out, int out_is_device, void* stream_) {
  BfIndex* ix = (BfIndex*)lookup(h, KIND_BF_INDEX);
  if (!ix) return MRAG_ERR_INVALID;
  if (row0 < 0 || n < 0 || row0 + n > ix->n) return fail(MRAG_ERR_INVALID, "row range [%lld,%lld) outside [0,%lld)", (long long)row0, (long long)(row0 + n), (long long)ix->n);
  if (n == 0) return MRAG_OK;
  if (!out) return fail(MRAG_ERR_INVALID, "out is NULL");
  MRAG_TRY(use_device(ix->device));
  hipStream_t stream = (hipStream_t)stream_;
  float* dst = out;
  const size_t bytes = (size_t)n * ix->dim * 4;
  if (!out_is_device) { MRAG_TRY(ix->stage_in.ensure(bytes)); dst = (float*)ix->stage_in.p; }
  const int64_t tot = n * ix->dim;
  hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream,
                     ix->rows + (size_t)row0 * ix->ld, ix->ld, ix->dim, ix->dtype == MRAG_BF16 ? 1 : 0, n, dst);
  MRAG_HIP(hipGetLastError());
  if (!out_is_device) {
    MRAG_HIP(hipMemcpyAsync(out, dst, bytes, hipMemcpyDeviceToHost, stream));
    MRAG_HIP(hipStreamSynchronize(stream));
  }
  return MRAG_OK;
}

int mrag_index_search(mrag_handle h, const void* queries, int64_t nq, int q_dtype, int normalize,
                      int queries_is_device, int k, float* out_scores, int64_t* out_ids, int out_is_device,
                      void* stream_) {
  BfIndex* ix = (BfIndex*)lookup(h, KIND_BF_INDEX);
  if (!ix) return MRAG_ERR_INVALID;
  if (nq < 0) return fail(MRAG_ERR_INVALID, "nq < 0");
  if (k <= 0) return fail(MRAG_ERR_INVALID, "k must be positive");
  if (k > KMAX_WIDE) return fail(MRAG_ERR_UNSUPPORTED, "k = %d exceeds the top-k limit %d (see mrag_index_max_k)", k, KMAX_WIDE);
  if (nq == 0) return MRAG_OK;
  if (!queries || !out_scores || !out_ids) return fail(MRAG_ERR_INVALID, "NULL buffer");
  const size_t esz = dtype_size(q_dtype);
  if (!esz) return fail(MRAG_ERR_INVALID, "unknown query dtype %d", q_dtype);
  if (nq > (1ll << 26)) return fail(MRAG_ERR_UNSUPPORTED, "nq too large for one call; batch the queries");
  MRAG_TRY(use_device(ix->device));
  hipStream_t stream = (hipStream_t)stream_;
  ix->timed = false;

  float* d_sc = out_scores;
  int64_t* d_id = out_ids;
  if (!out_is_device) {
    MRAG_TRY(ix->out_sc.ensure((size_t)nq * k * 4));
    MRAG_TRY(ix->out_id.ensure((size_t)nq * k * 8));
    d_sc = (float*)ix->out_sc.p;
    d_id = (int64_t*)ix->out_id.p;
  }
  MRAG_HIP(hipEventRecord(ix->ev[0], stream));
  if (ix->n == 0) {
    const int64_t tot = nq * k;
    hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, d_sc, d_id, tot);
    MRAG_HIP(hipGetLastError());
    MRAG_HIP(hipEventRecord(ix->ev[1], stream));
    MRAG_HIP(hipEventRecord(ix->ev[2], stream));
  } else {
    // queries -> storage dtype, zero padded to a multiple of 256 rows
    const int64_t nq_pad = round_up(nq, TQ) + SQ;   // + 16 zero rows: the last 8-query group of the k > 64 path reads a 16-row block
    const size_t qbytes = (size_t)nq_pad * ix->ld * 2;
    MRAG_TRY(ix->qbuf.ensure(qbytes));
    const void* qsrc = queries;
    if (!queries_is_device) {
      const size_t bytes = (size_t)nq * ix->dim * esz;
      MRAG_TRY(ix->stage_in.ensure(bytes));
      MRAG_HIP(hipMemcpyAsync(ix->stage_in.p, queries, bytes, hipMemcpyHostToDevice, stream));
      qsrc = ix->stage_in.p;
    }
    // rows [nq, nq_pad) must be zero; only rows an earlier (larger) batch wrote need clearing again, so a stream of
    // equal-sized searches (the online regime: one question per call) pays the fill launch once
    if (ix->qbuf.p != ix->q_cleared_buf || nq_pad > ix->q_cleared_rows) {
      MRAG_HIP(hipMemsetAsync(ix->qbuf.p, 0, qbytes, stream));
      ix->q_cleared_buf = ix->qbuf.p; ix->q_cleared_rows = nq_pad; ix->q_dirty_rows = 0;
    } else if (ix->q_dirty_rows > nq) {
      MRAG_HIP(hipMemsetAsync((char*)ix->qbuf.p + (size_t)nq * ix->ld * 2, 0, (size_t)(ix->q_dirty_rows - nq) * ix->ld * 2, stream));
      ix->q_dirty_rows = nq;
    }
    ix->q_dirty_rows = std::max<int64_t>(ix->q_dirty_rows, nq);
    MRAG_TRY(launch_prep_rows(qsrc, q_dtype, nq, ix->dim, ix->qbuf.p, ix->ld, ix->dtype,
                              normalize && ix->metric == MRAG_METRIC_COSINE, stream));
    BfLaunch a;
    a.corpus = ix->rows;
    a.queries = (const uint16_t*)ix->qbuf.p;
    a.ld = ix->ld; a.dtype = ix->dtype; a.k = k;
    a.nq = nq; a.n_rows = ix->n;
    a.id_base = ix->id_base;
    a.out_scores = d_sc; a.out_ids = d_id;
    a.lists = &ix->lists; a.counts = &ix->counts;
    a.stream = stream;
    a.ev_k2_begin = ix->ev[1]; a.ev_k2_end = ix->ev[2];
    if (ix->want_clock) {
      MRAG_TRY(ix->clock.ensure(64));
      MRAG_HIP(hipMemsetAsync(ix->clock.p, 0, 64, stream));
      a.clock = (long long*)ix->clock.p;
    }
    MRAG_TRY(bf_launch(a));
  }
  MRAG_HIP(hipEventRecord(ix->ev[3], stream));
  ix->timed = true;
  if (!out_is_device) {
    MRAG_HIP(hipMemcpyAsync(out_scores, d_sc, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream));
    MRAG_HIP(hipMemcpyAsync(out_ids, d_id, (size_t)nq * k * 8, hipMemcpyDeviceToHost, stream));
    MRAG_HIP(hipStreamSynchronize(stream));
  }
  return MRAG_OK;
}

int mrag_index_score_rows(mrag_handle h, const void* query, int q_dtype, int normalize, const int64_t* row_ids, int64_t n,
                          float* out_scores, void* stream_) {
  BfIndex* ix = (BfIndex*)lookup(h, KIND_BF_INDEX);
  if (!ix) return MRAG_ERR_INVALID;
  if (n < 0) return fail(MRAG_ERR_INVALID, "n < 0");
  if (n == 0) return MRAG_OK;
  if (!query || !row_ids || !out_scores) return fail(MRAG_ERR_INVALID, "NULL buffer");
  const size_t esz = dtype_size(q_dtype);
  if (!esz) return fail(MRAG_ERR_INVALID, "unknown query dtype %d", q_dtype);
  MRAG_TRY(use_device(ix->device));
  hipStream_t stream = (hipStream_t)stream_;
  // staging: [query src | row ids | scores]; prepared query in qbuf (one row)
  const size_t qb = round_up((int64_t)(ix->dim * esz), 256), ib = round_up(n * 8, 256), ob = (size_t)n * 4;
  MRAG_TRY(ix->stage_in.ensure(qb + ib + ob));
  MRAG_TRY(ix->qbuf.ensure((size_t)(TQ + SQ) * ix->ld * 2));
  char* st = (char*)ix->stage_in.p;
  MRAG_HIP(hipMemcpyAsync(st, query, (size_t)ix->dim * esz, hipMemcpyHostToDevice, stream));
  MRAG_HIP(hipMemcpyAsync(st + qb, row_ids, (size_t)n * 8, hipMemcpyHostToDevice, stream));
  MRAG_TRY(launch_prep_rows(st, q_dtype, 1, ix->dim, ix->qbuf.p, ix->ld, ix->dtype,
                            normalize && ix->metric == MRAG_METRIC_COSINE, stream));
  if (ix->qbuf.p == ix->q_cleared_buf) ix->q_dirty_rows = std::max<int64_t>(ix->q_dirty_rows, 1);   // row 0 of the query buffer was written
  hipLaunchKernelGGL(score_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, ix->rows, (const uint16_t*)ix->qbuf.p,
                     ix->ld, ix->dtype == MRAG_BF16 ? 1 : 0, (const int64_t*)(st + qb), n, ix->n, (float*)(st + qb + ib));
  MRAG_HIP(hipGetLastError());
  MRAG_HIP(hipMemcpyAsync(out_scores, st + qb + ib, ob, hipMemcpyDeviceToHost, stream));
  MRAG_HIP(hipStreamSynchronize(stream));
  return MRAG_OK;
}

int mrag_index_measure_clock(mrag_handle h, int enable) {
  BfIndex* ix = (BfIndex*)lookup(h, KIND_BF_INDEX);
  if (!ix) return MRAG_ERR_INVALID;
  ix->want_clock = enable != 0;
  return MRAG_OK;
}

int mrag_index_last_clock(mrag_handle h, float* out_ghz) {
  BfIndex* ix = (BfIndex*)lookup(h, KIND_BF_INDEX);
  if (!ix) return MRAG_ERR_INVALID;
  if (!out_ghz) return fail(MRAG_ERR_INVALID, "out_ghz is NULL");
  if (!ix->timed || !ix->want_clock || !ix->clock.p) return fail(MRAG_ERR_INVALID, "no search with the clock reading enabled");
  MRAG_TRY(use_device(ix->device));
  MRAG_HIP(hipEventSynchronize(ix->ev[3]));
  long long w[4] = {0, 0, 0, 0};
  MRAG_HIP(hipMemcpy(w, ix->clock.p, sizeof(w), hipMemcpyDeviceToHost));
  const double dc = (double)(w[2] - w[0]), dr = (double)(w[3] - w[1]);
  if (dr <= 0 || dc <= 0) return fail(MRAG_ERR_UNSUPPORTED, "the last search did not run the batch kernel (no clock stamps)");
  *out_ghz = (float)(dc / dr * 0.1);   // s_memrealtime ticks at 100 MHz
  return MRAG_OK;
}

int mrag_index_last_timing(mrag_handle h, float* out_gemm_ms, float* out_total_ms) {
  BfIndex* ix = (BfIndex*)lookup(h, KIND_BF_INDEX);
  if (!ix) return MRAG_ERR_INVALID;
  if (!ix->timed) return fail(MRAG_ERR_INVALID, "no completed search to time");
  MRAG_TRY(use_device(ix->device));
  MRAG_HIP(hipEventSynchronize(ix->ev[3]));
  float g = 0.f, t = 0.f;
  MRAG_HIP(hipEventElapsedTime(&g, ix->ev[1], ix->ev[2]));
  MRAG_HIP(hipEventElapsedTime(&t, ix->ev[0], ix->ev[3]));
  if (out_gemm_ms) *out_gemm_ms = g;
  if (out_total_ms) *out_total_ms = t;
  return MRAG_OK;
}

}  // extern "C"
