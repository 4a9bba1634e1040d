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
//   Block -> (t, s) keeps the 8 query tiles x 4 splits that run together on one XCD
//   (32 CUs) sharing: 8 query tiles stay in that XCD's L2 (3 MiB at d=768), every corpus
//   tile is fetched from HBM once per 8 query tiles.
//
// Top-k filter (exact):
//   per query a threshold tau = a lower bound of its current k-th best; a score enters the
//   query's candidate list only if it beats tau.  First tile of a split: tau0 = min over
//   the 8 (lane, wave) groups that share the query of the group's 2nd-largest score, so at
//   least 16 >= k candidates pass (k <= 16; larger k start from -inf).  Later tiles: strict
//   '>' -- a later row that only ties tau loses the (score desc, row asc) tie-break to the
//   rows already listed.  Lists hold CAP = 256 entries; at HW = 192 a wave compacts the list
//   to its k best (rank by counting) and raises tau to the k-th.  Random data: ~16 ln(n/256)
//   entries per (query, split), compaction is rare.  Adversarial data (every tile beats the
//   last): the tile is replayed in 4 sub-rounds of <= 64 pushes per query with a compaction
//   between them -- slower, never wrong.
#include "common.h"

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
constexpr int OFF_TAU = GEMM_LDS;               // float[256]
constexpr int OFF_CNT = OFF_TAU + 1024;         // int[256]
constexpr int OFF_CNTPRE = OFF_CNT + 1024;      // int[256]
constexpr int OFF_STAT = OFF_CNTPRE + 1024;     // uint[2][256] threshold certificates, orderable bits
constexpr int OFF_FLAGS = OFF_STAT + 2048;      // int[4]
constexpr int LDS_TOTAL = OFF_FLAGS + 16;

constexpr int CAP = 256;   // candidate list capacity per (workgroup, query)
constexpr int HW = 192;    // compaction high-water mark (CAP - HW >= 64 = max pushes per sub-round)
constexpr int KMAX = 64;   // largest k the fused filter serves
constexpr int K_CERT = 16;  // k served by the 16-row threshold certificate

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
  int xcd_map;              // 1: S % 8 == 0, use the XCD-aware block -> (t, s) map
  int dbg;                  // development ablations (MRAG_DEBUG_FLAGS); 0 in production
  long long* stamps;        // dbg & 16: block 0 / wave 0 writes s_memtime stamps here
  uint32_t* list_sc;        // [T*S][256][CAP]  score bits   (SoA: two dword stores per push, no
  uint32_t* list_row;       // [T*S][256][CAP]  local row      64-bit store operands to pre-form)
  int* counts;              // [T*S][256]
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

#define MRAG_STAMP(i)                                                                     \
  do {                                                                                    \
    if ((p.dbg & 16) && blockIdx.x == 0 && tid == 0 && stamp_n < 64) {                    \
      p.stamps[stamp_n * 2] = (i);                                                        \
      p.stamps[stamp_n * 2 + 1] = (long long)__builtin_readcyclecounter();                \
      ++stamp_n;                                                                          \
    }                                                                                     \
  } while (0)

// One wave compacts query q's list to its k best (sorted, best first) and raises tau_c.
// Rank by counting over <= CAP entries; rare by construction (see file header).
__device__ __forceinline__ void compact_query(uint32_t* __restrict__ lsc, uint32_t* __restrict__ lrow, int q, int k, int lane) {
  int* cnt = (int*)(smem + OFF_CNT);
  float* tau_c = (float*)(smem + OFF_TAU);
  const int c = min(cnt[q], CAP);
  constexpr int R = CAP / 64;
  uint2 ent[R];
  uint64_t key[R];
  int rank[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int idx = r * 64 + lane;
    ent[r] = make_uint2(0u, 0u);
    key[r] = 0ull;
    rank[r] = 0;
    if (idx < c) {
      ent[r] = make_uint2(lsc[idx], lrow[idx]);
      key[r] = make_key(ent[r].x, ent[r].y);
    }
  }
#pragma unroll
  for (int r2 = 0; r2 < R; ++r2) {
    const int lim = min(64, c - r2 * 64);
    const uint32_t khi = (uint32_t)(key[r2] >> 32), klo = (uint32_t)key[r2];
    for (int jj = 0; jj < lim; ++jj) {
      const uint64_t kj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)khi, jj) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)klo, jj);
#pragma unroll
      for (int r = 0; r < R; ++r) rank[r] += (kj > key[r]) ? 1 : 0;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int idx = r * 64 + lane;
    if (idx < c && rank[r] < k) {
      lsc[rank[r]] = ent[r].x;
      lrow[rank[r]] = ent[r].y;
      if (rank[r] == k - 1) tau_c[q] = __uint_as_float(ent[r].x);
    }
  }
  if (lane == 0) cnt[q] = min(c, k);
}

// Owner wave w (queries 32w .. 32w+31) compacts every list at or above the high-water mark.
__device__ __forceinline__ void compact_owned(uint32_t* __restrict__ wg_sc, uint32_t* __restrict__ wg_row, int w, int lane, int k, bool record_pre) {
  int* cnt = (int*)(smem + OFF_CNT);
  int* cnt_pre = (int*)(smem + OFF_CNTPRE);
  const int q = w * 32 + (lane & 31);
  const bool need = (lane < 32) && (cnt[q] >= HW);
  unsigned long long m = __ballot(need);
  while (m) {
    const int b = __builtin_ctzll(m);
    m &= m - 1;
    const int qq = w * 32 + b;
    compact_query(wg_sc + (size_t)qq * CAP, wg_row + (size_t)qq * CAP, qq, k, lane);
  }
  if (record_pre && lane < 32) cnt_pre[q] = cnt[q];
}

template <int DT>
__global__ __launch_bounds__(NTHR, 2) void bf_gemm_topk_kernel(BfParams p) {
  typedef typename Mfma<DT>::frag frag;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  int stamp_n = 0;
  MRAG_STAMP(0);

  // ---- block -> (query tile t, corpus split s) ------------------------------------------
  int t, s;
  {
    const int b = blockIdx.x;
    if (p.xcd_map) {
      const int x = b & 7, idx = b >> 3, sx = p.S >> 3;
      const int full = (p.T >> 3) * sx * 8;
      int g, j, tl;
      if (idx < full) {
        g = idx / (sx * 8);
        const int rem = idx - g * (sx * 8);
        j = rem >> 3;
        tl = rem & 7;
      } else {
        const int sz = p.T & 7, r2 = idx - full;
        g = p.T >> 3;
        j = r2 / sz;
        tl = r2 - j * sz;
      }
      t = g * 8 + tl;
      s = x + 8 * j;
    } else {
      s = b % p.S;
      t = b / p.S;
    }
  }
  const int wg = t * p.S + s;
  uint32_t* wg_sc = p.list_sc + (size_t)wg * TQ * CAP;
  uint32_t* wg_row = p.list_row + (size_t)wg * TQ * CAP;
  const int tile_lo = (int)(((long long)s * p.n_ctiles) / p.S);
  const int tile_hi = (int)(((long long)(s + 1) * p.n_ctiles) / p.S);
  const int ksteps = p.ksteps;
  const int n_steps = (tile_hi - tile_lo) * ksteps;

  float* tau_c = (float*)(smem + OFF_TAU);
  int* cnt = (int*)(smem + OFF_CNT);
  int* cnt_pre = (int*)(smem + OFF_CNTPRE);
  uint32_t* stat = (uint32_t*)(smem + OFF_STAT);   // [2][256] orderable bits, atomicMin target
  int* flags = (int*)(smem + OFF_FLAGS);
  if (tid < TQ) {
    tau_c[tid] = -INFINITY;
    cnt[tid] = 0;
    cnt_pre[tid] = 0;
    stat[tid] = 0xFFFFFFFFu;
    stat[TQ + tid] = 0xFFFFFFFFu;
  }
  if (tid == 0) flags[0] = 0;

  // ---- LDS-DMA source offsets: wave w fills 1-KiB chunks 4w..4w+3 of each operand ---------
  // chunk c = 8 rows x 128 B; lane l -> row r = 8c + (l>>3), physical 16-B chunk l&7,
  // logical chunk (l&7) ^ ((r>>1)&7)
  int src_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (4 * w + i) * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((r >> 1) & 7);
    src_off[i] = r * p.ld + kc * 8;
  }
  const uint16_t* qbase = p.queries + (size_t)t * TQ * p.ld;

  auto stage = [&](int step, int buf) {
    const int tile = tile_lo + step / ksteps;
    const int kk = step - (step / ksteps) * ksteps;
    const uint16_t* a = p.corpus + (size_t)tile * TM * p.ld + kk * BK;
    const uint16_t* b = qbase + kk * BK;
    char* la = smem + buf * STAGE_BYTES + (4 * w) * 1024;
    char* lb = la + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_vptr)(a + src_off[i]), (lds_vptr)(la + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_vptr)(b + src_off[i]), (lds_vptr)(lb + i * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets ------------------------------------------------------------
  const int frow = lane & 15;
  const int fsw = frow >> 1;
  const int a_rd = (wm * 128 + frow) * 128;
  const int b_rd = A_BYTES + (wn * 64 + frow) * 128;
  const int ph0 = (((lane >> 4)) ^ fsw) * 16;      // k sub-step 0: logical chunks 0..3
  const int ph1 = ((4 + (lane >> 4)) ^ fsw) * 16;  // k sub-step 1: logical chunks 4..7

  f32x4 acc[8][4];
#pragma unroll
  for (int mf = 0; mf < 8; ++mf)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // running two largest per-tile maxima of this lane, per query column (k <= 16 certificate)
  float rm1[4], rm2[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) rm1[nf] = rm2[nf] = -INFINITY;

  if (n_steps > 0) stage(0, 0);
  __syncthreads();
  MRAG_STAMP(1);

  for (int step = 0; step < n_steps; ++step) {
    const int buf = step & 1;
    if (step + 1 < n_steps) stage(step + 1, buf ^ 1);

    const char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ph = ks ? ph1 : ph0;
      frag af[8], bfr[4];
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) af[mf] = *(const frag*)(sb + a_rd + mf * 2048 + ph);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) bfr[nf] = *(const frag*)(sb + b_rd + nf * 2048 + ph);
#pragma unroll
      for (int mf = 0; mf < 8; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = Mfma<DT>::run(af[mf], bfr[nf], acc[mf][nf]);
    }

    const int kk = step % ksteps;
    if (kk == ksteps - 1 && (p.dbg & 1)) {
      // ablation: no top-k filter; keep the accumulators observable, then clear them
      float x = 0.f;
#pragma unroll
      for (int mf = 0; mf < 8; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          x += acc[mf][nf][0] + acc[mf][nf][1] + acc[mf][nf][2] + acc[mf][nf][3];
          acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      if (x == 123456.789f) cnt[0] = 1;
    } else if (kk == ksteps - 1) {
      // =========================== fused top-k epilogue ===================================
      const int ti = step / ksteps;  // tile index inside the split
      const int tile = tile_lo + ti;
      // lane-derived values are recomputed from an opaque copy so that nothing the epilogue
      // needs is hoisted out of the K loop (the loop body is near the 256-VGPR budget)
      int elane = lane;
      asm volatile("" : "+v"(elane));
      const int row0 = tile * TM + wm * 128 + (elane >> 4) * 4;  // + mf*16 + j
      const int q0 = wn * 64 + (elane & 15);                     // + nf*16
      const bool certify = p.k <= K_CERT;
      MRAG_STAMP(10 + (ti == 0 ? 0 : 100));
      if ((tile + 1) * TM > p.n_rows) {
#pragma unroll
        for (int mf = 0; mf < 8; ++mf)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (row0 + mf * 16 + j >= p.n_rows) {
#pragma unroll
              for (int nf = 0; nf < 4; ++nf) acc[mf][nf][j] = -INFINITY;
            }
      }
      // per-lane maximum of the tile, per query column
      float tmax[4];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        float m = acc[0][nf][0];
#pragma unroll
        for (int mf = 0; mf < 8; ++mf)
#pragma unroll
          for (int j = 0; j < 4; ++j) m = fmaxf(m, acc[mf][nf][j]);
        tmax[nf] = m;
      }
      float thr[4];
      uint32_t* stat_cur = stat + (ti & 1) * TQ;        // certificate built from tiles <= ti
      uint32_t* stat_prev = stat + ((ti & 1) ^ 1) * TQ;  // certificate from tiles < ti
      if (ti == 0) {
        // ---- first tile: tau0 = min over the query's 8 lane groups of the group's 2nd largest
        // score IN this tile; >= 16 rows are >= tau0, so pass on '>=' (as '> next_below').
        if (certify) {
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) {
            float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
            for (int mf = 0; mf < 8; ++mf)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float v = acc[mf][nf][j];
                m2 = fmaxf(m2, fminf(m1, v));
                m1 = fmaxf(m1, v);
              }
            rm1[nf] = m1;
            rm2[nf] = m2;
            float x = m2;
            x = fminf(x, __shfl_xor(x, 16));
            x = fminf(x, __shfl_xor(x, 32));
            if (elane < 16) atomicMin(&stat_cur[q0 + nf * 16], f32_ord(x));
          }
          __syncthreads();
          MRAG_STAMP(11);
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) thr[nf] = next_below(ord_f32(stat_cur[q0 + nf * 16]));
        } else {
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) thr[nf] = -INFINITY;
        }
      } else {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          const float sc = certify ? ord_f32(stat_prev[q0 + nf * 16]) : -INFINITY;
          thr[nf] = fmaxf(sc, tau_c[q0 + nf * 16]);
        }
      }
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        if (t * TQ + q0 + nf * 16 >= p.nq) thr[nf] = INFINITY;   // padding queries never list anything

      // ---- push: attempt 0 = whole tile at once; on list overflow replay in 4 sub-rounds ------
      int attempt = 0, round = 0;
      while (true) {
        const uint32_t srmask = attempt ? (0xFFu << (8 * round)) : 0xFFFFFFFFu;
        int rbase = row0;
        asm volatile("" : "+v"(rbase));   // keep the 128 {score,row} store operands from being pre-formed (and spilled) outside this loop
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          if (!__any(tmax[nf] > thr[nf])) continue;
          uint32_t mask = 0;
#pragma unroll
          for (int mf = 0; mf < 8; ++mf)
#pragma unroll
            for (int j = 0; j < 4; ++j) mask |= (acc[mf][nf][j] > thr[nf]) ? (1u << (mf * 4 + j)) : 0u;
          mask &= srmask;
          const int q = q0 + nf * 16;
          const int n_new = __popc(mask);
          int base = 0;
          if (n_new) {
            base = atomicAdd(&cnt[q], n_new);
            if (base + n_new > CAP) flags[0] = 1;
          }
          uint32_t* lsc = wg_sc + (size_t)q * CAP;
          uint32_t* lrow = wg_row + (size_t)q * CAP;
#pragma unroll
          for (int mf = 0; mf < 8; ++mf) {
            if (!__any((mask >> (mf * 4)) & 0xFu)) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int bit = mf * 4 + j;
              if (mask & (1u << bit)) {
                const int slot = base + __popc(mask & ((1u << bit) - 1u));
                if (slot < CAP) {
                  lsc[slot] = __float_as_uint(acc[mf][nf][j]);
                  lrow[slot] = (uint32_t)(rbase + mf * 16 + j);
                }
              }
            }
          }
        }
        if (attempt == 0 && certify) {
          // fold this tile's maxima into the running certificate used from the NEXT tile on
          if (ti != 0) {
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
              rm2[nf] = fmaxf(rm2[nf], fminf(rm1[nf], tmax[nf]));
              rm1[nf] = fmaxf(rm1[nf], tmax[nf]);
              float x = rm2[nf];
              x = fminf(x, __shfl_xor(x, 16));
              x = fminf(x, __shfl_xor(x, 32));
              if (elane < 16) atomicMin(&stat_cur[q0 + nf * 16], f32_ord(x));
            }
          }
        }
        MRAG_STAMP(12 + (ti == 0 ? 0 : 100));
        __syncthreads();
        MRAG_STAMP(13 + (ti == 0 ? 0 : 100));
        if (attempt == 0) {
          if (!flags[0]) {
            compact_owned(wg_sc, wg_row, w, elane, p.k, true);
            if (tid < TQ) stat_prev[tid] = 0xFFFFFFFFu;   // becomes stat_cur of tile ti+1
            break;
          }
          __syncthreads();               // everyone has seen the flag
          if (tid < TQ) cnt[tid] = cnt_pre[tid];
          if (tid == 0) flags[0] = 0;
          __syncthreads();
          attempt = 1;
          round = 0;
        } else {
          compact_owned(wg_sc, wg_row, w, elane, p.k, round == 3);
          __syncthreads();
          if (round == 3) {
            if (tid < TQ) stat_prev[tid] = 0xFFFFFFFFu;
            break;
          }
          ++round;
        }
        if (attempt) {
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) {
            const float sc = (certify && ti != 0) ? ord_f32(stat_prev[q0 + nf * 16]) : -INFINITY;
            float th = fmaxf(sc, tau_c[q0 + nf * 16]);
            if (t * TQ + q0 + nf * 16 >= p.nq) th = INFINITY;
            thr[nf] = fmaxf(thr[nf], th);
          }
        }
      }
      MRAG_STAMP(14 + (ti == 0 ? 0 : 100));
#pragma unroll
      for (int mf = 0; mf < 8; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
  }

  __syncthreads();
  MRAG_STAMP(99);
  if (tid < TQ) p.counts[(size_t)wg * TQ + tid] = min(cnt[tid], CAP);
}

// ------------------------------------------------------------------------------------------
// K4: one wave per query selects the k best of the S candidate lists.
// ------------------------------------------------------------------------------------------
constexpr int MERGE_LDS_ENT = 4096;

struct MergeParams {
  const uint32_t* list_sc;
  const uint32_t* list_row;
  const int* counts;
  int T, S, k;
  int64_t nq;
  int64_t id_base;
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

__global__ __launch_bounds__(64) void bf_merge_kernel(MergeParams p) {
  __shared__ uint64_t keys[MERGE_LDS_ENT];
  __shared__ uint64_t best[KMAX];
  const int lane = threadIdx.x;
  const int64_t q = blockIdx.x;
  if (q >= p.nq) return;
  const int t = (int)(q / TQ), ql = (int)(q % TQ);
  const int k = p.k;
  int nbest = 0;  // entries of `best` carried into the next chunk
  int s = 0, pos = 0;
  bool done = false;
  while (!done) {
    // fill keys[0..fill) with carried best + as many list entries as fit
    int fill = 0;
    for (int i = lane; i < nbest; i += 64) keys[i] = best[i];
    fill = nbest;
    while (s < p.S) {
      const int wg = t * p.S + s;
      const int c = p.counts[(size_t)wg * TQ + ql];
      const uint32_t* lsc = p.list_sc + ((size_t)wg * TQ + ql) * CAP;
      const uint32_t* lrow = p.list_row + ((size_t)wg * TQ + ql) * CAP;
      const int take = min(c - pos, MERGE_LDS_ENT - fill);
      for (int i = lane; i < take; i += 64) {
        keys[fill + i] = make_key(lsc[pos + i], lrow[pos + i]);
      }
      fill += take;
      pos += take;
      if (pos >= c) { ++s; pos = 0; }
      if (fill >= MERGE_LDS_ENT) break;
    }
    done = (s >= p.S);
    __syncthreads();
    // extract the k best of keys[0..fill)
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
    nbest = nsel;
    __syncthreads();
  }
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

// ------------------------------------------------------------------------------------------
struct BfIndex : Object {
  int dim = 0, ld = 0, metric = 0, dtype = MRAG_F16;
  int64_t n = 0, cap_rows = 0, id_base = 0;
  uint16_t* rows = nullptr;
  DevBuf qbuf, lists, counts, stage_in, out_sc, out_id;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool timed = false;
  ~BfIndex() override {
    if (rows) (void)hipFree(rows);
    qbuf.release(); lists.release(); counts.release(); stage_in.release(); out_sc.release(); out_id.release();
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
  }
};

static long long* g_stamps = nullptr;  // diagnostic s_memtime stamps (MRAG_DEBUG_FLAGS & 16)

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
  if (n_ctiles < 8) { *S_out = std::max(1, n_ctiles); *xcd_out = 0; return; }
  int g = T, b = 256;
  while (b) { int r = g % b; g = b; b = r; }   // gcd(T, 256)
  int S = 256 / g;
  if (S < 8) S = 8;
  if (S > n_ctiles) S = (n_ctiles / 8) * 8;
  *S_out = S;
  *xcd_out = 1;
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
  return grow_rows(ix, n_rows, nullptr);
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
  if (k > KMAX) return fail(MRAG_ERR_UNSUPPORTED, "k = %d exceeds the fused top-k limit %d", k, KMAX);
  if (nq == 0) return MRAG_OK;
  if (!queries || !out_scores || !out_ids) return fail(MRAG_ERR_INVALID, "NULL buffer");
  const size_t esz = dtype_size(q_dtype);
  if (!esz) return fail(MRAG_ERR_INVALID, "unknown query dtype %d", q_dtype);
  if (nq > (1ll << 24)) return fail(MRAG_ERR_UNSUPPORTED, "nq too large for one call; batch the queries");
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
    const int T = (int)((nq + TQ - 1) / TQ);
    const int n_ctiles = (int)((ix->n + TM - 1) / TM);
    int S, xcd;
    choose_split(T, n_ctiles, &S, &xcd);
    const size_t grid = (size_t)T * S;
    // queries -> storage dtype, zero padded to T*256 rows
    const size_t qbytes = (size_t)T * TQ * ix->ld * 2;
    MRAG_TRY(ix->qbuf.ensure(qbytes));
    const void* qsrc = queries;
    if (!queries_is_device) {
      const size_t bytes = (size_t)nq * ix->dim * esz;
      MRAG_TRY(ix->stage_in.ensure(bytes));
      MRAG_HIP(hipMemcpyAsync(ix->stage_in.p, queries, bytes, hipMemcpyHostToDevice, stream));
      qsrc = ix->stage_in.p;
    }
    if ((int64_t)T * TQ > nq)
      MRAG_HIP(hipMemsetAsync((char*)ix->qbuf.p + (size_t)nq * ix->ld * 2, 0, qbytes - (size_t)nq * ix->ld * 2, stream));
    MRAG_TRY(launch_prep_rows(qsrc, q_dtype, nq, ix->dim, ix->qbuf.p, ix->ld, ix->dtype,
                              normalize && ix->metric == MRAG_METRIC_COSINE, stream));
    MRAG_TRY(ix->lists.ensure(grid * TQ * CAP * 8));
    MRAG_TRY(ix->counts.ensure(grid * TQ * sizeof(int)));

    BfParams p;
    p.corpus = ix->rows;
    p.queries = (const uint16_t*)ix->qbuf.p;
    p.ld = ix->ld;
    p.ksteps = ix->ld / BK;
    p.n_rows = (int)ix->n;
    p.n_ctiles = n_ctiles;
    p.nq = (int)nq;
    p.T = T;
    p.S = S;
    p.k = k;
    p.xcd_map = xcd;
    {
      static int dbg = -1;
      if (dbg < 0) { const char* e = getenv("MRAG_DEBUG_FLAGS"); dbg = e ? atoi(e) : 0; }
      p.dbg = dbg;
    }
    p.stamps = nullptr;
    if (p.dbg & 16) {
      if (!g_stamps) MRAG_HIP(hipMalloc((void**)&g_stamps, 128 * 8));
      MRAG_HIP(hipMemsetAsync(g_stamps, 0, 128 * 8, stream));
      p.stamps = g_stamps;
    }
    p.list_sc = (uint32_t*)ix->lists.p;
    p.list_row = p.list_sc + grid * TQ * CAP;
    p.counts = (int*)ix->counts.p;
    static bool attr_done[2] = {false, false};
    if (ix->dtype == MRAG_F16) {
      if (!attr_done[0]) { MRAG_HIP(hipFuncSetAttribute((const void*)bf_gemm_topk_kernel<MRAG_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL)); attr_done[0] = true; }
      MRAG_HIP(hipEventRecord(ix->ev[1], stream));
      hipLaunchKernelGGL((bf_gemm_topk_kernel<MRAG_F16>), dim3((unsigned)grid), dim3(NTHR), LDS_TOTAL, stream, p);
    } else {
      if (!attr_done[1]) { MRAG_HIP(hipFuncSetAttribute((const void*)bf_gemm_topk_kernel<MRAG_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL)); attr_done[1] = true; }
      MRAG_HIP(hipEventRecord(ix->ev[1], stream));
      hipLaunchKernelGGL((bf_gemm_topk_kernel<MRAG_BF16>), dim3((unsigned)grid), dim3(NTHR), LDS_TOTAL, stream, p);
    }
    MRAG_HIP(hipGetLastError());
    MRAG_HIP(hipEventRecord(ix->ev[2], stream));

    MergeParams mp;
    mp.list_sc = p.list_sc;
    mp.list_row = p.list_row;
    mp.counts = p.counts;
    mp.T = T; mp.S = S; mp.k = k;
    mp.nq = nq;
    mp.id_base = ix->id_base;
    mp.out_scores = d_sc;
    mp.out_ids = d_id;
    hipLaunchKernelGGL(bf_merge_kernel, dim3((unsigned)nq), dim3(64), 0, stream, mp);
    MRAG_HIP(hipGetLastError());
  }
  MRAG_HIP(hipEventRecord(ix->ev[3], stream));
  ix->timed = true;
  if (g_stamps && ix->n > 0) {   // diagnostic build path only (MRAG_DEBUG_FLAGS & 16)
    long long hs[128];
    MRAG_HIP(hipStreamSynchronize(stream));
    MRAG_HIP(hipMemcpy(hs, g_stamps, sizeof(hs), hipMemcpyDeviceToHost));
    fprintf(stderr, "[mrag stamps]");
    for (int i = 0; i < 64 && (i == 0 || hs[2 * i + 1]); ++i)
      fprintf(stderr, " %lld:%lld", hs[2 * i], hs[2 * i + 1] - hs[1]);
    fprintf(stderr, "\n");
  }
  if (!out_is_device) {
    MRAG_HIP(hipMemcpyAsync(out_scores, d_sc, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream));
    MRAG_HIP(hipMemcpyAsync(out_ids, d_id, (size_t)nq * k * 8, hipMemcpyDeviceToHost, stream));
    MRAG_HIP(hipStreamSynchronize(stream));
  }
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
