// IVF-flat, "score segments + select" regime (BASELINE.json config 5; the reference has no IVF index).
//
// The list scan of an IVF search is many SMALL similarity problems: ~150 rows x ~80 queries per list at
// nlist 4096 / nprobe 32 / 10 000 queries.  The fused GEMM + top-k kernel (bf_index.hip, descriptor mode)
// spends them one 256 x 256 workgroup per CU at a time: its K loop waits out an HBM round trip per K step
// and its top-k bootstrap / end-of-split compaction costs as much as the scores (IVF section).
// Here the two halves are separate kernels:
//
//   ivfs_scan_kernel    one workgroup = one descriptor: <= 128 queries x a run of corpus rows, as 128 x 128 x 64
//                       MFMA tiles fed by LDS-DMA (two 32-KiB stages, two workgroups per CU so that one's loads
//                       overlap the other's MFMAs).  The QUERY rows are gathered by per-lane source addresses,
//                       so no gathered copy of the queries is ever written.  Scores go to HBM as fp32
//                       segments S[query slot][row] (16-byte stores, 4 consecutive rows of one query per lane).
//                       Same MFMA instruction and K order as the fused kernel: the scores are bit-identical to it.
//   ivfs_select_*_kernel one workgroup per query: its nprobe segments (a few thousand scores, L2-resident) are
//                       read once into LDS as orderable 32-bit keys, an MSB-first radix select finds the k-th
//                       largest score, ties on it are resolved by original row (ascending), the k winners are
//                       ranked by counting.  Order = (score desc, original row asc), as everywhere in this library.
//
// The same two kernels serve the probe selection (top-nprobe over the centroids): one "list" holding all the
// centroids, descriptors cut it into (128 queries, 512 centroids) pieces, one segment per query.
// Bytes: rows x ld x 2 from HBM once per list (the workgroups of a list run on one XCD and share it in L2),
// + 4 B per (query, row) score written and read back.  When the segments of one search would not fit the
// score buffer (exhaustive probing of a big index) the caller falls back to the fused kernel.
#include "common.h"

#include <stdlib.h>
#include <algorithm>

namespace mrag {

typedef _Float16 s_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s_bf16x8 __attribute__((ext_vector_type(8)));
typedef float s_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* s_lds_vptr;
typedef const __attribute__((address_space(1))) void* s_glb_vptr;

template <int DT> struct SMfma;
template <> struct SMfma<MRAG_F16> {
  typedef s_f16x8 frag;
  static __device__ __forceinline__ s_f32x4 run(frag a, frag b, s_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct SMfma<MRAG_BF16> {
  typedef s_bf16x8 frag;
  static __device__ __forceinline__ s_f32x4 run(frag a, frag b, s_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ uint32_t s_f32_ord(float f) {          // order-preserving fp32 -> uint32
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float s_ord_f32(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

constexpr int ST = 128, SK = 64, STHR = 256;
// timing-only ablations (make EXTRA=-DMRAG_IVFS_DIAG=<bits>; results are wrong by design):
//   1 no score stores | 2 no MFMA | 4 queries not gathered (slot s reads row s) | 8 no corpus loads
//   128 clock stamps of every workgroup into `dbg` (tools/ivf_stamps2.py) | 256 load and multiply the padding too
//   16 scores stored as the register image (1 KiB per instruction) | 32 non-temporal score stores | 64 stores into a small window
#ifdef MRAG_IVFS_DIAG
#define IVFS_DBG(bit) ((MRAG_IVFS_DIAG) & (bit))
#else
#define IVFS_DBG(bit) 0
#endif
constexpr int S_A_BYTES = ST * SK * 2;          // 16 KiB: corpus rows of a stage
constexpr int S_STAGE = 2 * S_A_BYTES;          // 32 KiB

// descriptor words (IVFS_DESC_WORDS ints per workgroup), see common.h
// Two LDS stages; deeper rings (3 / 4 stages behind counted vmcnt waits) were measured and change nothing: in-kernel
// stamps show 1.85 us per K step per workgroup with two workgroups per CU = 35 GB/s per CU of L2 -> LDS traffic, the
// same request-side bound as the fused kernel's stream.  What pays is not fetching padding: a workgroup
// loads and multiplies only the 8-row chunks / 16-row fragments that hold real rows and real queries.
template <int DT>
__global__ __launch_bounds__(STHR) void ivfs_scan_kernel(const uint16_t* __restrict__ corpus, const uint16_t* __restrict__ queries,
                                                         int ld, int64_t zero_row, const int* __restrict__ desc, int n_desc,
                                                         const int* __restrict__ n_desc_dev, const int64_t* __restrict__ gq,
                                                         float* __restrict__ S, long long* __restrict__ dbg) {
  typedef typename SMfma<DT>::frag frag;
  constexpr int NS = 2;
  __shared__ __attribute__((aligned(16))) char sm[NS * S_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  // blocks go to XCDs round-robin: give each XCD a contiguous run of descriptors (the workgroups of one list
  // are neighbours, so the list is fetched from HBM into ONE L2)
  const int n_real = n_desc_dev ? *n_desc_dev : n_desc;     // (n_desc_dev: the count is still on the device)
  const int per = (n_real + 7) >> 3;
  // persistent: the grid is two workgroups per CU, each walks its XCD's run of descriptors with stride gridDim/8
  for (int it = (int)(blockIdx.x >> 3); it < per; it += (int)(gridDim.x >> 3)) {
  const int lin = (int)(blockIdx.x & 7) * per + it;
  if (lin >= n_real) break;
  __syncthreads();                                           // every wave is done with the previous descriptor's LDS stages
#define IVFS_STAMP(i) do { if (IVFS_DBG(128) && dbg && tid == 0) dbg[(size_t)lin * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  IVFS_STAMP(0);
  const int* d = desc + (size_t)lin * IVFS_DESC_WORDS;
  const int gq_base = d[0], nq_local = d[1], n_rows = d[3], r_off = d[4], pitch = d[7];
  const int64_t row0 = (int64_t)(uint32_t)d[2];
  const int64_t soff = (int64_t)(uint32_t)d[5] | ((int64_t)d[6] << 32);
  const int n_tiles = (n_rows + ST - 1) / ST;
  const int nk = ld / SK;

  // per-lane LDS-DMA sources: wave w fills 1-KiB chunks 4w..4w+3 of each operand (8 rows x 128 B per chunk),
  // the 16-byte piece a lane fetches is XOR-swizzled by its row so that the fragment reads are conflict-free
  const uint16_t* a_src[4];
  const uint16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (4 * w + i) * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((r >> 1) & 7);
    a_src[i] = corpus + (size_t)(row0 + r) * ld + kc * 8;
    int64_t qrow;
    if (gq_base < 0) qrow = (int64_t)(-1 - gq_base) + r;                 // identity: queries first .. first + 127 (the buffer is padded)
    else if (IVFS_DBG(4)) qrow = r;
    else { qrow = r < nq_local ? gq[(size_t)gq_base + r] : -1; if (qrow < 0) qrow = zero_row; }
    b_src[i] = queries + (size_t)qrow * ld + kc * 8;
  }
  const int n_steps = n_tiles * nk;
  // issue the loads of the next K step: only the 1-KiB chunks (8 rows) that hold rows of the list / queries of the group
  // (the rest of the LDS stage keeps whatever it held: those rows' scores are never stored)
  int st_t = 0, st_kk = 0, st_buf = 0;                                    // cursor of the NEXT step to stage
  auto stage_next = [&]() {
    char* la = sm + st_buf * S_STAGE + (4 * w) * 1024;
    char* lb = la + S_A_BYTES;
    const size_t a_adv = (size_t)st_t * ST * ld + (size_t)st_kk * SK;
    const int rows_left = IVFS_DBG(256) ? ST : n_rows - st_t * ST, q_left = IVFS_DBG(256) ? ST : nq_local;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if ((4 * w + i) * 8 < rows_left && !IVFS_DBG(8))
        __builtin_amdgcn_global_load_lds((s_glb_vptr)(a_src[i] + a_adv), (s_lds_vptr)(la + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if ((4 * w + i) * 8 < q_left)
        __builtin_amdgcn_global_load_lds((s_glb_vptr)(b_src[i] + st_kk * SK), (s_lds_vptr)(lb + i * 1024), 16, 0, 0);
    if (++st_kk == nk) { st_kk = 0; ++st_t; }
    st_buf ^= 1;
  };
  const int frow = lane & 15, fsw = frow >> 1;
  const int a_rd = (wm * 64 + frow) * 128, b_rd = S_A_BYTES + (wn * 64 + frow) * 128;
  const int ph0 = ((lane >> 4) ^ fsw) * 16;
  s_f32x4 acc[4][4];
  float* Sb = S + (IVFS_DBG(64) ? (int64_t)(lin & 63) * 32768 : soff);   // diag 64: every workgroup stores into a small L2-resident window
  int staged = 0;
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (staged < n_steps) { stage_next(); ++staged; }
  int kk = 0, tile = 0, buf = 0;
  IVFS_STAMP(1);
  for (int step = 0; step < n_steps; ++step) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stage `step` have landed (and its score stores of the previous tile)
    __builtin_amdgcn_s_barrier();          // everyone's pieces of the stage are in LDS; everyone is done with the buffer restaged below
    asm volatile("" ::: "memory");
    if (step == 0) IVFS_STAMP(2);
    if (staged < n_steps) { stage_next(); ++staged; }
    const char* sb = sm + buf * S_STAGE;
    if (kk == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (s_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // this wave's 64 x 64 piece: only the 16-row / 16-query fragments that hold real rows / queries (wave-uniform tests)
    const int ni = IVFS_DBG(256) ? 4 : min(4, (n_rows - tile * ST - wm * 64 + 15) >> 4);
    const int nj = IVFS_DBG(256) ? 4 : min(4, (nq_local - wn * 64 + 15) >> 4);
    if (ni > 0 && nj > 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ph = ks ? (ph0 ^ 64) : ph0;
        frag af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < ni) af[i] = *(const frag*)(sb + a_rd + i * 2048 + ph);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < nj) bf[j] = *(const frag*)(sb + b_rd + j * 2048 + ph);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (i < ni && j < nj) { if (!IVFS_DBG(2)) acc[i][j] = SMfma<DT>::run(af[i], bf[j], acc[i][j]); else acc[i][j][0] += (float)af[i][0] + (float)bf[j][0]; }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of `buf` are done before it reaches the next barrier
    buf ^= 1;
    if (++kk == nk) {
      if (tile == n_tiles - 1) IVFS_STAMP(3);
      // scores of this tile: lane holds rows r0..r0+3 (consecutive) of query slot q per accumulator
      const int rl = tile * ST + wm * 64 + (lane >> 4) * 4;
      const int rows4 = (n_rows + 3) & ~3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = wn * 64 + j * 16 + (lane & 15);
        if (q >= nq_local) continue;
        float* sq = Sb + (size_t)q * pitch + r_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r0 = rl + i * 16;
          if (IVFS_DBG(16)) { if (r0 < rows4) *(s_f32x4*)(Sb + ((size_t)((tile * 4 + w) * 16 + i * 4 + j) * 64 + lane) * 4) = acc[i][j]; }   // register image: 1 KiB per instruction
          else if (IVFS_DBG(32)) { if (r0 < rows4) __builtin_nontemporal_store(acc[i][j], (s_f32x4*)(sq + r0)); }
          else if (r0 < rows4 && (!IVFS_DBG(1) || acc[i][j][0] == 123456.789f)) *(s_f32x4*)(sq + r0) = acc[i][j];
        }
      }
      kk = 0;
      ++tile;
      if (IVFS_DBG(128) && tile == n_tiles) {
        IVFS_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        IVFS_STAMP(5);
        if (dbg && tid == 0) { dbg[(size_t)lin * 8 + 6] = n_steps; dbg[(size_t)lin * 8 + 7] = (long long)nq_local << 32 | (unsigned)n_rows; }
      }
    }
  }
  }
}

// ------------------------------------------------------------------ per-query selection
constexpr int SEL_THR = 256;
constexpr int SEL_EPT = 32;                        // candidate scores a thread keeps in registers
constexpr int SEL_CAP = SEL_THR * SEL_EPT;         // 8192 per query on the fast path; beyond it the serial (exact) path
constexpr int SEL_CAND = 512;                      // shortlisted entries (>= the k-th largest thread maximum)
constexpr int SEL_MAXP = 256;                      // probes per query
constexpr int SEL_MAXK = 256;

struct SelParams {
  const float* S;
  const int4* pinfo;          // lists: per (query, probe) the score segment (float offset low / high, rows, first stored row); dense: unused
  const int64_t* row_ids;     // lists: stored position -> original row; dense: nullptr (row = position)
  int nprobe, k;
  int dense_rows;             // dense: rows per segment
  int64_t dense_pitch;        // dense: floats between segments
  int64_t id_base;
  float* out_scores;          // [nq][k] or nullptr
  int64_t* out_ids;           // [nq][k]
  int* count_out;             // dense, optional: histogram of the selected rows ...
  const int* row_weight;      // ... counting only rows with row_weight > 0
};

// Selection = (score desc, original row asc), exact.
//   fast path  every thread keeps its <= 32 scores in registers (entry e = tid + 256 i).  The k-th largest of the 256
//              THREAD MAXIMA is a lower bound of the k-th largest score (k distinct entries reach it) and sits within
//              a few ranks of it, so the entries >= that bound are a shortlist of ~k(1 + k/256): they get their original
//              rows, are ranked by counting on (score, row), and ranks < k are the answer, already in order.
//              The bound comes from a bitwise bisection (2 bits per step) with ballot counts -- 16 light steps.
//   serial path (more than 8192 candidates, or a shortlist overflow = degenerate ties): k rounds of a workgroup-wide
//              arg-max over the segments, rows looked up only for entries tied with the current score.
template <bool DENSE>
__device__ __forceinline__ void ivfs_select_body(const SelParams& p) {
  __shared__ const float* seg_ptr[SEL_MAXP];
  __shared__ int seg_cnt[SEL_MAXP];
  __shared__ int seg_pre[SEL_MAXP + 1];
  __shared__ int64_t seg_row0[SEL_MAXP];
  __shared__ uint32_t cand_key[SEL_CAND];
  __shared__ uint32_t cand_e[SEL_CAND];
  __shared__ int64_t cand_row[SEL_CAND];
  __shared__ unsigned long long red[2][SEL_THR / 64];
  __shared__ int n_cand;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t q = blockIdx.x;
  const int k = p.k;
  const int np = DENSE ? 1 : p.nprobe;
  int red_slot = 0;
  auto block_reduce = [&](unsigned long long v, bool is_max) -> unsigned long long {
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(v, off);
      v = is_max ? (o > v ? o : v) : (o < v ? o : v);
    }
    if (lane == 0) red[red_slot][w] = v;
    __syncthreads();
    unsigned long long r = red[red_slot][0];
    for (int i = 1; i < SEL_THR / 64; ++i) { const unsigned long long o = red[red_slot][i]; r = is_max ? (o > r ? o : r) : (o < r ? o : r); }
    red_slot ^= 1;
    return r;
  };

  // ---- the query's segments
  if (tid < np) {
    const float* ptr = nullptr;
    int cnt = 0;
    int64_t r0 = 0;
    if (DENSE) {
      ptr = p.S + (size_t)q * p.dense_pitch;
      cnt = p.dense_rows;
    } else {
      const int4 pi = p.pinfo[(size_t)q * np + tid];     // (segment offset low, high, rows, first stored row); rows = 0: none
      if (pi.z > 0) {
        ptr = p.S + ((int64_t)(uint32_t)pi.x | ((int64_t)pi.y << 32));
        cnt = pi.z;
        r0 = (int64_t)(uint32_t)pi.w;
      }
    }
    seg_ptr[tid] = ptr; seg_cnt[tid] = cnt; seg_row0[tid] = r0;
  }
  if (tid == 0) n_cand = 0;
  __syncthreads();
  if (w == 0) {                                      // prefix sums of the <= 256 segment lengths: 4 per lane + a wave scan
    int c[4], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int j = lane * 4 + i; c[i] = j < np ? seg_cnt[j] : 0; sum += c[i]; }
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    int run = incl - sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int j = lane * 4 + i; if (j < np) seg_pre[j] = run; run += c[i]; }
    if (lane == 63) seg_pre[np] = incl;
  }
  __syncthreads();
  const int total = seg_pre[np];
  auto row_at = [&](int j, int t) -> int64_t { return DENSE ? (int64_t)t : p.row_ids[seg_row0[j] + t]; };
  auto emit = [&](int rank, uint32_t key, int64_t row) {
    if (p.out_scores) p.out_scores[q * k + rank] = s_ord_f32(key);
    p.out_ids[q * k + rank] = row + p.id_base;
    if (DENSE && p.count_out && p.row_weight[row] > 0) atomicAdd(&p.count_out[row], 1);
  };
  const int kk = total < k ? total : k;
  bool serial = total > SEL_CAP || total <= k;     // (fewer candidates than k: all of them, in order -- the serial path does that)

  if (!serial) {
    // ---- fast path: scores -> registers
    uint32_t kr[SEL_EPT];
    uint32_t tmax = 0u;
    {
      // all loads first (no use of a loaded value inside this loop: the <= 32 loads of a thread are in flight together)
      float raw[SEL_EPT];
      int j = 0;
#pragma unroll
      for (int i = 0; i < SEL_EPT; ++i) {
        const int e = tid + i * SEL_THR;
        raw[i] = 0.f;
        if (e < total) {
          if (!DENSE) while (seg_pre[j + 1] <= e) ++j;
          raw[i] = seg_ptr[j][e - seg_pre[j]];
        }
      }
#pragma unroll
      for (int i = 0; i < SEL_EPT; ++i) {
        const int e = tid + i * SEL_THR;
        const uint32_t key = e < total ? s_f32_ord(raw[i]) : 0u;   // 0: below every real key (s_f32_ord never returns 0 for a non-NaN score)
        kr[i] = key;
        tmax = key > tmax ? key : tmax;
      }
    }
    // k-th largest of the 256 thread maxima: the largest P with #(tmax >= P) >= k, two bits per step
    // (four bits per step -- 15 ballots, half the barriers -- was slower: the 64-bit unpacking of 15 counts outweighs them)
    uint32_t P = 0u;
    for (int bit = 30; bit >= 0; bit -= 2) {
      const uint32_t c1 = P | (1u << bit), c2 = P | (2u << bit), c3 = P | (3u << bit);
      const unsigned long long n1 = __popcll(__ballot(tmax >= c1)), n2 = __popcll(__ballot(tmax >= c2)), n3 = __popcll(__ballot(tmax >= c3));
      if (lane == 0) red[red_slot][w] = n1 | (n2 << 16) | (n3 << 32);
      __syncthreads();
      unsigned long long t = 0;
      for (int i = 0; i < SEL_THR / 64; ++i) t += red[red_slot][i];
      red_slot ^= 1;
      const int N1 = (int)(t & 0xFFFF), N2 = (int)((t >> 16) & 0xFFFF), N3 = (int)((t >> 32) & 0xFFFF);
      P = N3 >= k ? c3 : N2 >= k ? c2 : N1 >= k ? c1 : P;
    }
    // shortlist
    // (wave-aggregated append: one LDS atomic per wave and register slot that has any hit -- a lane-by-lane atomicAdd on the
    // one counter serialises ~k lanes)
#pragma unroll
    for (int i = 0; i < SEL_EPT; ++i) {
      const int e = tid + i * SEL_THR;
      const bool hit = e < total && kr[i] >= P;
      const unsigned long long hm = __ballot(hit);
      if (hm) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&n_cand, (int)__popcll(hm));
        base = __shfl(base, 0);
        const int c = base + (int)__popcll(hm & ((1ull << lane) - 1ull));
        if (hit && c < SEL_CAND) { cand_key[c] = kr[i]; cand_e[c] = (uint32_t)e; }
      }
    }
    __syncthreads();
    const int nc = n_cand;
    if (nc > SEL_CAND) serial = true;              // (workgroup-uniform)
    else {
      for (int c = tid; c < nc; c += SEL_THR) {
        const int e = (int)cand_e[c];
        int j = 0;
        if (!DENSE) {                              // segment of entry e: binary search over the prefix sums
          int lo = 0, hi = np;                     // seg_pre[lo] <= e < seg_pre[hi]
          while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_pre[mid] <= e) lo = mid; else hi = mid; }
          j = lo;
        }
        cand_row[c] = row_at(j, e - seg_pre[j]);
      }
      __syncthreads();
      for (int c = tid; c < nc; c += SEL_THR) {
        const uint32_t mk = cand_key[c];
        const int64_t mr = cand_row[c];
        int rank = 0;
        for (int i = 0; i < nc; ++i) {
          const uint32_t ok = cand_key[i];
          rank += (ok > mk || (ok == mk && cand_row[i] < mr)) ? 1 : 0;
        }
        if (rank < k) emit(rank, mk, mr);
      }
    }
  }
  if (serial) {
    // ---- serial path: kk rounds of arg-max in (score desc, row asc) order, strictly after the previous winner
    auto for_each = [&](auto&& fn) {
      if (DENSE) {
        for (int t = tid; t < total; t += SEL_THR) fn(0, t);
      } else {
        for (int j = w; j < np; j += SEL_THR / 64) {
          const int c = seg_cnt[j];
          for (int t = lane; t < c; t += 64) fn(j, t);
        }
      }
    };
    uint32_t last_key = 0u;
    int64_t last_row = -1;
    for (int it = 0; it < kk; ++it) {
      unsigned long long best = 0ull;                                   // key + 1 (0 = none)
      for_each([&](int j, int t) {
        const uint32_t key = s_f32_ord(seg_ptr[j][t]);
        bool ok = it == 0 || key < last_key;
        if (!ok && key == last_key) ok = row_at(j, t) > last_row;
        if (ok && (unsigned long long)key + 1ull > best) best = (unsigned long long)key + 1ull;
      });
      best = block_reduce(best, true);
      const uint32_t kmax = (uint32_t)(best - 1ull);
      unsigned long long br = ~0ull;                                    // row << 32 | position inside the workgroup's walk is not needed: rows are distinct
      for_each([&](int j, int t) {
        if (s_f32_ord(seg_ptr[j][t]) != kmax) return;
        const int64_t r = row_at(j, t);
        if (it == 0 || kmax < last_key || r > last_row) br = (unsigned long long)r < br ? (unsigned long long)r : br;
      });
      br = block_reduce(br, false);
      if (tid == 0) emit(it, kmax, (int64_t)br);
      last_key = kmax;
      last_row = (int64_t)br;
    }
  }
  for (int i = kk + tid; i < k; i += SEL_THR) {
    if (p.out_scores) p.out_scores[q * k + i] = -INFINITY;
    p.out_ids[q * k + i] = -1;
  }
}

// (two entry points: the single-segment form fits 64 registers and runs 8 waves per SIMD -- 88 -> 71 us for 10 000 probe
// selections --, the multi-segment form spills at that budget and is faster left alone)
__global__ __launch_bounds__(SEL_THR) __attribute__((amdgpu_waves_per_eu(8, 8))) void ivfs_select_dense_kernel(SelParams p) { ivfs_select_body<true>(p); }
__global__ __launch_bounds__(SEL_THR) void ivfs_select_lists_kernel(SelParams p) { ivfs_select_body<false>(p); }

// ------------------------------------------------------------------ descriptors of the dense (probe selection) case
__global__ void ivfs_dense_desc_kernel(int* __restrict__ desc, int n_qt, int n_chunks, int64_t nq, int n_rows, int rows_per_wg, int pitch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_qt * n_chunks) return;
  const int qt = i / n_chunks, c = i - qt * n_chunks;
  int* d = desc + (size_t)i * IVFS_DESC_WORDS;
  const int64_t q0 = (int64_t)qt * ST;
  const int64_t soff = q0 * pitch;
  d[0] = (int)(-1 - q0);
  d[1] = (int)(nq - q0 < ST ? nq - q0 : ST);
  d[2] = c * rows_per_wg;
  d[3] = n_rows - c * rows_per_wg < rows_per_wg ? n_rows - c * rows_per_wg : rows_per_wg;
  d[4] = c * rows_per_wg;
  d[5] = (int)(uint32_t)(soff & 0xFFFFFFFFll);
  d[6] = (int)(soff >> 32);
  d[7] = pitch;
}

int ivfs_scan(const uint16_t* corpus, const uint16_t* queries, int ld, int dtype, int64_t zero_row, const int* desc, int n_desc,
              const int* n_desc_dev, const int64_t* gq, float* S, hipStream_t stream, long long* dbg) {
  if (n_desc <= 0) return MRAG_OK;
  if (ld % SK) return fail(MRAG_ERR_INVALID, "ivfs_scan: ld %d is not a multiple of %d", ld, SK);
  static int cus = 0;
  if (!cus) { hipDeviceProp_t pr; int dev = 0; cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
  static const int wg_per_cu = [] { const char* e = getenv("MRAG_IVFS_WG_PER_CU"); return e ? atoi(e) : 2; }();
  const unsigned grid = wg_per_cu > 0 ? (unsigned)std::min<int64_t>((int64_t)(n_desc + 7) / 8 * 8, (int64_t)cus * wg_per_cu / 8 * 8)
                                      : (unsigned)((n_desc + 7) / 8 * 8);      // (0: one workgroup per descriptor)
  if (dtype == MRAG_F16) hipLaunchKernelGGL((ivfs_scan_kernel<MRAG_F16>), dim3(grid), dim3(STHR), 0, stream, corpus, queries, ld, zero_row, desc, n_desc, n_desc_dev, gq, S, dbg);
  else hipLaunchKernelGGL((ivfs_scan_kernel<MRAG_BF16>), dim3(grid), dim3(STHR), 0, stream, corpus, queries, ld, zero_row, desc, n_desc, n_desc_dev, gq, S, dbg);
  MRAG_HIP(hipGetLastError());
  return MRAG_OK;
}

int ivfs_select_lists(const float* S, const void* pinfo, int nprobe, int64_t nq, int k, const int64_t* row_ids,
                      int64_t id_base, float* out_scores, int64_t* out_ids, hipStream_t stream) {
  if (nq <= 0) return MRAG_OK;
  if (nprobe > SEL_MAXP || k > SEL_MAXK) return fail(MRAG_ERR_UNSUPPORTED, "ivfs_select: nprobe %d / k %d above %d / %d", nprobe, k, SEL_MAXP, SEL_MAXK);
  SelParams p{};
  p.S = S; p.pinfo = (const int4*)pinfo; p.row_ids = row_ids; p.nprobe = nprobe; p.k = k;
  p.id_base = id_base; p.out_scores = out_scores; p.out_ids = out_ids;
  hipLaunchKernelGGL(ivfs_select_lists_kernel, dim3((unsigned)nq), dim3(SEL_THR), 0, stream, p);
  MRAG_HIP(hipGetLastError());
  return MRAG_OK;
}

// top-k rows of a dense [nq] x [n_rows] problem (probe selection): descriptors, scan, select.  `desc` must hold
// ivfs_dense_n_desc(nq, n_rows) descriptors, S nq_round128 x ivfs_pitch(n_rows) floats.
int ivfs_dense_n_desc(int64_t nq, int n_rows) {
  return (int)((nq + ST - 1) / ST) * ((n_rows + IVFS_DENSE_ROWS - 1) / IVFS_DENSE_ROWS);
}

int ivfs_dense_topk(const uint16_t* corpus, int n_rows, const uint16_t* queries, int64_t nq, int ld, int dtype, int k, int* desc,
                    float* S, float* out_scores, int64_t* out_ids, hipStream_t stream, int* count_out, const int* row_weight) {
  if (nq <= 0) return MRAG_OK;
  if (k > SEL_MAXK) return fail(MRAG_ERR_UNSUPPORTED, "ivfs_dense_topk: k %d above %d", k, SEL_MAXK);
  const int n_qt = (int)((nq + ST - 1) / ST), n_chunks = (n_rows + IVFS_DENSE_ROWS - 1) / IVFS_DENSE_ROWS;
  const int pitch = (int)ivfs_pitch(n_rows);
  const int nd = n_qt * n_chunks;
  hipLaunchKernelGGL(ivfs_dense_desc_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, stream, desc, n_qt, n_chunks, nq, n_rows,
                     IVFS_DENSE_ROWS, pitch);
  MRAG_HIP(hipGetLastError());
  MRAG_TRY(ivfs_scan(corpus, queries, ld, dtype, 0, desc, nd, nullptr, nullptr, S, stream));
  SelParams p{};
  p.S = S; p.nprobe = 1; p.k = k; p.dense_rows = n_rows; p.dense_pitch = pitch; p.out_scores = out_scores; p.out_ids = out_ids;
  p.count_out = count_out; p.row_weight = row_weight;
  hipLaunchKernelGGL(ivfs_select_dense_kernel, dim3((unsigned)nq), dim3(SEL_THR), 0, stream, p);
  MRAG_HIP(hipGetLastError());
  return MRAG_OK;
}

}  // namespace mrag
