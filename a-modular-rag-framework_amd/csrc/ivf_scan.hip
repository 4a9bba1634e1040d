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
  int64_t qid[4];                                            // the four query-row ids this lane gathers: all four loads in flight together
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (4 * w + i) * 8 + (lane >> 3);
    qid[i] = (gq_base >= 0 && !IVFS_DBG(4) && r < nq_local) ? gq[(size_t)gq_base + r] : -1;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (4 * w + i) * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((r >> 1) & 7);
    a_src[i] = corpus + (size_t)(row0 + r) * ld + kc * 8;
    int64_t qrow;
    if (gq_base < 0) qrow = (int64_t)(-1 - gq_base) + r;                 // identity: queries first .. first + 127 (the buffer is padded)
    else if (IVFS_DBG(4)) qrow = r;
    else { qrow = qid[i]; if (qrow < 0) qrow = zero_row; }
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

#ifdef MRAG_IVFS_SCAN256
// ------------------------------------------------------------------ EXPERIMENT (round 3, not built by default): 256-row tiles, ring of three stages
// make EXTRA=-DMRAG_IVFS_SCAN256 builds it; MRAG_IVFS_SCAN=256 then selects it at run time.  Hypothesis: the kernel above is
// latency-bound per K step (1.85 us per step whatever it carries), so a 150-row list pays 24 of them (its second 128-row tile
// holds 24 rows) with one 32-KiB stage in flight per workgroup.  This form:
//   * one workgroup of 8 waves per CU, tile = 256 corpus rows x 128 queries x K 64: a 129..256-row list is ONE tile
//     (12 K steps at d = 768 instead of 24), the gathered query rows are fetched once per 256 rows instead of per 128;
//   * a ring of three 48-KiB stages behind COUNTED vmcnt waits: two stages (<= 96 KiB) in flight per CU at any time;
//   * the stream of K steps runs ACROSS descriptors: while the last steps of a list are multiplied the first stages of the
//     workgroup's next list are already on their way (a 12-step list would otherwise pay a pipeline fill per list).  The
//     next lists' query-row ids (gq) ride the same queue: one 1-KiB LDS-DMA per descriptor, three descriptors ahead (a
//     descriptor can be a single K step: its ids must have landed AND passed a barrier before the issue side reads them);
//   * score stores stay in flight across K steps: the waits count them (they are younger than the loads they follow).
// Same MFMA instruction, K order and (score, row) semantics as above: scores are bit-identical (the IVF tests pass on it).
// MEASURED (same box, rocprofv3 kernel trace, tools/ivf_split.sh): C5 share 625 k rows: list scan 340 us (above) vs 384-394 us
// (this), probe scan over the 4 096 centroids 110 vs 128 us; whole corpus 5 M rows: 2.47 vs 2.34-2.44 ms (2.37 with
// non-temporal corpus loads, which cost the probe scan 20 us).  Fewer K steps, fewer bytes and twice the bytes in flight buy
// nothing: the stream is bound by the per-CU request path at the rate the guide gives for HBM-sourced LDS-DMA (23-24 GB/s per
// CU, MI355X_MICROARCH.md "Indexed rows"), not by round trips, and one 8-wave workgroup per CU has no second workgroup to
// cover its per-descriptor seams.  Kept as a record; the shipped scan is the kernel above.
constexpr int R_TR = 256;                           // corpus rows per tile
constexpr int R_THR = 512;                          // 8 waves: wave (wm = w >> 1, wn = w & 1) owns rows wm*64.., queries wn*64..
constexpr int R_A_BYTES = R_TR * SK * 2;            // 32 KiB
constexpr int R_B_BYTES = ST * SK * 2;              // 16 KiB
constexpr int R_STAGE = R_A_BYTES + R_B_BYTES;      // 48 KiB
constexpr int R_NS = 3;
#ifndef MRAG_IVFS_A_AUX
#define MRAG_IVFS_A_AUX 0
#endif
constexpr int R_A_AUX = MRAG_IVFS_A_AUX;            // cache policy of the corpus-row loads (2 = non-temporal)
constexpr int R_GQ = 4;                             // ring of query-row id blocks: the descriptor being staged and the three after it
constexpr int R_GQ_OFF = R_NS * R_STAGE;            // int64 [R_GQ][128]
constexpr int R_DS_OFF = R_GQ_OFF + R_GQ * ST * 8;  // int [R_GQ][IVFS_DESC_WORDS]: the descriptors' words, passed from the issue side to the compute side
constexpr int R_LDS = R_DS_OFF + R_GQ * IVFS_DESC_WORDS * 4;   // 151 680 B

__device__ __forceinline__ void r_wait_vm(int n) {   // s_waitcnt vmcnt(<= n): the immediate must be a constant
  if (n >= 38) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
  else if (n >= 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
  else if (n >= 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
  else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

struct RDesc {                 // one descriptor's scalars (common.h: IVFS_DESC_WORDS)
  int gq_base, nq, n_rows, r_off, pitch, n_tiles;
  int64_t row0, soff;
};

template <int DT>
__global__ __launch_bounds__(R_THR) void ivfs_scan256_kernel(const uint16_t* __restrict__ corpus, const uint16_t* __restrict__ queries,
                                                             int ld, int64_t zero_row, const int* __restrict__ desc, int n_desc,
                                                             const int* __restrict__ n_desc_dev, const int64_t* __restrict__ gq,
                                                             float* __restrict__ S) {
  typedef typename SMfma<DT>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char rsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int n_real = n_desc_dev ? *n_desc_dev : n_desc;
  // blocks go to XCDs round-robin: XCD x owns the contiguous run [x * per, (x + 1) * per) of descriptors (a list's
  // workgroups are neighbours: one L2 fetches it); this workgroup takes every (gridDim / 8)-th of them
  const int per = (n_real + 7) >> 3;
  const int it0 = (int)(blockIdx.x >> 3), it_step = (int)(gridDim.x >> 3);
  const int xbase = (int)(blockIdx.x & 7) * per;
  auto lin_of = [&](int ord) -> int {                  // ord-th descriptor of this workgroup, or -1
    const int it = it0 + ord * it_step;
    if (it >= per) return -1;
    const int lin = xbase + it;
    return lin < n_real ? lin : -1;
  };
  int* ds_lds = (int*)(rsm + R_DS_OFF);
  auto load_desc = [&](const int* d) -> RDesc {
    RDesc r;
    r.gq_base = d[0]; r.nq = d[1]; r.n_rows = d[3]; r.r_off = d[4]; r.pitch = d[7];
    r.row0 = (int64_t)(uint32_t)d[2];
    r.soff = (int64_t)(uint32_t)d[5] | ((int64_t)d[6] << 32);
    r.n_tiles = (r.n_rows + R_TR - 1) / R_TR;
    return r;
  };
  const int nk = ld / SK;
  const uint32_t row_b = (uint32_t)ld * 2u;
  int64_t* gq_lds = (int64_t*)(rsm + R_GQ_OFF);

  // per-lane parts of the LDS-DMA sources.  A: wave w fills the 1-KiB chunks 4w..4w+3 (8 rows x 128 B) of the corpus tile,
  // B: chunks 2w, 2w+1 of the query block; the 16-byte piece a lane fetches is XOR-swizzled by its row (conflict-free reads)
  uint32_t a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (4 * w + i) * 8 + (lane >> 3);
    a_off[i] = (uint32_t)r * row_b + (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
  }
  int b_row[2];
  uint32_t b_kc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    b_row[i] = (2 * w + i) * 8 + (lane >> 3);
    b_kc[i] = (uint32_t)(((lane & 7) ^ ((b_row[i] >> 1) & 7)) * 16);
  }
  // one LDS-DMA (wave 0, 1 KiB) brings the 128 query-row ids of descriptor `lin` into gq slot `slot`; returns ops issued
  auto issue_gq = [&](int lin, int slot) -> int {
    if (lin < 0 || w != 0) return 0;
    const int gb = desc[(size_t)lin * IVFS_DESC_WORDS];
    if (gb < 0) return 0;                                             // identity map: no ids to fetch
    __builtin_amdgcn_global_load_lds((s_glb_vptr)((const char*)(gq + gb) + lane * 16), (s_lds_vptr)(rsm + R_GQ_OFF + slot * (ST * 8)), 16, 0, 0);
    return 1;
  };

  // ---- issue side: cursor over (descriptor, tile, k step) of the NEXT stage to issue ------------------------------
  int i_ord = 0, i_lin = lin_of(0), i_tile = 0, i_kk = 0, i_buf = 0;
  RDesc id{};
  const char* i_bptr[2] = {nullptr, nullptr};                        // this lane's two query-row sources of the issue descriptor
  auto issue_enter = [&]() {                                          // the issue cursor has just moved to descriptor i_ord (i_lin >= 0)
    const int* dg = desc + (size_t)i_lin * IVFS_DESC_WORDS;
    id = load_desc(dg);
    // the compute side reaches this descriptor at least one barrier later: it takes the words from LDS instead of waiting
    // out a global (scalar) load with all eight waves idle
    if (tid < IVFS_DESC_WORDS) ds_lds[(i_ord % R_GQ) * IVFS_DESC_WORDS + tid] = dg[tid];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t qrow;
      if (id.gq_base < 0) qrow = (int64_t)(-1 - id.gq_base) + b_row[i];
      else { qrow = b_row[i] < id.nq ? gq_lds[(i_ord % R_GQ) * ST + b_row[i]] : -1; if (qrow < 0) qrow = zero_row; }
      i_bptr[i] = (const char*)queries + (size_t)qrow * row_b + b_kc[i];
    }
  };
  // issue the next stage of the stream (if any); returns this wave's number of VMEM ops
  auto issue_next = [&]() -> int {
    if (i_lin < 0) return 0;
    int ops = 0;
    if (i_tile == 0 && i_kk == 0) ops += issue_gq(lin_of(i_ord + 3), (i_ord + 3) % R_GQ);   // ids of the third descriptor from here
    char* la = rsm + i_buf * R_STAGE;
    const char* abase = (const char*)corpus + (size_t)(id.row0 + (int64_t)i_tile * R_TR) * row_b + (size_t)i_kk * (SK * 2);
    const int rows_left = id.n_rows - i_tile * R_TR;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if ((4 * w + i) * 8 < rows_left) {
        __builtin_amdgcn_global_load_lds((s_glb_vptr)(abase + a_off[i]), (s_lds_vptr)(la + (4 * w + i) * 1024), 16, 0, R_A_AUX);
        ++ops;
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if ((2 * w + i) * 8 < id.nq) {
        __builtin_amdgcn_global_load_lds((s_glb_vptr)(i_bptr[i] + (size_t)i_kk * (SK * 2)), (s_lds_vptr)(la + R_A_BYTES + (2 * w + i) * 1024), 16, 0, 0);
        ++ops;
      }
    if (++i_buf == R_NS) i_buf = 0;
    if (++i_kk == nk) {
      i_kk = 0;
      if (++i_tile == id.n_tiles) {
        i_tile = 0;
        ++i_ord;
        i_lin = lin_of(i_ord);
        if (i_lin >= 0) issue_enter();
      }
    }
    return ops;
  };

  // ---- prologue: query-row ids of the first three descriptors, then two stages in flight -----------------------------
  // (ids of descriptor X are issued with the first stage of X - 3 and read when the issue side enters X, i.e. while it
  // issues the last stage of X - 1: at least three stages, hence one wait + barrier of the issuing wave, later.  The
  // barriers between the prologue's issue calls keep a slot's readers ahead of its next writer.)
  if (i_lin < 0) return;
  issue_gq(i_lin, 0);
  issue_gq(lin_of(1), 1);
  issue_gq(lin_of(2), 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  issue_enter();
  const RDesc first_desc = id;      // (the compute side starts on descriptor 0; the prologue's issues below may already move `id` on)
  __syncthreads();
  // ops of the stages issued but not yet waited for and the stores issued behind each of them
  issue_next();                     // stage s     (the one the first step needs; its op count is never waited on alone)
  __syncthreads();
  int ops2 = issue_next();          // stage s + 1
  int st1 = 0, st2 = 0;             // score stores issued after stage s / after stage s + 1 (younger than them)

  // ---- compute side ------------------------------------------------------------------------------------------------
  int c_ord = 0, c_tile = 0, c_kk = 0, c_buf = 0;
  RDesc cd = first_desc;
  const int frow = lane & 15, fsw = frow >> 1;
  const int a_rd = (wm * 64 + frow) * 128, b_rd = R_A_BYTES + (wn * 64 + frow) * 128;
  const int ph0 = ((lane >> 4) ^ fsw) * 16;
  s_f32x4 acc[4][4];
  while (true) {
    // stage s has landed once everything older than {stores after s, stage s + 1, stores after s + 1} is done
    r_wait_vm(st1 + ops2 + st2);
    __builtin_amdgcn_s_barrier();          // everyone's pieces of stage s are in LDS; everyone is done with stage s - 1's buffer
    asm volatile("" ::: "memory");
    const int ops3 = issue_next();         // stage s + 2 goes into the buffer stage s - 1 occupied
    int st3 = 0;
    const char* sb = rsm + c_buf * R_STAGE;
    if (c_kk == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (s_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // this wave's 64 x 64 piece: only the 16-row / 16-query fragments that hold real rows / queries (wave-uniform tests)
    const int ni = min(4, (cd.n_rows - c_tile * R_TR - wm * 64 + 15) >> 4);
    const int nj = min(4, (cd.nq - wn * 64 + 15) >> 4);
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
            if (i < ni && j < nj) acc[i][j] = SMfma<DT>::run(af[i], bf[j], acc[i][j]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of the buffer are done before it reaches the next barrier
    if (++c_buf == R_NS) c_buf = 0;
    bool done = false;
    if (++c_kk == nk) {
      // scores of this tile: lane holds rows r0..r0+3 (consecutive) of query slot q per accumulator
      const int rl = c_tile * R_TR + wm * 64 + (lane >> 4) * 4;
      const int rows4 = (cd.n_rows + 3) & ~3;
      float* Sb = S + cd.soff;
      if (ni > 0 && nj > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j >= nj) continue;                                   // (wave-uniform: no store instruction is issued)
          const int q = wn * 64 + j * 16 + (lane & 15);
          float* sq = Sb + (size_t)q * cd.pitch + cd.r_off;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i >= ni) continue;
            const int r0 = rl + i * 16;
            if (q < cd.nq && r0 < rows4) *(s_f32x4*)(sq + r0) = acc[i][j];
            ++st3;                                                 // one VMEM op per (i, j) the wave executes
          }
        }
      }
      c_kk = 0;
      if (++c_tile == cd.n_tiles) {
        c_tile = 0;
        ++c_ord;
        const int nl = lin_of(c_ord);
        if (nl < 0) done = true; else cd = load_desc(ds_lds + (c_ord % R_GQ) * IVFS_DESC_WORDS);
      }
    }
    if (done) break;
    ops2 = ops3; st1 = st2; st2 = st3;
  }
}

#endif  // MRAG_IVFS_SCAN256

// ------------------------------------------------------------------ per-query selection
constexpr int SEL_THR = 256;
constexpr int SEL_EPT = 32;                        // candidate scores a thread keeps in registers
constexpr int SEL_CAP = SEL_THR * SEL_EPT;         // 8192 per query on the fast path; beyond it the serial (exact) path
constexpr int SEL_CAND = 512;                      // shortlisted entries (>= the k-th largest thread maximum)
constexpr int SEL_MAXP = 256;                      // probes per query
constexpr int SEL_MAXK = 256;
#ifndef MRAG_SEL_WAVE_K
#define MRAG_SEL_WAVE_K 16
#endif
constexpr int SEL_WAVE_K = MRAG_SEL_WAVE_K;        // k up to which the register path bounds per wave instead of per workgroup (0: never)

// k-th largest of the 64 lane values of a wave (0 when fewer than k lanes hold a non-zero value): the largest P with
// #(v >= P) >= k, two bits per step, ballots only
__device__ __forceinline__ uint32_t wave_kth_of_lane_maxima(uint32_t v, int k) {
  uint32_t P = 0u;
  for (int bit = 30; bit >= 0; bit -= 2) {
    const uint32_t c1 = P | (1u << bit), c2 = P | (2u << bit), c3 = P | (3u << bit);
    const int n1 = __popcll(__ballot(v >= c1)), n2 = __popcll(__ballot(v >= c2)), n3 = __popcll(__ballot(v >= c3));
    P = n3 >= k ? c3 : n2 >= k ? c2 : n1 >= k ? c1 : P;
  }
  return P;
}

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
  bool serial = total <= k;                        // (fewer candidates than k: all of them, in order -- the serial path does that)
  const bool streamed = !serial && total > SEL_CAP;   // more candidates than the register path holds: two streaming passes

  // k-th largest of the 256 thread maxima: the largest P with #(tmax >= P) >= k, two bits per step
  // (four bits per step -- 15 ballots, half the barriers -- was slower: the 64-bit unpacking of 15 counts outweighs them)
  auto kth_of_thread_maxima = [&](uint32_t tmax) -> uint32_t {
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
    return P;
  };
  // wave-aggregated shortlist append: one LDS atomic per wave and call that has any hit (a lane-by-lane atomicAdd on the one
  // counter serialises ~k lanes); call with the whole wave converged
  auto shortlist_append = [&](bool hit, uint32_t key, int e) {
    const unsigned long long hm = __ballot(hit);
    if (hm) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&n_cand, (int)__popcll(hm));
      base = __shfl(base, 0);
      const int c = base + (int)__popcll(hm & ((1ull << lane) - 1ull));
      if (hit && c < SEL_CAND) { cand_key[c] = key; cand_e[c] = (uint32_t)e; }
    }
  };
  // the shortlisted entries get their original rows, are ranked by counting on (score desc, row asc); ranks < k are the answer
  auto rank_shortlist = [&]() -> bool {
    __syncthreads();
    const int nc = n_cand;
    if (nc > SEL_CAND) return false;               // (workgroup-uniform) degenerate ties: the serial path
    for (int c = tid; c < nc; c += SEL_THR) {
      const int e = (int)cand_e[c];
      int j = 0;
      if (!DENSE) {                                // segment of entry e: binary search over the prefix sums
        int lo = 0, hi = np;                       // seg_pre[lo] <= e < seg_pre[hi]
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
    return true;
  };

  if (!serial && !streamed) {
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
    // k <= 16: every WAVE bounds its own quarter of the entries -- the k-th largest of its 64 lane maxima, found with ballots
    // only (no LDS, no workgroup barrier: the block-wide bisection costs 16 barriers per query).  A wave's top k all reach its
    // bound, so the union of the four shortlists (~ 4k(1 + k/64) entries) holds the query's top k; ranking as before.
    // Measured (C5 share, k = 10): 112 -> 108 us; at k = 32 (probe selection) the fourfold shortlist costs more in the
    // ranking than the barriers saved (70 -> 131 us), hence the limit.
    const uint32_t P = k <= SEL_WAVE_K ? wave_kth_of_lane_maxima(tmax, k) : kth_of_thread_maxima(tmax);
#pragma unroll
    for (int i = 0; i < SEL_EPT; ++i) {
      const int e = tid + i * SEL_THR;
      shortlist_append(e < total && kr[i] >= P, kr[i], e);
    }
    if (!rank_shortlist()) serial = true;
  }
  if (streamed) {
    // ---- more than 8192 candidates (long lists: C5's whole 5 M rows on one GPU give ~39 000 per query): the same
    // selection with the scores streamed twice instead of held in registers.  Pass 1: every thread's maximum over its
    // entries (segment by segment, entry t of a segment on thread t mod 256: coalesced, 8 loads in flight) -> the k-th
    // largest thread maximum P, a lower bound of the k-th best score.  Pass 2: the entries >= P are the shortlist.
    // (pass 1 only needs SOME k entries' lower bound: it reads the leading segments -- the best-ranked probes, where most of
    // the answer sits -- until it has seen `want` entries: enough that the entries >= P over ALL segments stay a shortlist of
    // ~ k (total / want)(1 + k / 256) <= ~384 entries; 5 M rows, k = 10: 8 192 of ~39 000 entries, selection 1.24 -> 0.93 ms)
    const long long need = ((long long)total * k * (256 + k) / 256 + 383) / 384;
    const int want = (int)(need > total ? total : need < SEL_CAP ? SEL_CAP : need);
    uint32_t tmax = 0u;
    int seen = 0;
    for (int j = 0; j < np && seen < want; ++j) {
      const float* sp = seg_ptr[j];
      const int cnt = seg_cnt[j];
      seen += cnt;
#pragma unroll 8
      for (int t = tid; t < cnt; t += SEL_THR) {
        const uint32_t key = s_f32_ord(sp[t]);
        tmax = key > tmax ? key : tmax;
      }
    }
    const uint32_t P = kth_of_thread_maxima(tmax);
    for (int j = 0; j < np; ++j) {
      const float* sp = seg_ptr[j];
      const int cnt = seg_cnt[j], pre = seg_pre[j];
      for (int t0 = 0; t0 < cnt; t0 += 8 * SEL_THR) {      // (uniform trip count: the append's ballot needs the wave converged)
        float raw[8];                                      // eight loads in flight per thread, then the (rare) appends
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int t = t0 + u * SEL_THR + tid; raw[u] = t < cnt ? sp[t] : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = t0 + u * SEL_THR + tid;
          const uint32_t key = t < cnt ? s_f32_ord(raw[u]) : 0u;
          shortlist_append(t < cnt && key >= P, key, pre + t);
        }
      }
    }
    if (!rank_shortlist()) serial = true;
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
#ifdef MRAG_IVFS_SCAN256
  // experiment build only: MRAG_IVFS_SCAN=256 selects the 256-row / ring-of-three form (see its header for the numbers)
  static const bool scan256 = [] { const char* e = getenv("MRAG_IVFS_SCAN"); return e && atoi(e) == 256; }();
  if (scan256 && !dbg) {
    typedef void (*Fn)(const uint16_t*, const uint16_t*, int, int64_t, const int*, int, const int*, const int64_t*, float*);
    const Fn fn = dtype == MRAG_F16 ? (Fn)ivfs_scan256_kernel<MRAG_F16> : (Fn)ivfs_scan256_kernel<MRAG_BF16>;
    static bool attr_done[2] = {false, false};
    if (!attr_done[dtype == MRAG_F16]) { MRAG_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS)); attr_done[dtype == MRAG_F16] = true; }
    const unsigned grid256 = (unsigned)std::min<int64_t>((int64_t)(n_desc + 7) / 8 * 8, (int64_t)cus / 8 * 8);   // one workgroup per CU
    hipLaunchKernelGGL(fn, dim3(grid256), dim3(R_THR), R_LDS, stream, corpus, queries, ld, zero_row, desc, n_desc, n_desc_dev, gq, S);
    MRAG_HIP(hipGetLastError());
    return MRAG_OK;
  }
#endif
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
