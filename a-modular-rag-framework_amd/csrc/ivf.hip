// IVF-flat over the same fused similarity + top-k kernels (BASELINE.json config 5; the reference
// has no IVF index -- PARITY UNPINNED by the reference, checked against this library's own brute
// force and the oracle's ivf_search on identical centroids / assignments).
//
//   train    spherical k-means on the device: assignment = the dense kernel with the CENTROIDS as
//            corpus and the rows as queries (k = 1, ties to the lowest list id), update = float
//            atomics into [nlist][dim] sums, re-normalised and rounded to the storage dtype.
//   add      rows are normalised/rounded once, kept in arrival order, assigned as above.
//   search   1) probe lists: dense kernel over the centroids, k = nprobe
//            2) rows regrouped by list (stable by original row, each list padded to 256 rows);
//               every list becomes workgroups of <= 256 of the queries that probe it: the list's
//               rows are streamed from HBM ONCE for all of them (HBM-bound: bytes = list rows x
//               ld x 2 per workgroup), scored on MFMA and filtered by the same in-kernel top-k
//            3) per query the nprobe candidate sets are merged by (score desc, original row asc)
//   The (list -> queries) regrouping of step 2 runs on the device (count / plan / scatter kernels below);
//   only the workgroup count comes back to the host.
#include "common.h"

#include <algorithm>
#include <numeric>

namespace mrag {

struct IvfIndex : Object {
  int dim = 0, ld = 0, nlist = 0, metric = 0, dtype = MRAG_F16;
  bool has_centroids = false;
  int64_t n = 0, cap = 0, id_base = 0;
  uint16_t* cen = nullptr;        // [round256(nlist)][ld]
  uint16_t* raw = nullptr;        // [cap][ld] rows in arrival order
  int32_t* assign = nullptr;      // [cap]
  // regrouped storage (rebuilt lazily)
  bool dirty = true;
  uint16_t* sorted = nullptr;     // [n_sorted][ld]
  int64_t* row_ids = nullptr;     // [n_sorted] original row or -1
  int64_t n_sorted = 0;
  std::vector<int> list_tile_lo, list_count;
  int max_list_rows = 0;
  DevBuf qbuf, qg, lists, counts, stage_in, tmp_sc, tmp_id, out_sc, out_id, desc, ploc, gq, perm, sums, cnts;
  DevBuf d_list_count, d_list_tile_lo, plan;   // device copies of the list layout; plan = lcount | wg_first | cursor | n_wg
  DevBuf scores, sdesc;                        // "score segments + select" regime (ivf_scan.hip): fp32 segments, dense descriptors
  bool last_scores_path = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // search start | (unused) | (unused) | search end
  std::vector<hipEvent_t> cev;    // per chunk of the last search: list scan begin, list scan end
  int n_chunks = 0;               // chunks of the last search (mrag_ivf_last_timing sums their scan brackets)
  DevBuf stats;                   // device accumulators of the last search: int64 rows streamed by the list scans, int64 workgroups
  bool timed = false, end_recorded = false;
  ~IvfIndex() override {
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : cev) if (e) (void)hipEventDestroy(e);
    for (void* p : {(void*)cen, (void*)raw, (void*)assign, (void*)sorted, (void*)row_ids}) if (p) (void)hipFree(p);
    for (DevBuf* b : {&qbuf, &qg, &lists, &counts, &stage_in, &tmp_sc, &tmp_id, &out_sc, &out_id, &desc, &ploc, &gq, &perm, &sums, &cnts, &d_list_count, &d_list_tile_lo, &plan, &scores, &sdesc, &stats}) b->release();
  }
};

static size_t esize(int dt) { return dt == MRAG_F32 ? 4 : dt == MRAG_F64 ? 8 : (dt == MRAG_F16 || dt == MRAG_BF16) ? 2 : 0; }

__global__ void gather_rows_kernel(const uint16_t* __restrict__ src, const int64_t* __restrict__ idx, int64_t n, int ld,
                                   uint16_t* __restrict__ dst, int zero_fill) {
  // one wave per destination row; idx < 0 -> zero row
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int64_t s = idx[r];
  if (s < 0 && !zero_fill) return;   // search-time query slabs: rows past a group's queries are never read
  const uint4* sp = (const uint4*)(src + (size_t)(s < 0 ? 0 : s) * ld);
  uint4* dp = (uint4*)(dst + (size_t)r * ld);
  for (int i = lane; i < ld / 8; i += 64) dp[i] = s < 0 ? make_uint4(0, 0, 0, 0) : sp[i];
}

// ---- search-time regrouping on the device: list -> the (query, probe) pairs that hit it -------------
// probes [npairs] = list id per (query, probe slot) (< 0: none).  Pass 1 counts the pairs per list;
// pass 2 (one block) turns the counts into "first workgroup of the list" (256 queries per workgroup)
// and writes the workgroup descriptors; pass 3 hands every pair its (workgroup, slot) with an atomic
// cursor per list.  The slot order inside a list depends on the atomics' order, the results do not:
// every (query, row) score is computed the same way whichever slot the query sits in.
__global__ void ivf_count_kernel(const int64_t* __restrict__ probes, int64_t npairs, const int* __restrict__ list_count,
                                 int* __restrict__ lcount) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npairs) return;
  const int64_t l = probes[i];
  if (l >= 0 && list_count[l] > 0) atomicAdd(&lcount[l], 1);
}

__global__ __launch_bounds__(1024) void ivf_plan_kernel(const int* __restrict__ lcount, const int* __restrict__ list_count,
                                                        const int* __restrict__ list_tile_lo, int nlist,
                                                        int* __restrict__ wg_first, int* __restrict__ cursor,
                                                        int* __restrict__ desc, int desc_cap, int* __restrict__ n_wg_out,
                                                        unsigned long long* __restrict__ stats) {
  __shared__ int part[1024];
  __shared__ int carry;
  __shared__ unsigned long long rows_acc;
  const int tid = threadIdx.x;
  if (tid == 0) { carry = 0; rows_acc = 0ull; }
  __syncthreads();
  for (int base = 0; base < nlist; base += 1024) {
    const int l = base + tid;
    const int cnt = l < nlist ? lcount[l] : 0;
    const int nt = (cnt + 255) >> 8;
    if (nt && l < nlist) atomicAdd(&rows_acc, (unsigned long long)nt * (unsigned long long)list_count[l]);
    part[tid] = nt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {           // inclusive scan (Hillis-Steele)
      const int v = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    const int first = carry + part[tid] - nt;
    if (l < nlist) {
      wg_first[l] = first;
      cursor[l] = 0;
      const int tlo = list_tile_lo[l], lc = list_count[l];
      for (int c = 0; c < nt; ++c) {
        const int wgi = first + c;
        if (wgi < desc_cap) {
          int* d = desc + (size_t)wgi * 8;
          d[0] = wgi * 256;
          d[1] = min(256, cnt - c * 256);
          d[2] = tlo;
          d[3] = tlo + ((lc + 255) >> 8);
          d[4] = tlo * 256 + lc;
          d[5] = 0; d[6] = 0; d[7] = 0;
        }
      }
    }
    __syncthreads();
    if (tid == 1023) carry += part[1023];
    __syncthreads();
  }
  if (tid == 0) {
    *n_wg_out = carry;
    if (stats) { atomicAdd(&stats[0], rows_acc); atomicAdd(&stats[1], (unsigned long long)carry); }
  }
}

// (score-segment regime, pinfo != nullptr: the thread that draws slot 0 of a query group also writes the group's scan
// descriptors -- one per IVFS_CHUNK_ROWS rows of the list --, so the single-workgroup plan kernel only runs the prefix sums)
__global__ void ivf_scatter_kernel(const int64_t* __restrict__ probes, int64_t npairs, int nprobe,
                                   const int* __restrict__ list_count, const int* __restrict__ wg_first,
                                   int* __restrict__ cursor, int64_t* __restrict__ gq, int2* __restrict__ ploc, int qshift,
                                   const long long* __restrict__ foff, const int* __restrict__ list_tile_lo, int4* __restrict__ pinfo,
                                   const int* __restrict__ lcount, const int* __restrict__ dfirst, int* __restrict__ desc, int desc_cap) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npairs) return;
  const int64_t l = probes[i];
  if (l < 0 || list_count[l] == 0) {
    if (pinfo) pinfo[i] = make_int4(0, 0, 0, 0); else ploc[i] = make_int2(-1, 0);
    return;
  }
  const int pos = atomicAdd(&cursor[l], 1);
  const int wgi = wg_first[l] + (pos >> qshift), slot = pos & ((1 << qshift) - 1);   // 2^qshift queries per workgroup
  gq[((size_t)wgi << qshift) + slot] = i / nprobe;
  if (pinfo) {
    // score-segment regime: everything the per-query selection needs about this pair, in one 16-byte record -- its score
    // segment (its group's block + slot x pitch), the rows and the first stored row of the whole LIST (the list's rows may be
    // cut over several scan descriptors; every one of them writes its rows of this segment)
    const int lc = list_count[l], grp = pos >> qshift, tlo = list_tile_lo[l];
    const long long pitch = (lc + IVFS_PITCH_ALIGN - 1) / IVFS_PITCH_ALIGN * IVFS_PITCH_ALIGN;
    const long long gso = foff[l] + (long long)grp * IVFS_QUERIES * pitch;      // the group's score block
    const long long so = gso + (long long)slot * pitch;
    pinfo[i] = make_int4((int)(unsigned)(so & 0xFFFFFFFFll), (int)(so >> 32), lc, tlo * 256);
    if (slot == 0) {
      const int cnt = lcount[l], nch = (lc + IVFS_CHUNK_ROWS - 1) / IVFS_CHUNK_ROWS;
      for (int r = 0; r < nch; ++r) {
        const int di = dfirst[l] + grp * nch + r;
        if (di < desc_cap) {
          int4* d = (int4*)(desc + (size_t)di * IVFS_DESC_WORDS);
          d[0] = make_int4(wgi * IVFS_QUERIES, min(IVFS_QUERIES, cnt - grp * IVFS_QUERIES), tlo * 256 + r * IVFS_CHUNK_ROWS,
                           min(IVFS_CHUNK_ROWS, lc - r * IVFS_CHUNK_ROWS));
          d[1] = make_int4(r * IVFS_CHUNK_ROWS, (int)(unsigned)(gso & 0xFFFFFFFFll), (int)(gso >> 32), (int)pitch);
        }
      }
    }
  } else {
    ploc[i] = make_int2(wgi, slot);
  }
}

// Plan of the "score segments + select" regime (ivf_scan.hip).  A list probed by cnt queries becomes ceil(cnt / 128) query
// GROUPS (one block of scores S[query slot of the group][pitch = list rows rounded to 32] each, one block of 128 query-row
// ids in gq each) x ceil(rows / IVFS_CHUNK_ROWS) row CHUNKS = that many scan descriptors: a long list (skewed data: thousands
// of rows where the median list has a hundred) is scanned by several workgroups instead of being one workgroup's tail.
// Measured (round 3, one box, whole search): chunk = whole list / 1024 / 512 / 256 / 128 rows -- C5 share (152-row lists) 0.743 /
// - / 0.743 / 0.713 / 0.700 ms; 5 M rows (1 220-row lists) 4.02 / 3.81 / 3.64 / 3.51 / 3.56 ms; Zipf-sized lists (0 ... 5 330 rows),
// nprobe 1 / 32: 0.98 / 1.38 -> 0.62 / 1.13 -> 0.61 / 1.05 -> 0.52 / 0.94 -> 0.50 / 0.94 ms.
// out[0] = descriptors, out[1..2] = score floats; per list: wg_first = first group, dfirst = first descriptor, foff = float offset of its score blocks.
__global__ __launch_bounds__(1024) void ivf_plan_scores_kernel(const int* __restrict__ lcount, const int* __restrict__ list_count,
                                                               int nlist, int* __restrict__ wg_first, int* __restrict__ cursor,
                                                               int* __restrict__ dfirst, long long* __restrict__ foff,
                                                               int* __restrict__ out, unsigned long long* __restrict__ stats) {
  // three exclusive prefix sums over the lists (query groups, scan descriptors, score floats): wave scans by shuffles, the 16
  // wave totals through LDS -- two barriers per 1024 lists (the Hillis-Steele form took 20 per sum and wrote every descriptor
  // from this one workgroup: 37 us at the C5 share, 86 us over 5 M rows; the descriptors are now written by ivf_scatter_kernel)
  __shared__ int wt_g[16], wt_d[16];
  __shared__ long long wt_f[16];
  __shared__ unsigned long long rows_acc;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int carry = 0, dcarry = 0;
  long long fcarry = 0;
  if (tid == 0) rows_acc = 0ull;
  __syncthreads();
  for (int base = 0; base < nlist; base += 1024) {
    const int l = base + tid;
    const int cnt = l < nlist ? lcount[l] : 0;
    const int lc = l < nlist ? list_count[l] : 0;
    const int nt = (cnt + IVFS_QUERIES - 1) / IVFS_QUERIES;
    const int nch = (lc + IVFS_CHUNK_ROWS - 1) / IVFS_CHUNK_ROWS;
    if (nt && lc) atomicAdd(&rows_acc, (unsigned long long)nt * (unsigned long long)lc);   // rows this list's workgroups stream
    const int pitch = (lc + IVFS_PITCH_ALIGN - 1) / IVFS_PITCH_ALIGN * IVFS_PITCH_ALIGN;
    const long long fl = (long long)cnt * pitch;
    int ig = nt, id = nt * nch;
    long long iff = fl;
    for (int off = 1; off < 64; off <<= 1) {
      const int vg = __shfl_up(ig, off), vd = __shfl_up(id, off);
      const long long vf = __shfl_up(iff, off);
      if (lane >= off) { ig += vg; id += vd; iff += vf; }
    }
    if (lane == 63) { wt_g[w] = ig; wt_d[w] = id; wt_f[w] = iff; }
    __syncthreads();
    int bg = 0, bd = 0, tg = 0, td = 0;
    long long bf = 0, tf = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int g_ = wt_g[i], d_ = wt_d[i];
      const long long f_ = wt_f[i];
      if (i < w) { bg += g_; bd += d_; bf += f_; }
      tg += g_; td += d_; tf += f_;
    }
    if (l < nlist) {
      wg_first[l] = carry + bg + ig - nt;
      dfirst[l] = dcarry + bd + id - nt * nch;
      foff[l] = fcarry + bf + iff - fl;
      cursor[l] = 0;
    }
    carry += tg; dcarry += td; fcarry += tf;     // (every thread keeps the same running totals)
    __syncthreads();                             // the wave totals are rewritten by the next 1024 lists
  }
  if (tid == 0) {
    out[0] = dcarry;
    out[1] = (int)(unsigned)(fcarry & 0xFFFFFFFFll);
    out[2] = (int)(fcarry >> 32);
    if (stats) { atomicAdd(&stats[0], rows_acc); atomicAdd(&stats[1], (unsigned long long)dcarry); }
  }
}

// nprobe == nlist: every query probes every list (the exhaustive limit, == brute force)
__global__ void ivf_all_lists_kernel(int64_t* __restrict__ probes, int64_t npairs, int nlist) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npairs) probes[i] = i % nlist;
}

__global__ void ids_to_i32_kernel(const int64_t* __restrict__ ids, int64_t n, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)ids[i];
}

template <int BF16>
__global__ void kmeans_accum_kernel(const uint16_t* __restrict__ x, const int64_t* __restrict__ a, int64_t n, int ld, int dim,
                                    float* __restrict__ sums, int* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int64_t c = a[r];
  if (c < 0) return;
  const uint16_t* row = x + (size_t)r * ld;
  for (int i = lane; i < dim; i += 64) {
    float v;
    if (BF16) v = __uint_as_float((uint32_t)row[i] << 16);
    else { _Float16 h; __builtin_memcpy(&h, &row[i], 2); v = (float)h; }
    atomicAdd(&sums[(size_t)c * dim + i], v);
  }
  if (lane == 0) atomicAdd(&cnt[c], 1);
}

template <int BF16>
__global__ void kmeans_keep_empty_kernel(const uint16_t* __restrict__ cen, int ld, int dim, int nlist, const int* __restrict__ cnt,
                                         float* __restrict__ sums) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)nlist * dim) return;
  const int c = (int)(i / dim), d = (int)(i % dim);
  if (cnt[c] > 0) return;
  const uint16_t b = cen[(size_t)c * ld + d];
  float v;
  if (BF16) v = __uint_as_float((uint32_t)b << 16);
  else { _Float16 h; __builtin_memcpy(&h, &b, 2); v = (float)h; }
  sums[i] = v;   // an empty list keeps its centroid
}

static int ivf_grow(IvfIndex* ix, int64_t need, hipStream_t stream) {
  if (need <= ix->cap) return MRAG_OK;
  int64_t cap = bf_round_rows(std::max<int64_t>(need, ix->cap + ix->cap / 2));
  uint16_t* nr = nullptr;
  int32_t* na = nullptr;
  MRAG_HIP(hipMalloc((void**)&nr, (size_t)cap * ix->ld * 2));
  MRAG_HIP(hipMalloc((void**)&na, (size_t)cap * 4));
  MRAG_HIP(hipMemsetAsync(nr, 0, (size_t)cap * ix->ld * 2, stream));
  if (ix->n) {
    MRAG_HIP(hipMemcpyAsync(nr, ix->raw, (size_t)ix->n * ix->ld * 2, hipMemcpyDeviceToDevice, stream));
    MRAG_HIP(hipMemcpyAsync(na, ix->assign, (size_t)ix->n * 4, hipMemcpyDeviceToDevice, stream));
  }
  MRAG_HIP(hipStreamSynchronize(stream));
  if (ix->raw) (void)hipFree(ix->raw);
  if (ix->assign) (void)hipFree(ix->assign);
  ix->raw = nr; ix->assign = na; ix->cap = cap;
  return MRAG_OK;
}

// nearest centroid (k = 1) of prepared rows x [n][ld] (n need not be a multiple of 256: the
// buffer behind it must be readable up to the next multiple) -> out_ids int64 [n]
static int ivf_assign_rows(IvfIndex* ix, const uint16_t* x, int64_t n, int64_t* out_ids, float* out_sc, hipStream_t stream) {
  BfLaunch a;
  a.corpus = ix->cen; a.queries = x; a.ld = ix->ld; a.dtype = ix->dtype; a.k = 1;
  a.nq = n; a.n_rows = ix->nlist; a.id_base = 0;
  a.out_scores = out_sc; a.out_ids = out_ids;
  a.lists = &ix->lists; a.counts = &ix->counts; a.stream = stream;
  return bf_launch(a);
}

// rows (any dtype, host or device) -> prepared storage rows at dst (device), padded buffer rows zeroed by caller
static int ivf_prepare(IvfIndex* ix, const void* rows, int64_t n, int src_dtype, int normalize, int is_device, uint16_t* dst,
                       hipStream_t stream) {
  const void* src = rows;
  if (!is_device) {
    const size_t bytes = (size_t)n * ix->dim * esize(src_dtype);
    MRAG_TRY(ix->stage_in.ensure(bytes));
    MRAG_HIP(hipMemcpyAsync(ix->stage_in.p, rows, bytes, hipMemcpyHostToDevice, stream));
    src = ix->stage_in.p;
  }
  return launch_prep_rows(src, src_dtype, n, ix->dim, dst, ix->ld, ix->dtype, normalize && ix->metric == MRAG_METRIC_COSINE, stream);
}

static uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// regroup rows by list: stable by original row, each list padded to a multiple of 256 rows
static int ivf_finalize(IvfIndex* ix, hipStream_t stream) {
  if (!ix->dirty) return MRAG_OK;
  const int64_t n = ix->n;
  std::vector<int32_t> a((size_t)n);
  if (n) MRAG_HIP(hipMemcpyAsync(a.data(), ix->assign, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
  MRAG_HIP(hipStreamSynchronize(stream));
  ix->list_count.assign(ix->nlist, 0);
  for (int64_t i = 0; i < n; ++i) ix->list_count[a[i]]++;
  ix->list_tile_lo.assign(ix->nlist, 0);
  int64_t tiles = 0;
  for (int l = 0; l < ix->nlist; ++l) { ix->list_tile_lo[l] = (int)tiles; tiles += (ix->list_count[l] + 255) / 256; }
  const int64_t ns = std::max<int64_t>(tiles, 1) * 256;
  std::vector<int64_t> perm((size_t)ns, -1);
  std::vector<int64_t> cur(ix->nlist);
  for (int l = 0; l < ix->nlist; ++l) cur[l] = (int64_t)ix->list_tile_lo[l] * 256;
  for (int64_t i = 0; i < n; ++i) perm[cur[a[i]]++] = i;
  if (ix->sorted) (void)hipFree(ix->sorted);
  if (ix->row_ids) (void)hipFree(ix->row_ids);
  ix->sorted = nullptr; ix->row_ids = nullptr;
  MRAG_HIP(hipMalloc((void**)&ix->sorted, (size_t)ns * ix->ld * 2));
  MRAG_HIP(hipMalloc((void**)&ix->row_ids, (size_t)ns * 8));
  MRAG_HIP(hipMemcpyAsync(ix->row_ids, perm.data(), (size_t)ns * 8, hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((ns + 3) / 4)), dim3(256), 0, stream, ix->raw ? ix->raw : ix->sorted, ix->row_ids, ns, ix->ld, ix->sorted, 1);
  MRAG_HIP(hipGetLastError());
  MRAG_TRY(ix->d_list_count.ensure((size_t)ix->nlist * 4));
  MRAG_TRY(ix->d_list_tile_lo.ensure((size_t)ix->nlist * 4));
  MRAG_HIP(hipMemcpyAsync(ix->d_list_count.p, ix->list_count.data(), (size_t)ix->nlist * 4, hipMemcpyHostToDevice, stream));
  MRAG_HIP(hipMemcpyAsync(ix->d_list_tile_lo.p, ix->list_tile_lo.data(), (size_t)ix->nlist * 4, hipMemcpyHostToDevice, stream));
  MRAG_HIP(hipStreamSynchronize(stream));
  ix->n_sorted = ns;
  ix->max_list_rows = ix->list_count.empty() ? 0 : *std::max_element(ix->list_count.begin(), ix->list_count.end());
  ix->dirty = false;
  return MRAG_OK;
}

}  // namespace mrag

using namespace mrag;

extern "C" {

int mrag_ivf_create(int dim, int nlist, int metric, int storage_dtype, int device, mrag_handle* out) {
  if (!out) return fail(MRAG_ERR_INVALID, "out is NULL");
  if (dim <= 0 || dim > 8192) return fail(MRAG_ERR_INVALID, "dim %d out of range", dim);
  if (nlist <= 0 || nlist > (1 << 20)) return fail(MRAG_ERR_INVALID, "nlist %d out of range", nlist);
  if (storage_dtype != MRAG_F16 && storage_dtype != MRAG_BF16) return fail(MRAG_ERR_INVALID, "storage dtype must be fp16 or bf16");
  MRAG_TRY(use_device(device));
  IvfIndex* ix = new IvfIndex();
  ix->kind = KIND_IVF; ix->device = device; ix->dim = dim; ix->ld = (int)round_up(dim, 64);
  ix->nlist = nlist; ix->metric = metric; ix->dtype = storage_dtype;
  const size_t cb = (size_t)bf_round_rows(nlist) * ix->ld * 2;
  if (hipMalloc((void**)&ix->cen, cb) != hipSuccess) { delete ix; return fail(MRAG_ERR_OOM, "centroid allocation failed"); }
  (void)hipMemset(ix->cen, 0, cb);
  for (auto& e : ix->ev)
    if (hipEventCreate(&e) != hipSuccess) { delete ix; return fail(MRAG_ERR_HIP, "hipEventCreate failed"); }
  *out = register_object(ix);
  return MRAG_OK;
}

int mrag_ivf_destroy(mrag_handle h) {
  Object* o = take(h, KIND_IVF);
  if (!o) return MRAG_ERR_INVALID;
  (void)hipSetDevice(o->device);
  (void)hipDeviceSynchronize();
  delete o;
  return MRAG_OK;
}

int mrag_ivf_set_centroids(mrag_handle h, const void* centroids, int src_dtype, int normalize, int is_device, void* stream_) {
  IvfIndex* ix = (IvfIndex*)lookup(h, KIND_IVF);
  if (!ix) return MRAG_ERR_INVALID;
  if (!centroids || !esize(src_dtype)) return fail(MRAG_ERR_INVALID, "bad centroid buffer / dtype");
  if (ix->n) return fail(MRAG_ERR_INVALID, "centroids cannot change after rows were added");
  MRAG_TRY(use_device(ix->device));
  hipStream_t stream = (hipStream_t)stream_;
  MRAG_TRY(ivf_prepare(ix, centroids, ix->nlist, src_dtype, normalize, is_device, ix->cen, stream));
  MRAG_HIP(hipStreamSynchronize(stream));
  ix->has_centroids = true;
  return MRAG_OK;
}

int mrag_ivf_get_centroids(mrag_handle h, float* out, int out_is_device, void* stream_) {
  IvfIndex* ix = (IvfIndex*)lookup(h, KIND_IVF);
  if (!ix) return MRAG_ERR_INVALID;
  if (!out) return fail(MRAG_ERR_INVALID, "out is NULL");
  MRAG_TRY(use_device(ix->device));
  hipStream_t stream = (hipStream_t)stream_;
  std::vector<uint16_t> hc((size_t)ix->nlist * ix->ld);
  MRAG_HIP(hipMemcpyAsync(hc.data(), ix->cen, hc.size() * 2, hipMemcpyDeviceToHost, stream));
  MRAG_HIP(hipStreamSynchronize(stream));
  std::vector<float> f((size_t)ix->nlist * ix->dim);
  for (int r = 0; r < ix->nlist; ++r)
    for (int c = 0; c < ix->dim; ++c) {
      const uint16_t b = hc[(size_t)r * ix->ld + c];
      float v;
      if (ix->dtype == MRAG_BF16) { uint32_t u = (uint32_t)b << 16; memcpy(&v, &u, 4); }
      else { _Float16 hh; memcpy(&hh, &b, 2); v = (float)hh; }
      f[(size_t)r * ix->dim + c] = v;
    }
  if (out_is_device) MRAG_HIP(hipMemcpy(out, f.data(), f.size() * 4, hipMemcpyHostToDevice));
  else memcpy(out, f.data(), f.size() * 4);
  return MRAG_OK;
}

int mrag_ivf_train(mrag_handle h, const void* rows, int64_t n, int src_dtype, int normalize, int rows_is_device, int iters,
                   uint64_t seed, void* stream_) {
  IvfIndex* ix = (IvfIndex*)lookup(h, KIND_IVF);
  if (!ix) return MRAG_ERR_INVALID;
  if (!rows || !esize(src_dtype)) return fail(MRAG_ERR_INVALID, "bad rows buffer / dtype");
  if (n < ix->nlist) return fail(MRAG_ERR_INVALID, "need at least nlist = %d training rows, got %lld", ix->nlist, (long long)n);
  if (ix->n) return fail(MRAG_ERR_INVALID, "train before adding rows");
  if (iters < 0) return fail(MRAG_ERR_INVALID, "iters < 0");
  MRAG_TRY(use_device(ix->device));
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t npad = bf_round_rows(n);
  DevBuf x;
  MRAG_TRY(x.ensure((size_t)npad * ix->ld * 2));
  MRAG_HIP(hipMemsetAsync(x.p, 0, (size_t)npad * ix->ld * 2, stream));
  int st = ivf_prepare(ix, rows, n, src_dtype, normalize, rows_is_device, (uint16_t*)x.p, stream);
  if (st != MRAG_OK) { x.release(); return st; }
  // init: nlist distinct rows (Floyd's sampling from splitmix64(seed)), in ascending row order
  std::vector<int64_t> pick;
  {
    std::vector<int64_t> chosen;
    uint64_t s = seed;
    for (int64_t j = n - ix->nlist; j < n; ++j) {
      int64_t t = (int64_t)(splitmix64(s) % (uint64_t)(j + 1));
      if (std::find(chosen.begin(), chosen.end(), t) != chosen.end()) t = j;
      chosen.push_back(t);
    }
    std::sort(chosen.begin(), chosen.end());
    pick = chosen;
  }
  DevBuf dpick, ids, sc;
  st = dpick.ensure((size_t)ix->nlist * 8);
  if (st == MRAG_OK) st = ids.ensure((size_t)n * 8);
  if (st == MRAG_OK) st = sc.ensure((size_t)n * 4);
  if (st == MRAG_OK) st = ix->sums.ensure((size_t)ix->nlist * ix->dim * 4);
  if (st == MRAG_OK) st = ix->cnts.ensure((size_t)ix->nlist * 4);
  auto cleanup = [&]() { x.release(); dpick.release(); ids.release(); sc.release(); };
  if (st != MRAG_OK) { cleanup(); return st; }
  hipError_t e = hipMemcpyAsync(dpick.p, pick.data(), (size_t)ix->nlist * 8, hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) { cleanup(); return fail(MRAG_ERR_HIP, "H2D failed"); }
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((ix->nlist + 3) / 4)), dim3(256), 0, stream, (const uint16_t*)x.p,
                     (const int64_t*)dpick.p, (int64_t)ix->nlist, ix->ld, ix->cen, 1);
  (void)hipStreamSynchronize(stream);
  for (int it = 0; it < iters; ++it) {
    st = ivf_assign_rows(ix, (const uint16_t*)x.p, n, (int64_t*)ids.p, (float*)sc.p, stream);
    if (st != MRAG_OK) { cleanup(); return st; }
    (void)hipMemsetAsync(ix->sums.p, 0, (size_t)ix->nlist * ix->dim * 4, stream);
    (void)hipMemsetAsync(ix->cnts.p, 0, (size_t)ix->nlist * 4, stream);
    const dim3 g((unsigned)((n + 3) / 4)), b(256);
    const dim3 g2((unsigned)(((int64_t)ix->nlist * ix->dim + 255) / 256));
    if (ix->dtype == MRAG_BF16) {
      hipLaunchKernelGGL((kmeans_accum_kernel<1>), g, b, 0, stream, (const uint16_t*)x.p, (const int64_t*)ids.p, n, ix->ld, ix->dim, (float*)ix->sums.p, (int*)ix->cnts.p);
      hipLaunchKernelGGL((kmeans_keep_empty_kernel<1>), g2, b, 0, stream, ix->cen, ix->ld, ix->dim, ix->nlist, (const int*)ix->cnts.p, (float*)ix->sums.p);
    } else {
      hipLaunchKernelGGL((kmeans_accum_kernel<0>), g, b, 0, stream, (const uint16_t*)x.p, (const int64_t*)ids.p, n, ix->ld, ix->dim, (float*)ix->sums.p, (int*)ix->cnts.p);
      hipLaunchKernelGGL((kmeans_keep_empty_kernel<0>), g2, b, 0, stream, ix->cen, ix->ld, ix->dim, ix->nlist, (const int*)ix->cnts.p, (float*)ix->sums.p);
    }
    st = launch_prep_rows(ix->sums.p, MRAG_F32, ix->nlist, ix->dim, ix->cen, ix->ld, ix->dtype, ix->metric == MRAG_METRIC_COSINE, stream);
    if (st != MRAG_OK) { cleanup(); return st; }
  }
  e = hipStreamSynchronize(stream);
  cleanup();
  if (e != hipSuccess) return fail(MRAG_ERR_HIP, "k-means failed: %s", hipGetErrorString(e));
  ix->has_centroids = true;
  return MRAG_OK;
}

int mrag_ivf_add(mrag_handle h, const void* rows, int64_t n, int src_dtype, int normalize, int rows_is_device, void* stream_) {
  IvfIndex* ix = (IvfIndex*)lookup(h, KIND_IVF);
  if (!ix) return MRAG_ERR_INVALID;
  if (n < 0) return fail(MRAG_ERR_INVALID, "n < 0");
  if (n == 0) return MRAG_OK;
  if (!rows || !esize(src_dtype)) return fail(MRAG_ERR_INVALID, "bad rows buffer / dtype");
  if (!ix->has_centroids) return fail(MRAG_ERR_INVALID, "train or set centroids before adding rows");
  if (ix->n + n > 0x7FFFFF00ll) return fail(MRAG_ERR_UNSUPPORTED, "more than 2^31 rows per index; shard the corpus");
  MRAG_TRY(use_device(ix->device));
  hipStream_t stream = (hipStream_t)stream_;
  MRAG_TRY(ivf_grow(ix, ix->n + n, stream));
  uint16_t* dst = ix->raw + (size_t)ix->n * ix->ld;
  MRAG_TRY(ivf_prepare(ix, rows, n, src_dtype, normalize, rows_is_device, dst, stream));
  MRAG_TRY(ix->tmp_id.ensure((size_t)n * 8));
  MRAG_TRY(ix->tmp_sc.ensure((size_t)n * 4));
  // (the raw buffer is a multiple of 256 rows and zero beyond n: safe to read as padded queries)
  MRAG_TRY(ivf_assign_rows(ix, dst, n, (int64_t*)ix->tmp_id.p, (float*)ix->tmp_sc.p, stream));
  hipLaunchKernelGGL(ids_to_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const int64_t*)ix->tmp_id.p, n, ix->assign + ix->n);
  MRAG_HIP(hipGetLastError());
  MRAG_HIP(hipStreamSynchronize(stream));
  ix->n += n;
  ix->dirty = true;
  return MRAG_OK;
}

int mrag_ivf_size(mrag_handle h, int64_t* out_rows) {
  IvfIndex* ix = (IvfIndex*)lookup(h, KIND_IVF);
  if (!ix) return MRAG_ERR_INVALID;
  if (!out_rows) return fail(MRAG_ERR_INVALID, "out_rows is NULL");
  *out_rows = ix->n;
  return MRAG_OK;
}

int mrag_ivf_set_id_base(mrag_handle h, int64_t id_base) {
  IvfIndex* ix = (IvfIndex*)lookup(h, KIND_IVF);
  if (!ix) return MRAG_ERR_INVALID;
  ix->id_base = id_base;
  return MRAG_OK;
}

int mrag_ivf_get_assignments(mrag_handle h, int32_t* out, int out_is_device, void* stream_) {
  IvfIndex* ix = (IvfIndex*)lookup(h, KIND_IVF);
  if (!ix) return MRAG_ERR_INVALID;
  if (!out && ix->n) return fail(MRAG_ERR_INVALID, "out is NULL");
  if (!ix->n) return MRAG_OK;
  MRAG_TRY(use_device(ix->device));
  hipStream_t stream = (hipStream_t)stream_;
  MRAG_HIP(hipMemcpyAsync(out, ix->assign, (size_t)ix->n * 4, out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
  MRAG_HIP(hipStreamSynchronize(stream));
  return MRAG_OK;
}

// one chunk of a search (<= 2^20 queries whose score segments fit the buffer): probe selection, plan, list scan, selection
static int ivf_search_chunk(IvfIndex* ix, const void* queries, int64_t nq, int q_dtype, int normalize, int queries_is_device, int nprobe,
                            int k, float* out_scores, int64_t* out_ids, int out_is_device, hipStream_t stream, int chunk_idx) {
  hipEvent_t ev_scan0 = ix->cev[2 * chunk_idx], ev_scan1 = ix->cev[2 * chunk_idx + 1];
  unsigned long long* d_stats = (unsigned long long*)ix->stats.p;

  float* d_sc = out_scores;
  int64_t* d_id = out_ids;
  if (!out_is_device) {
    MRAG_TRY(ix->out_sc.ensure((size_t)nq * k * 4));
    MRAG_TRY(ix->out_id.ensure((size_t)nq * k * 8));
    d_sc = (float*)ix->out_sc.p;
    d_id = (int64_t*)ix->out_id.p;
  }
  // 1) queries -> storage dtype; probe lists
  const int64_t nq_pad = bf_round_rows(nq) + 16;   // + 16 zero rows: the streaming kernel's last 8-query group reads a 16-row block
  MRAG_TRY(ix->qbuf.ensure((size_t)nq_pad * ix->ld * 2));
  MRAG_HIP(hipMemsetAsync((char*)ix->qbuf.p + (size_t)nq * ix->ld * 2, 0, (size_t)(nq_pad - nq) * ix->ld * 2, stream));   // (rows < nq: written whole by the prepare kernel)
  MRAG_TRY(ivf_prepare(ix, queries, nq, q_dtype, normalize, queries_is_device, (uint16_t*)ix->qbuf.p, stream));
  MRAG_TRY(ix->tmp_id.ensure((size_t)nq * nprobe * 8));
  MRAG_TRY(ix->tmp_sc.ensure((size_t)nq * nprobe * 4));
  // The list scan (and the probe selection before it) runs as "score segments + select" (ivf_scan.hip) whenever the
  // fp32 segments of this search fit the score buffer; otherwise (exhaustive probing of a big index) as the fused
  // GEMM + top-k kernel in descriptor mode.  MRAG_IVF_SCORES_MB = 0 forces the fused path (tests).
  static const int64_t scores_cap = [] { const char* e = getenv("MRAG_IVF_SCORES_MB"); return (int64_t)(e ? atoll(e) : 16384) << 20; }();
  const int nl = ix->nlist;
  MRAG_TRY(ix->plan.ensure((size_t)(4 * nl + 4) * 4 + (size_t)nl * 8));
  int* d_lcount = (int*)ix->plan.p;      // queries per list | first group of the list | cursor | first descriptor | descriptor count, score floats | float offset of the list's score blocks
  int* d_wg_first = d_lcount + nl;
  int* d_cursor = d_wg_first + nl;
  int* d_dfirst = d_cursor + nl;
  int* d_nwg = d_dfirst + nl;
  long long* d_foff = (long long*)(d_nwg + 4);
  MRAG_HIP(hipMemsetAsync(d_lcount, 0, (size_t)nl * 4, stream));
  bool counted = false;                  // the probe selection below can fill d_lcount itself
  if (nprobe == ix->nlist) {
    const int64_t np_ = nq * nprobe;
    hipLaunchKernelGGL(ivf_all_lists_kernel, dim3((unsigned)((np_ + 255) / 256)), dim3(256), 0, stream, (int64_t*)ix->tmp_id.p, np_, ix->nlist);
    MRAG_HIP(hipGetLastError());
  } else if (round_up(nq, IVFS_QUERIES) * ivfs_pitch(ix->nlist) * 4 <= scores_cap) {
    MRAG_TRY(ix->sdesc.ensure((size_t)ivfs_dense_n_desc(nq, ix->nlist) * IVFS_DESC_WORDS * 4));
    MRAG_TRY(ix->scores.ensure((size_t)(round_up(nq, IVFS_QUERIES) * ivfs_pitch(ix->nlist) * 4)));
    MRAG_TRY(ivfs_dense_topk(ix->cen, ix->nlist, (const uint16_t*)ix->qbuf.p, nq, ix->ld, ix->dtype, nprobe, (int*)ix->sdesc.p,
                             (float*)ix->scores.p, (float*)ix->tmp_sc.p, (int64_t*)ix->tmp_id.p, stream, d_lcount,
                             (const int*)ix->d_list_count.p));
    counted = true;
  } else {
    BfLaunch a;
    a.corpus = ix->cen; a.queries = (const uint16_t*)ix->qbuf.p; a.ld = ix->ld; a.dtype = ix->dtype; a.k = nprobe;
    a.nq = nq; a.n_rows = ix->nlist;
    a.out_scores = (float*)ix->tmp_sc.p; a.out_ids = (int64_t*)ix->tmp_id.p;
    a.lists = &ix->lists; a.counts = &ix->counts; a.stream = stream;
    MRAG_TRY(bf_launch(a));
  }
  // 2) device: list -> the queries that probe it, cut into workgroups of <= 128 (score segments) or <= 256 (fused
  //    kernel) queries.  Only the workgroup count and the size of the segments come back to the host (12 bytes).
  const size_t npairs = (size_t)nq * nprobe;
  const int64_t wg_bound = (int64_t)std::min<size_t>((size_t)nl, npairs) + (int64_t)(npairs / IVFS_QUERIES) + 1;   // (covers the 256-query plan too)
  // score-segment regime: every group is cut into ceil(list rows / IVFS_CHUNK_ROWS) scan descriptors
  const int64_t desc_bound = wg_bound * std::max<int64_t>(1, (ix->max_list_rows + IVFS_CHUNK_ROWS - 1) / IVFS_CHUNK_ROWS);
  MRAG_TRY(ix->desc.ensure((size_t)desc_bound * 8 * 4));
  MRAG_TRY(ix->gq.ensure((size_t)wg_bound * 256 * 8));
  MRAG_TRY(ix->ploc.ensure(npairs * 16));   // int2 (fused regime) or int4 (score segments) per pair
  const unsigned pgrid = (unsigned)((npairs + 255) / 256);
  if (!counted)
    hipLaunchKernelGGL(ivf_count_kernel, dim3(pgrid), dim3(256), 0, stream, (const int64_t*)ix->tmp_id.p, (int64_t)npairs,
                       (const int*)ix->d_list_count.p, d_lcount);
  bool use_scores = scores_cap > 0 && nprobe <= 256;   // (the select kernel holds <= 256 segments per query; exhaustive probing takes the fused path)
  // Score floats of this search: known exactly only on the device (sum over the probed lists of queries x rows).  When
  // even the bound "every pair probes the longest list" fits the buffer, nothing comes back to the host at all: the
  // scan is launched over the BOUND of the workgroup count and reads the real one from device memory.
  const int64_t max_pitch = ivfs_pitch(ix->max_list_rows);
  const bool no_sync = use_scores && (int64_t)npairs * max_pitch * 4 <= scores_cap;
  int plan_out[3] = {0, 0, 0};
  const bool planned_scores = use_scores;
  if (use_scores) {
    hipLaunchKernelGGL(ivf_plan_scores_kernel, dim3(1), dim3(1024), 0, stream, (const int*)d_lcount, (const int*)ix->d_list_count.p,
                       nl, d_wg_first, d_cursor, d_dfirst, d_foff, d_nwg, d_stats);
    MRAG_HIP(hipGetLastError());
    if (no_sync) {
      MRAG_TRY(ix->scores.ensure((size_t)std::max<int64_t>((int64_t)npairs * max_pitch, 4) * 4));
    } else {
      MRAG_HIP(hipMemcpyAsync(plan_out, d_nwg, 12, hipMemcpyDeviceToHost, stream));
      MRAG_HIP(hipStreamSynchronize(stream));
      const int64_t floats = (int64_t)(uint32_t)plan_out[1] | ((int64_t)plan_out[2] << 32);
      if (plan_out[0] < 0 || plan_out[0] > desc_bound) return fail(MRAG_ERR_HIP, "IVF plan produced %d workgroups (bound %lld)", plan_out[0], (long long)desc_bound);
      if (floats * 4 > scores_cap) use_scores = false;
      else MRAG_TRY(ix->scores.ensure((size_t)std::max<int64_t>(floats, 4) * 4));
    }
  }
  int n_wg = -1;   // -1: still on the device (read back by mrag_ivf_last_timing if asked)
  if (use_scores) {
    hipLaunchKernelGGL(ivf_scatter_kernel, dim3(pgrid), dim3(256), 0, stream, (const int64_t*)ix->tmp_id.p, (int64_t)npairs, nprobe,
                       (const int*)ix->d_list_count.p, (const int*)d_wg_first, d_cursor, (int64_t*)ix->gq.p, nullptr, 7,
                       (const long long*)d_foff, (const int*)ix->d_list_tile_lo.p, (int4*)ix->ploc.p, (const int*)d_lcount,
                       (const int*)d_dfirst, (int*)ix->desc.p, (int)desc_bound);
    MRAG_HIP(hipGetLastError());
    // 3) scores of every (query, probed list) pair, then the k best per query
    MRAG_HIP(hipEventRecord(ev_scan0, stream));
    static const char* stamp_path = getenv("MRAG_IVFS_STAMPS");   // diagnostic builds (MRAG_IVFS_DIAG & 128): per-workgroup clock stamps -> file
    long long* dbg = nullptr;
    if (stamp_path) { MRAG_TRY(ix->qg.ensure((size_t)desc_bound * 64)); MRAG_HIP(hipMemsetAsync(ix->qg.p, 0, (size_t)desc_bound * 64, stream)); dbg = (long long*)ix->qg.p; }
    if (no_sync) {
      MRAG_TRY(ivfs_scan(ix->sorted, (const uint16_t*)ix->qbuf.p, ix->ld, ix->dtype, nq, (const int*)ix->desc.p, (int)desc_bound, d_nwg,
                         (const int64_t*)ix->gq.p, (float*)ix->scores.p, stream, dbg));
      if (stamp_path) {
        std::vector<long long> h((size_t)desc_bound * 8);
        MRAG_HIP(hipStreamSynchronize(stream));
        MRAG_HIP(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(stamp_path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
      }
    } else {
      n_wg = plan_out[0];
      MRAG_TRY(ivfs_scan(ix->sorted, (const uint16_t*)ix->qbuf.p, ix->ld, ix->dtype, nq, (const int*)ix->desc.p, n_wg, nullptr,
                         (const int64_t*)ix->gq.p, (float*)ix->scores.p, stream));
    }
    MRAG_TRY(ivfs_select_lists((const float*)ix->scores.p, ix->ploc.p, nprobe, nq, k, ix->row_ids, ix->id_base, d_sc, d_id, stream));
    MRAG_HIP(hipEventRecord(ev_scan1, stream));
  } else {
    MRAG_HIP(hipMemsetAsync(ix->gq.p, 0xFF, (size_t)wg_bound * 256 * 8, stream));   // (the gather below reads whole 256-slot slabs)
    hipLaunchKernelGGL(ivf_plan_kernel, dim3(1), dim3(1024), 0, stream, (const int*)d_lcount, (const int*)ix->d_list_count.p,
                       (const int*)ix->d_list_tile_lo.p, nl, d_wg_first, d_cursor, (int*)ix->desc.p, (int)wg_bound, d_nwg,
                       planned_scores ? nullptr : d_stats);   // (a scores plan that turned out too big has already counted this chunk)
    hipLaunchKernelGGL(ivf_scatter_kernel, dim3(pgrid), dim3(256), 0, stream, (const int64_t*)ix->tmp_id.p, (int64_t)npairs, nprobe,
                       (const int*)ix->d_list_count.p, (const int*)d_wg_first, d_cursor, (int64_t*)ix->gq.p, (int2*)ix->ploc.p, 8,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
    MRAG_HIP(hipGetLastError());
    MRAG_HIP(hipMemcpyAsync(&n_wg, d_nwg, 4, hipMemcpyDeviceToHost, stream));
    MRAG_HIP(hipStreamSynchronize(stream));
    if (n_wg < 0 || n_wg > wg_bound) return fail(MRAG_ERR_HIP, "IVF plan produced %d workgroups (bound %lld)", n_wg, (long long)wg_bound);
    // 3) gather queries per workgroup, scan the lists, merge
    if (n_wg) {
      MRAG_TRY(ix->qg.ensure((size_t)n_wg * 256 * ix->ld * 2));
      const int64_t ng = (int64_t)n_wg * 256;
      hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((ng + 3) / 4)), dim3(256), 0, stream, (const uint16_t*)ix->qbuf.p,
                         (const int64_t*)ix->gq.p, ng, ix->ld, (uint16_t*)ix->qg.p, 0);
      MRAG_HIP(hipGetLastError());
    }
    BfLaunch a;
    a.corpus = ix->sorted; a.queries = (const uint16_t*)ix->qg.p; a.ld = ix->ld; a.dtype = ix->dtype; a.k = k;
    a.nq = nq;
    static const int dummy_desc = 0;
    a.wg_desc = n_wg ? (const int*)ix->desc.p : &dummy_desc; a.n_wg = n_wg;
    a.pair_loc = ix->ploc.p; a.nprobe = nprobe; a.row_ids = ix->row_ids;
    a.id_base = ix->id_base;
    a.out_scores = d_sc; a.out_ids = d_id;
    a.lists = &ix->lists; a.counts = &ix->counts; a.stream = stream;
    a.ev_k2_begin = ev_scan0; a.ev_k2_end = ev_scan1;
    MRAG_TRY(bf_launch(a));
  }
  ix->last_scores_path = use_scores;
  (void)n_wg;
  if (!out_is_device) {
    MRAG_HIP(hipMemcpyAsync(out_scores, d_sc, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream));
    MRAG_HIP(hipMemcpyAsync(out_ids, d_id, (size_t)nq * k * 8, hipMemcpyDeviceToHost, stream));
  }
  return MRAG_OK;
}


int mrag_ivf_search(mrag_handle h, const void* queries, int64_t nq, int q_dtype, int normalize, int queries_is_device, int nprobe,
                    int k, float* out_scores, int64_t* out_ids, int out_is_device, void* stream_) {
  IvfIndex* ix = (IvfIndex*)lookup(h, KIND_IVF);
  if (!ix) return MRAG_ERR_INVALID;
  if (nq < 0 || k <= 0 || nprobe <= 0) return fail(MRAG_ERR_INVALID, "bad nq / k / nprobe");
  if (k > bf_max_k()) return fail(MRAG_ERR_UNSUPPORTED, "k = %d exceeds the fused top-k limit %d", k, bf_max_k());
  nprobe = std::min(nprobe, ix->nlist);
  // probe selection = top-nprobe over the centroids: the fused kernel up to 64, the streaming kernel up to 256
  // (8 queries per launch); nprobe == nlist needs no selection at all
  if (nprobe > bf_max_k_wide() && nprobe < ix->nlist)
    return fail(MRAG_ERR_UNSUPPORTED, "nprobe = %d: supported are 1..%d and nlist (= %d, exhaustive)", nprobe, bf_max_k_wide(), ix->nlist);
  if (nq == 0) return MRAG_OK;
  if (!queries || !out_scores || !out_ids || !esize(q_dtype)) return fail(MRAG_ERR_INVALID, "bad buffer / dtype");
  if (!ix->has_centroids) return fail(MRAG_ERR_INVALID, "index has no centroids");
  MRAG_TRY(use_device(ix->device));
  hipStream_t stream = (hipStream_t)stream_;
  MRAG_TRY(ivf_finalize(ix, stream));
  // Large batches are cut so that a chunk's score segments fit the score buffer even if every pair probes the longest
  // list (ivf_scan.hip's regime, without a host round trip); results are per query, so the chunks are independent.
  // Every chunk streams the probed lists again, so the default buffer is generous (16 GiB of the 288: C5's 10 000 x 32
  // pairs over 5 M rows need 1.6 GB, their no-round-trip bound ~5 GB); MRAG_IVF_SCORES_MB sets it, 0 = fused regime.
  static const int64_t cap = [] { const char* e = getenv("MRAG_IVF_SCORES_MB"); return (int64_t)(e ? atoll(e) : 16384) << 20; }();
  const int64_t per_query = std::max<int64_t>((int64_t)std::min(nprobe, 256) * ivfs_pitch(ix->max_list_rows), ivfs_pitch(ix->nlist)) * 4;
  int64_t chunk = std::max<int64_t>(1024, cap / std::max<int64_t>(per_query, 1) / IVFS_QUERIES * IVFS_QUERIES);
  if (!(cap > 0 && nprobe <= 256 && nq > chunk)) chunk = nq;
  else chunk = round_up((nq + (nq + chunk - 1) / chunk - 1) / ((nq + chunk - 1) / chunk), IVFS_QUERIES);   // equal chunks
  if (chunk > (1 << 20)) return fail(MRAG_ERR_UNSUPPORTED, "nq too large for one IVF batch (2^20); cut the query batch");
  const int n_chunks = (int)((nq + chunk - 1) / chunk);
  while ((int)ix->cev.size() < 2 * n_chunks) {
    hipEvent_t e = nullptr;
    MRAG_HIP(hipEventCreate(&e));
    ix->cev.push_back(e);
  }
  ix->timed = false;
  // an asynchronous search (device queries + device results) returns with the handle's scratch buffers still in use on ITS
  // stream; a following search on another stream first waits for that one's end event (a no-op on the same stream or after a
  // synchronous search), so callers may switch streams between searches on one handle
  if (ix->end_recorded) MRAG_HIP(hipStreamWaitEvent(stream, ix->ev[3], 0));
  MRAG_HIP(hipEventRecord(ix->ev[0], stream));
  MRAG_TRY(ix->stats.ensure(16));
  MRAG_HIP(hipMemsetAsync(ix->stats.p, 0, 16, stream));
  const size_t qrow = (size_t)ix->dim * esize(q_dtype);
  for (int c = 0; c < n_chunks; ++c) {
    const int64_t off = (int64_t)c * chunk, m = std::min(chunk, nq - off);
    MRAG_TRY(ivf_search_chunk(ix, (const char*)queries + (size_t)off * qrow, m, q_dtype, normalize, queries_is_device, nprobe, k,
                              out_scores + (size_t)off * k, out_ids + (size_t)off * k, out_is_device, stream, c));
  }
  ix->n_chunks = n_chunks;
  MRAG_HIP(hipEventRecord(ix->ev[3], stream));
  ix->end_recorded = true;
  ix->timed = true;
  // host buffers (pageable query / result memory) must not be touched by the caller before the copies are done; with
  // queries and results in device memory the search is fully asynchronous on `stream`
  if (!out_is_device || !queries_is_device) MRAG_HIP(hipStreamSynchronize(stream));
  return MRAG_OK;
}

int mrag_ivf_last_timing(mrag_handle h, float* out_scan_ms, float* out_total_ms, int64_t* out_scanned_rows, int* out_n_wg) {
  IvfIndex* ix = (IvfIndex*)lookup(h, KIND_IVF);
  if (!ix) return MRAG_ERR_INVALID;
  if (!ix->timed) return fail(MRAG_ERR_INVALID, "no completed search to time");
  MRAG_TRY(use_device(ix->device));
  MRAG_HIP(hipEventSynchronize(ix->ev[3]));
  // a search cut into chunks (score buffer cap) reports the SUM of its chunks' list-scan brackets, rows and workgroups
  float g = 0.f, t = 0.f;
  for (int c = 0; c < ix->n_chunks; ++c) {
    float gc = 0.f;
    MRAG_HIP(hipEventElapsedTime(&gc, ix->cev[2 * c], ix->cev[2 * c + 1]));
    g += gc;
  }
  MRAG_HIP(hipEventElapsedTime(&t, ix->ev[0], ix->ev[3]));
  if (out_scan_ms) *out_scan_ms = g;
  if (out_total_ms) *out_total_ms = t;
  unsigned long long st[2] = {0ull, 0ull};
  MRAG_HIP(hipMemcpy(st, ix->stats.p, 16, hipMemcpyDeviceToHost));   // (accumulated on the device by the plan kernels)
  if (out_n_wg) *out_n_wg = (int)st[1];
  if (out_scanned_rows) *out_scanned_rows = (int64_t)st[0];
  return MRAG_OK;
}

}  // extern "C"
