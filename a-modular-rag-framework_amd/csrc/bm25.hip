// BM25 text channel on the device: the reference's BM25LiteIndex.search
// (app/modules/retrieval/text_index.py:59-97) over CSR postings resident in HBM.
//
//   reference                                   here
//   tf: term -> {doc: tf}        (:27,45-47)    CSR: indptr[term], post_doc[], post_tf[] (docs ascending)
//   _idf(term)                   (:54-56)       computed by the HOST binding with the reference's own expression
//                                               (math.log) and passed per query token: bit-identical
//   _score_doc: for t in q_terms (:59-69)       one launch per token POSITION j: every query's j-th token scatters
//     score += idf*(f*(k1+1))/(f+K(doc))        its postings into that query's dense fp64 score row -- a document
//                                               receives its contributions in token order, exactly the reference's
//                                               left-to-right fp64 sum (same IEEE operations, no contraction)
//   merge over queries: max | sum (:90-91)      bm25_merge_kernel, queries in order
//   keep s > 0, sort desc, [:top_k] (:92-96)    MSB-first radix select over (score bits, ~doc) + rank by counting:
//                                               (score desc, doc asc) -- the reference's tie order is the iteration
//                                               order of a set of ints; doc-ascending is this library's declared order
// HBM-bound integer/byte work: bytes = 8 per posting read + 16 per score read-modify-write, + 8 N per selection pass.
#include "common.h"

#include <algorithm>
#include <vector>

namespace mrag {

struct Bm25Index : Object {
  int64_t n_docs = 0, n_terms = 0, nnz = 0;
  double k1p1 = 2.5;
  int64_t* indptr = nullptr;     // [n_terms + 1]
  int32_t* post_doc = nullptr;   // [nnz]
  int32_t* post_tf = nullptr;    // [nnz]
  double* doc_norm = nullptr;    // [n_docs]  k1 * (1 - b + b * dl / avgdl)
  std::vector<int64_t> h_indptr; // host copy: posting counts size the launches
  DevBuf scores, fin, hist, sel, qbuf;
  ~Bm25Index() override {
    for (void* p : {(void*)indptr, (void*)post_doc, (void*)post_tf, (void*)doc_norm}) if (p) (void)hipFree(p);
    for (DevBuf* b : {&scores, &fin, &hist, &sel, &qbuf}) b->release();
  }
};

constexpr int BM25_MAX_K = 4096;
constexpr int BM25_PASSES = 12;   // 8 bytes of score bits + 4 bytes of ~doc

struct Bm25Job { int64_t lo, hi; double idf; int32_t query; int32_t pad; };   // one (query, token position) posting range

// one token position of every query: jobs[y] = that query's posting range; x covers the longest range
__global__ __launch_bounds__(256) void bm25_scatter_kernel(const Bm25Job* __restrict__ jobs, const int32_t* __restrict__ post_doc,
                                                           const int32_t* __restrict__ post_tf, const double* __restrict__ doc_norm,
                                                           double k1p1, int64_t n_docs, double* __restrict__ scores) {
#pragma clang fp contract(off)
  const Bm25Job j = jobs[blockIdx.y];
  const int64_t i = j.lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= j.hi) return;
  const int32_t d = post_doc[i];
  const double f = (double)post_tf[i];
  const double denom = f + doc_norm[d];
  const double num = j.idf * (f * k1p1);
  double* s = scores + (size_t)j.query * n_docs + d;
  *s = *s + num / denom;      // docs are unique inside one posting list: no two lanes share a word
}

// max / sum over the queries, in query order (text_index.py:90-91); s <= 0 or untouched -> 0 (not a candidate, :92)
__global__ __launch_bounds__(256) void bm25_merge_kernel(const double* __restrict__ scores, int nq, int64_t n_docs, int merge_sum,
                                                         double* __restrict__ fin) {
#pragma clang fp contract(off)
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n_docs) return;
  double s = scores[d];
  for (int q = 1; q < nq; ++q) {
    const double v = scores[(size_t)q * n_docs + d];
    s = merge_sum ? s + v : (v > s ? v : s);
  }
  fin[d] = s > 0.0 ? s : 0.0;
}

// digit p (0 = most significant) of the 96-bit key (score bits : ~doc); positive doubles order like their bit patterns
__device__ __forceinline__ uint32_t bm25_digit(uint64_t sb, uint32_t nd, int p) {
  return p < 8 ? (uint32_t)(sb >> (56 - 8 * p)) & 255u : (nd >> (24 - 8 * (p - 8))) & 255u;
}

// Selection state, derived by every workgroup from the histograms of the passes already run:
// prefix bytes pre[0..np), keys needed inside the prefix bucket, done = bucket wholly selected.
struct Bm25Pick { uint32_t pre[BM25_PASSES]; int np; int64_t need; bool done; };

__device__ inline void bm25_replay(const unsigned* __restrict__ hist, int passes, int64_t k, Bm25Pick& st) {
  st.np = 0; st.need = k; st.done = false;
  for (int p = 0; p < passes && !st.done; ++p) {
    const unsigned* h = hist + p * 256;
    int64_t cum = 0;
    int b = 255;
    for (; b >= 0; --b) {
      if (cum + h[b] >= st.need) break;
      cum += h[b];
    }
    if (b < 0) { st.done = true; st.need = 0; break; }   // fewer than `need` keys exist: everything is selected
    st.pre[st.np++] = (uint32_t)b;
    st.need -= cum;
    if ((int64_t)h[b] == st.need) st.done = true;
  }
}

__device__ __forceinline__ int bm25_cmp_prefix(uint64_t sb, uint32_t nd, const Bm25Pick& st) {
  // -1 below, 0 inside, +1 above the prefix bucket
  for (int p = 0; p < st.np; ++p) {
    const uint32_t dg = bm25_digit(sb, nd, p);
    if (dg != st.pre[p]) return dg > st.pre[p] ? 1 : -1;
  }
  return 0;
}

// pass `pass`: histogram of digit `pass` over the keys inside the current prefix bucket
__global__ __launch_bounds__(256) void bm25_hist_kernel(const double* __restrict__ fin, int64_t n_docs, int pass, int64_t k,
                                                        unsigned* __restrict__ hist) {
  __shared__ unsigned lh[256];
  __shared__ Bm25Pick st;
  if (threadIdx.x == 0) bm25_replay(hist, pass, k, st);
  lh[threadIdx.x] = 0u;
  __syncthreads();
  if (st.done) return;
  for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_docs; d += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t sb = (uint64_t)__double_as_longlong(fin[d]);
    if (sb == 0ull) continue;
    const uint32_t nd = 0xFFFFFFFFu - (uint32_t)d;
    if (bm25_cmp_prefix(sb, nd, st) == 0) atomicAdd(&lh[bm25_digit(sb, nd, pass)], 1u);
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&hist[pass * 256 + threadIdx.x], lh[threadIdx.x]);
}

// gather the selected keys (above the bucket, or inside a wholly selected bucket), unordered
__global__ __launch_bounds__(256) void bm25_gather_kernel(const double* __restrict__ fin, int64_t n_docs, int64_t k,
                                                          const unsigned* __restrict__ hist, int* __restrict__ n_sel,
                                                          int64_t* __restrict__ sel_doc, double* __restrict__ sel_score) {
  __shared__ Bm25Pick st;
  if (threadIdx.x == 0) bm25_replay(hist, BM25_PASSES, k, st);
  __syncthreads();
  for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_docs; d += (int64_t)gridDim.x * blockDim.x) {
    const double s = fin[d];
    const uint64_t sb = (uint64_t)__double_as_longlong(s);
    if (sb == 0ull) continue;
    const int c = bm25_cmp_prefix(sb, 0xFFFFFFFFu - (uint32_t)d, st);
    if (c > 0 || c == 0) {     // after all 12 passes the bucket is one key; an early stop means the bucket is taken whole
      const int slot = atomicAdd(n_sel, 1);
      if (slot < BM25_MAX_K) { sel_doc[slot] = d; sel_score[slot] = s; }
    }
  }
}

// one workgroup: order the <= k selected by (score desc, doc asc) with rank-by-counting
__global__ __launch_bounds__(1024) void bm25_rank_kernel(const int* __restrict__ n_sel, const int64_t* __restrict__ sel_doc,
                                                         const double* __restrict__ sel_score, int k, int64_t* __restrict__ out_doc,
                                                         double* __restrict__ out_score) {
  const int n = min(*n_sel, k);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double s = sel_score[i];
    const int64_t d = sel_doc[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const double sj = sel_score[j];
      rank += (sj > s || (sj == s && sel_doc[j] < d)) ? 1 : 0;
    }
    out_doc[rank] = d;
    out_score[rank] = s;
  }
}

}  // namespace mrag

using namespace mrag;

extern "C" {

int mrag_bm25_create(int device, int64_t n_docs, int64_t n_terms, const int64_t* indptr, const int32_t* post_doc,
                     const int32_t* post_tf, const double* doc_norm, double k1_plus_1, mrag_handle* out) {
  if (!out) return fail(MRAG_ERR_INVALID, "out is NULL");
  if (n_docs < 0 || n_terms < 0 || n_docs > 0x7FFFFFFFll) return fail(MRAG_ERR_INVALID, "bad sizes");
  if (n_terms && (!indptr || indptr[0] != 0)) return fail(MRAG_ERR_INVALID, "indptr must start at 0");
  const int64_t nnz = n_terms ? indptr[n_terms] : 0;
  if (nnz && (!post_doc || !post_tf)) return fail(MRAG_ERR_INVALID, "NULL postings");
  if (n_docs && !doc_norm) return fail(MRAG_ERR_INVALID, "NULL doc_norm");
  for (int64_t t = 0; t < n_terms; ++t)
    if (indptr[t + 1] < indptr[t]) return fail(MRAG_ERR_INVALID, "indptr not monotone at term %lld", (long long)t);
  MRAG_TRY(use_device(device));
  Bm25Index* ix = new Bm25Index();
  ix->kind = KIND_BM25; ix->device = device;
  ix->n_docs = n_docs; ix->n_terms = n_terms; ix->nnz = nnz; ix->k1p1 = k1_plus_1;
  ix->h_indptr.assign(indptr, indptr + (n_terms ? n_terms + 1 : 0));
  auto up = [&](void** dst, const void* src, size_t bytes) -> int {
    if (!bytes) return MRAG_OK;
    if (hipMalloc(dst, bytes) != hipSuccess) { (void)hipGetLastError(); return fail(MRAG_ERR_OOM, "hipMalloc(%zu) failed", bytes); }
    MRAG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return MRAG_OK;
  };
  int st = up((void**)&ix->indptr, indptr, (size_t)(n_terms ? n_terms + 1 : 0) * 8);
  if (st == MRAG_OK) st = up((void**)&ix->post_doc, post_doc, (size_t)nnz * 4);
  if (st == MRAG_OK) st = up((void**)&ix->post_tf, post_tf, (size_t)nnz * 4);
  if (st == MRAG_OK) st = up((void**)&ix->doc_norm, doc_norm, (size_t)n_docs * 8);
  if (st != MRAG_OK) { delete ix; return st; }
  *out = register_object(ix);
  return MRAG_OK;
}

int mrag_bm25_destroy(mrag_handle h) {
  Object* o = take(h, KIND_BM25);
  if (!o) return MRAG_ERR_INVALID;
  (void)hipSetDevice(o->device);
  (void)hipDeviceSynchronize();
  delete o;
  return MRAG_OK;
}

int mrag_bm25_search(mrag_handle h, int n_queries, const int32_t* q_ptr, const int32_t* q_terms, const double* q_idf,
                     int merge_sum, int k, int64_t* out_docs, double* out_scores, int* out_n, void* stream_) {
  Bm25Index* ix = (Bm25Index*)lookup(h, KIND_BM25);
  if (!ix) return MRAG_ERR_INVALID;
  if (!out_n) return fail(MRAG_ERR_INVALID, "out_n is NULL");
  *out_n = 0;
  if (n_queries < 0 || k < 0) return fail(MRAG_ERR_INVALID, "bad n_queries / k");
  if (k > BM25_MAX_K) return fail(MRAG_ERR_UNSUPPORTED, "k = %d exceeds %d", k, BM25_MAX_K);
  if (n_queries == 0 || k == 0 || ix->n_docs == 0) return MRAG_OK;
  if (!q_ptr || !out_docs || !out_scores) return fail(MRAG_ERR_INVALID, "NULL buffer");
  if (n_queries > 1024) return fail(MRAG_ERR_UNSUPPORTED, "more than 1024 queries per call");
  MRAG_TRY(use_device(ix->device));
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t N = ix->n_docs;
  // jobs per token position: the j-th token of every query that has one (unknown terms, id < 0, add nothing)
  int max_len = 0;
  for (int q = 0; q < n_queries; ++q) {
    if (q_ptr[q + 1] < q_ptr[q]) return fail(MRAG_ERR_INVALID, "q_ptr not monotone");
    max_len = std::max(max_len, q_ptr[q + 1] - q_ptr[q]);
  }
  std::vector<Bm25Job> jobs;
  std::vector<int> pos_first(max_len + 1, 0);
  std::vector<int64_t> pos_longest(max_len, 0);
  for (int j = 0; j < max_len; ++j) {
    pos_first[j] = (int)jobs.size();
    for (int q = 0; q < n_queries; ++q) {
      const int at = q_ptr[q] + j;
      if (at >= q_ptr[q + 1]) continue;
      const int32_t t = q_terms[at];
      if (t < 0) continue;
      if (t >= ix->n_terms) return fail(MRAG_ERR_INVALID, "term id %d outside the vocabulary (%lld)", t, (long long)ix->n_terms);
      const int64_t lo = ix->h_indptr[t], hi = ix->h_indptr[t + 1];
      if (hi == lo) continue;
      jobs.push_back(Bm25Job{lo, hi, q_idf[at], q, 0});
      pos_longest[j] = std::max(pos_longest[j], hi - lo);
    }
  }
  pos_first[max_len] = (int)jobs.size();
  MRAG_TRY(ix->scores.ensure((size_t)n_queries * N * 8));
  MRAG_TRY(ix->fin.ensure((size_t)N * 8));
  MRAG_TRY(ix->hist.ensure(BM25_PASSES * 256 * 4 + 16));
  MRAG_TRY(ix->sel.ensure((size_t)BM25_MAX_K * 16 * 2));
  MRAG_TRY(ix->qbuf.ensure(std::max<size_t>(jobs.size(), 1) * sizeof(Bm25Job)));
  MRAG_HIP(hipMemsetAsync(ix->scores.p, 0, (size_t)n_queries * N * 8, stream));
  MRAG_HIP(hipMemsetAsync(ix->hist.p, 0, BM25_PASSES * 256 * 4 + 16, stream));
  if (!jobs.empty()) MRAG_HIP(hipMemcpyAsync(ix->qbuf.p, jobs.data(), jobs.size() * sizeof(Bm25Job), hipMemcpyHostToDevice, stream));
  for (int j = 0; j < max_len; ++j) {
    const int nj = pos_first[j + 1] - pos_first[j];
    if (!nj) continue;
    const dim3 grid((unsigned)((pos_longest[j] + 255) / 256), (unsigned)nj);
    hipLaunchKernelGGL(bm25_scatter_kernel, grid, dim3(256), 0, stream, (const Bm25Job*)ix->qbuf.p + pos_first[j], ix->post_doc,
                       ix->post_tf, ix->doc_norm, ix->k1p1, N, (double*)ix->scores.p);
  }
  MRAG_HIP(hipGetLastError());
  const unsigned dgrid = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(bm25_merge_kernel, dim3(dgrid), dim3(256), 0, stream, (const double*)ix->scores.p, n_queries, N, merge_sum ? 1 : 0,
                     (double*)ix->fin.p);
  unsigned* hist = (unsigned*)ix->hist.p;
  int* n_sel = (int*)(hist + BM25_PASSES * 256);
  const unsigned sgrid = std::min<unsigned>(dgrid, 1024);
  for (int p = 0; p < BM25_PASSES; ++p)
    hipLaunchKernelGGL(bm25_hist_kernel, dim3(sgrid), dim3(256), 0, stream, (const double*)ix->fin.p, N, p, (int64_t)k, hist);
  int64_t* sel_doc = (int64_t*)ix->sel.p;
  double* sel_score = (double*)(sel_doc + BM25_MAX_K);
  int64_t* out_doc_d = (int64_t*)(sel_score + BM25_MAX_K);
  double* out_score_d = (double*)(out_doc_d + BM25_MAX_K);
  hipLaunchKernelGGL(bm25_gather_kernel, dim3(sgrid), dim3(256), 0, stream, (const double*)ix->fin.p, N, (int64_t)k, (const unsigned*)hist,
                     n_sel, sel_doc, sel_score);
  hipLaunchKernelGGL(bm25_rank_kernel, dim3(1), dim3(1024), 0, stream, (const int*)n_sel, (const int64_t*)sel_doc,
                     (const double*)sel_score, k, out_doc_d, out_score_d);
  MRAG_HIP(hipGetLastError());
  int n = 0;
  MRAG_HIP(hipMemcpyAsync(&n, n_sel, 4, hipMemcpyDeviceToHost, stream));
  MRAG_HIP(hipStreamSynchronize(stream));
  if (n > k) return fail(MRAG_ERR_HIP, "BM25 selection produced %d > k = %d entries", n, k);
  if (n) {
    MRAG_HIP(hipMemcpyAsync(out_docs, out_doc_d, (size_t)n * 8, hipMemcpyDeviceToHost, stream));
    MRAG_HIP(hipMemcpyAsync(out_scores, out_score_d, (size_t)n * 8, hipMemcpyDeviceToHost, stream));
    MRAG_HIP(hipStreamSynchronize(stream));
  }
  *out_n = n;
  return MRAG_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// a7 on the device: the fusion arithmetic of HybridRetrievalBackend.run (retrieval_backend.py:336-372) over the
// three channels' (key, score) lists -- key = the rank of the hit's NORMALISED id among all ids of the call
// (assigned by the host binding, so ascending key == ascending id, the declared tie order):
//   per channel  dedupe by key, strictly larger score wins (:336-348)
//                min-max over the kept scores, all-equal / empty -> 0.0 (:296-301)
//   union        score = a_t*ts + a_g*gs + a_d*ds, left to right, no contraction (:363)
//   sort (score desc, key asc), keep top_k (:371-372)
// One workgroup; the lists are a few hundred entries (pools of 200).  fp64, bit for bit the reference's values.
// ------------------------------------------------------------------------------------------
namespace mrag {

constexpr int FUSE_MAX = 4096;   // entries over the three channels

__global__ __launch_bounds__(1024) void fuse_topk_kernel(const int32_t* __restrict__ key, const double* __restrict__ score,
                                                         const int* __restrict__ chan_ptr /*[4]*/, double a_t, double a_g, double a_d,
                                                         int top_k, int32_t* __restrict__ out_key, double* __restrict__ out /*[4][top_k]*/,
                                                         int* __restrict__ out_n, double* __restrict__ work /*[4][n]: norm | fused | kept flag | carrier flag*/) {
#pragma clang fp contract(off)
  __shared__ double s_lo[3], s_hi[3];
  __shared__ int s_any[3], s_cnt;
  const int tid = threadIdx.x, n = chan_ptr[3];
  double* norm = work;
  double* fused = work + n;
  double* keep = work + 2 * n;     // 1.0 = this entry represents its key in its channel
  double* carrier = work + 3 * n;  // 1.0 = first kept entry of its key over all channels: carries the fused score
  if (tid < 3) { s_any[tid] = 0; }
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  // 1) dedupe inside each channel: an entry is kept unless another entry of the same channel and key has a
  //    strictly larger score, or the same score and a smaller index (the first one met keeps the slot)
  for (int i = tid; i < n; i += blockDim.x) {
    const int c = i >= chan_ptr[2] ? 2 : (i >= chan_ptr[1] ? 1 : 0);
    bool k = true;
    for (int j = chan_ptr[c]; j < chan_ptr[c + 1]; ++j)
      if (j != i && key[j] == key[i] && (score[j] > score[i] || (score[j] == score[i] && j < i))) { k = false; break; }
    keep[i] = k ? 1.0 : 0.0;
  }
  __syncthreads();
  // 2) per-channel min / max over the kept entries (one thread per channel: the lists are short)
  if (tid < 3) {
    double lo = 0.0, hi = 0.0;
    bool any = false;
    for (int j = chan_ptr[tid]; j < chan_ptr[tid + 1]; ++j) {
      if (keep[j] == 0.0) continue;
      const double v = score[j];
      if (!any) { lo = hi = v; any = true; }
      else { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    }
    s_lo[tid] = lo; s_hi[tid] = hi; s_any[tid] = any ? 1 : 0;
  }
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    const int c = i >= chan_ptr[2] ? 2 : (i >= chan_ptr[1] ? 1 : 0);
    const double lo = s_lo[c], hi = s_hi[c];
    norm[i] = (keep[i] != 0.0 && hi > lo) ? (score[i] - lo) / (hi - lo) : 0.0;
  }
  __syncthreads();
  // 3) union: the FIRST kept entry of a key (lowest index over all channels) carries the fused score
  for (int i = tid; i < n; i += blockDim.x) {
    fused[i] = 0.0;
    carrier[i] = 0.0;
    if (keep[i] == 0.0) continue;
    double ch[3] = {0.0, 0.0, 0.0};
    bool first = true;
    for (int j = 0; j < n; ++j) {
      if (keep[j] == 0.0 || key[j] != key[i]) continue;
      if (j < i) { first = false; break; }
      ch[j >= chan_ptr[2] ? 2 : (j >= chan_ptr[1] ? 1 : 0)] = norm[j];
    }
    if (!first) continue;
    fused[i] = a_t * ch[0] + a_g * ch[1] + a_d * ch[2];
    carrier[i] = 1.0;
  }
  __syncthreads();
  // 4) rank the carriers by (fused desc, key asc) and write the top_k
  for (int i = tid; i < n; i += blockDim.x) {
    if (carrier[i] == 0.0) continue;
    int rank = 0;
    for (int j = 0; j < n; ++j)
      if (carrier[j] != 0.0 && (fused[j] > fused[i] || (fused[j] == fused[i] && key[j] < key[i]))) ++rank;
    atomicAdd(&s_cnt, 1);
    if (rank < top_k) {
      double ch[3] = {0.0, 0.0, 0.0};
      for (int j = i; j < n; ++j)
        if (keep[j] != 0.0 && key[j] == key[i]) ch[j >= chan_ptr[2] ? 2 : (j >= chan_ptr[1] ? 1 : 0)] = norm[j];
      out_key[rank] = key[i];
      out[rank] = fused[i];
      out[top_k + rank] = ch[0];
      out[2 * top_k + rank] = ch[1];
      out[3 * top_k + rank] = ch[2];
    }
  }
  __syncthreads();
  if (tid == 0) *out_n = min(s_cnt, top_k);
}

}  // namespace mrag

extern "C" int mrag_fuse_topk(int device, const int32_t* keys, const double* scores, int n_text, int n_graph, int n_dense,
                              double alpha_text, double alpha_graph, double alpha_dense, int top_k, int32_t* out_keys,
                              double* out_scores, double* out_text_norm, double* out_graph_norm, double* out_dense_norm,
                              int* out_n, void* stream_) {
  using namespace mrag;
  if (!out_n) return fail(MRAG_ERR_INVALID, "out_n is NULL");
  *out_n = 0;
  if (n_text < 0 || n_graph < 0 || n_dense < 0 || top_k < 0) return fail(MRAG_ERR_INVALID, "negative size");
  const int n = n_text + n_graph + n_dense;
  if (n > FUSE_MAX) return fail(MRAG_ERR_UNSUPPORTED, "%d fusion entries exceed %d", n, FUSE_MAX);
  if (n == 0 || top_k == 0) return MRAG_OK;
  if (!keys || !scores || !out_keys || !out_scores || !out_text_norm || !out_graph_norm || !out_dense_norm)
    return fail(MRAG_ERR_INVALID, "NULL buffer");
  for (int i = 0; i < n; ++i)
    if (!(scores[i] == scores[i])) return fail(MRAG_ERR_INVALID, "NaN score at fusion entry %d", i);
  MRAG_TRY(use_device(device));
  hipStream_t stream = (hipStream_t)stream_;
  const int kk = std::min(top_k, n);
  // one scratch block: keys | scores | chan_ptr | work[4n] | out_key[kk] | out[4kk] | out_n
  const size_t o_scores = round_up((int64_t)n * 4, 16), o_ptr = o_scores + (size_t)n * 8, o_work = o_ptr + 16,
               o_okey = o_work + (size_t)4 * n * 8, o_out = o_okey + round_up((int64_t)kk * 4, 16), o_n = o_out + (size_t)4 * kk * 8,
               total = o_n + 16;
  char* d = nullptr;
  MRAG_HIP(hipMalloc((void**)&d, total));
  const int ptr[4] = {0, n_text, n_text + n_graph, n};
  hipError_t e = hipMemcpyAsync(d, keys, (size_t)n * 4, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d + o_scores, scores, (size_t)n * 8, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d + o_ptr, ptr, 16, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(fuse_topk_kernel, dim3(1), dim3(1024), 0, stream, (const int32_t*)d, (const double*)(d + o_scores),
                       (const int*)(d + o_ptr), alpha_text, alpha_graph, alpha_dense, kk, (int32_t*)(d + o_okey), (double*)(d + o_out),
                       (int*)(d + o_n), (double*)(d + o_work));
    e = hipGetLastError();
  }
  int n_out = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&n_out, d + o_n, 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e == hipSuccess && n_out > 0) {
    e = hipMemcpy(out_keys, d + o_okey, (size_t)n_out * 4, hipMemcpyDeviceToHost);
    double* o = (double*)(d + o_out);
    if (e == hipSuccess) e = hipMemcpy(out_scores, o, (size_t)n_out * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_text_norm, o + kk, (size_t)n_out * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_graph_norm, o + 2 * kk, (size_t)n_out * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_dense_norm, o + 3 * kk, (size_t)n_out * 8, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d);
  if (e != hipSuccess) return fail(MRAG_ERR_HIP, "fuse_topk failed: %s", hipGetErrorString(e));
  *out_n = n_out;
  return MRAG_OK;
}
