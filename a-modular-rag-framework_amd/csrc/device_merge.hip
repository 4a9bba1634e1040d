// mrag_topk_merge_device: the 8e merge on the device, for gathered partial top-k that already sits in
// HBM (the RCCL all-gather output).  Same order as mrag_topk_merge (host_merge.cpp): (score desc,
// id asc), empty slots (id < 0) last.  HBM-bound and tiny (C4 at 8 GPUs: 9.6 MB in, 1.2 MB out), so the
// design is simply "never leave the device": one wave per query, candidates staged in LDS, every
// candidate's final rank = its index in its own (sorted) part + a binary search in each other part.
#include "common.h"

namespace mrag {
namespace {

constexpr int DM_MAX_CAND = 2048;   // nparts * k staged per query (24 KiB of LDS)

struct DmParams {
  const float* scores;    // [nparts][nq][k]
  const int64_t* ids;     // [nparts][nq][k]
  const int64_t* words;   // packed form instead of (scores, ids): (score bits << 32) | shard-local row, 0xFFFFFFFF = empty slot
  const int64_t* bases;   // packed form: first global row of every part [nparts]
  int nparts, k;
  int64_t nq;
  float* out_scores;      // [nq][k]
  int64_t* out_ids;       // [nq][k]
};

// strict total order: does candidate a come before candidate b?
__device__ __forceinline__ bool dm_before(float sa, int64_t ia, int pa, float sb, int64_t ib, int pb) {
  if (sa != sb) return sa > sb;
  if (ia != ib) return ia < ib;
  return pa < pb;
}

__global__ __launch_bounds__(64) void topk_merge_device_kernel(DmParams p) {
  __shared__ float s_sc[DM_MAX_CAND];
  __shared__ int64_t s_id[DM_MAX_CAND];
  __shared__ int s_len[64];           // valid entries per part (nparts <= 64)
  const int64_t q = blockIdx.x;
  const int lane = threadIdx.x;
  const int k = p.k, np = p.nparts, total = np * k;
  for (int c = lane; c < total; c += 64) {
    const int part = c / k, i = c - part * k;
    const size_t src = ((size_t)part * p.nq + q) * k + i;
    if (p.words) {   // the all-gather's packed words, unpacked on the way into LDS (no separate unpack pass over the gathered buffer)
      const int64_t wv = p.words[src];
      const uint32_t low = (uint32_t)(wv & 0xFFFFFFFFll);
      s_sc[c] = __uint_as_float((uint32_t)((uint64_t)wv >> 32));
      s_id[c] = low == 0xFFFFFFFFu ? (int64_t)-1 : (int64_t)low + p.bases[part];
    } else {
      s_sc[c] = p.scores[src];
      s_id[c] = p.ids[src];
    }
  }
  __syncthreads();
  if (lane < np) {   // valid prefix length of the (sorted, empties last) part
    int lo = 0, hi = k;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_id[lane * k + mid] >= 0) lo = mid + 1; else hi = mid; }
    s_len[lane] = lo;
  }
  __syncthreads();
  int valid = 0;
  for (int pp = 0; pp < np; ++pp) valid += s_len[pp];
  for (int c = lane; c < total; c += 64) {
    const int part = c / k, i = c - part * k;
    if (i >= s_len[part]) continue;
    const float sc = s_sc[c];
    const int64_t id = s_id[c];
    int rank = i;
    for (int pp = 0; pp < np && rank < k; ++pp) {
      if (pp == part) continue;
      int lo = 0, hi = s_len[pp];   // entries of part pp that come before this candidate
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (dm_before(s_sc[pp * k + mid], s_id[pp * k + mid], pp, sc, id, part)) lo = mid + 1; else hi = mid;
      }
      rank += lo;
    }
    if (rank < k) {
      p.out_scores[q * k + rank] = sc;
      p.out_ids[q * k + rank] = id;
    }
  }
  for (int r = valid + lane; r < k; r += 64) {
    p.out_scores[q * k + r] = -INFINITY;
    p.out_ids[q * k + r] = -1;
  }
}

// (score, global id) -> packed word of the exchange; ids outside [id_base, id_base + 2^32 - 1) raise a flag (never wrapped)
__global__ void pack_partial_kernel(const float* __restrict__ scores, const int64_t* __restrict__ ids, int64_t n, int64_t id_base,
                                    int64_t* __restrict__ words, int* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t id = ids[i];
  uint32_t low = 0xFFFFFFFFu;
  if (id >= 0) {
    const int64_t local = id - id_base;
    if (local < 0 || local >= 0xFFFFFFFFll) { if (bad) atomicOr(bad, 1); }
    low = (uint32_t)local;
  }
  words[i] = (int64_t)(((uint64_t)__float_as_uint(scores[i]) << 32) | (uint64_t)low);
}

}  // namespace
}  // namespace mrag

extern "C" int mrag_pack_partial_device(int device, const float* scores, const int64_t* ids, int64_t n, int64_t id_base,
                                        int64_t* words, int* bad_flag, void* stream) {
  using namespace mrag;
  if (n < 0) return fail(MRAG_ERR_INVALID, "bad n");
  if (n == 0) return MRAG_OK;
  if (!scores || !ids || !words) return fail(MRAG_ERR_INVALID, "NULL buffer");
  MRAG_TRY(use_device(device));
  hipLaunchKernelGGL(pack_partial_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, ids, n, id_base, words, bad_flag);
  MRAG_HIP(hipGetLastError());
  return MRAG_OK;
}

extern "C" int mrag_topk_merge_device(int device, const float* scores, const int64_t* ids, int nparts, int64_t nq, int k,
                                      float* out_scores, int64_t* out_ids, void* stream) {
  using namespace mrag;
  if (!scores || !ids || !out_scores || !out_ids) return fail(MRAG_ERR_INVALID, "NULL buffer");
  if (nparts <= 0 || nq < 0 || k <= 0) return fail(MRAG_ERR_INVALID, "nparts = %d, nq = %lld, k = %d", nparts, (long long)nq, k);
  if (nparts > 64 || (int64_t)nparts * k > DM_MAX_CAND)
    return fail(MRAG_ERR_UNSUPPORTED, "device merge serves nparts <= 64 and nparts * k <= %d (got %d x %d); use mrag_topk_merge",
                DM_MAX_CAND, nparts, k);
  MRAG_TRY(use_device(device));
  if (nq == 0) return MRAG_OK;
  DmParams p{scores, ids, nullptr, nullptr, nparts, k, nq, out_scores, out_ids};
  hipLaunchKernelGGL(topk_merge_device_kernel, dim3((unsigned)nq), dim3(64), 0, (hipStream_t)stream, p);
  MRAG_HIP(hipGetLastError());
  return MRAG_OK;
}

extern "C" int mrag_topk_merge_packed_device(int device, const int64_t* words, const int64_t* bases, int nparts, int64_t nq, int k,
                                             float* out_scores, int64_t* out_ids, void* stream) {
  using namespace mrag;
  if (!words || !bases || !out_scores || !out_ids) return fail(MRAG_ERR_INVALID, "NULL buffer");
  if (nparts <= 0 || nq < 0 || k <= 0) return fail(MRAG_ERR_INVALID, "nparts = %d, nq = %lld, k = %d", nparts, (long long)nq, k);
  if (nparts > 64 || (int64_t)nparts * k > DM_MAX_CAND)
    return fail(MRAG_ERR_UNSUPPORTED, "device merge serves nparts <= 64 and nparts * k <= %d (got %d x %d); use mrag_topk_merge",
                DM_MAX_CAND, nparts, k);
  MRAG_TRY(use_device(device));
  if (nq == 0) return MRAG_OK;
  DmParams p{nullptr, nullptr, words, bases, nparts, k, nq, out_scores, out_ids};
  hipLaunchKernelGGL(topk_merge_device_kernel, dim3((unsigned)nq), dim3(64), 0, (hipStream_t)stream, p);
  MRAG_HIP(hipGetLastError());
  return MRAG_OK;
}
