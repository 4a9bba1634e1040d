// Host-side merge of per-shard partial top-k: after the RCCL
// all-gather every rank holds [nparts, nq, k] (score, global id) pairs; per query keep
// the k best by (score desc, id asc).  Plain C++ threads; no GPU involved.
#include "common.h"

#include <math.h>
#include <algorithm>
#include <thread>

using namespace mrag;

namespace {

struct Cand {
  float s;
  int64_t id;
};

inline bool better(const Cand& a, const Cand& b) { return a.s > b.s || (a.s == b.s && a.id < b.id); }

void merge_range(const float* scores, const int64_t* ids, int nparts, int64_t nq, int k, float* out_scores,
                 int64_t* out_ids, int64_t q_lo, int64_t q_hi) {
  std::vector<Cand> buf((size_t)nparts * k);
  for (int64_t q = q_lo; q < q_hi; ++q) {
    size_t m = 0;
    for (int p = 0; p < nparts; ++p) {
      const size_t base = ((size_t)p * nq + q) * k;
      for (int i = 0; i < k; ++i) {
        const int64_t id = ids[base + i];
        const float s = scores[base + i];
        if (id >= 0 && !(s != s)) buf[m++] = Cand{s, id};
      }
    }
    const size_t keep = std::min<size_t>(m, (size_t)k);
    std::partial_sort(buf.begin(), buf.begin() + keep, buf.begin() + m, better);
    for (size_t i = 0; i < keep; ++i) {
      out_scores[q * k + i] = buf[i].s;
      out_ids[q * k + i] = buf[i].id;
    }
    for (size_t i = keep; i < (size_t)k; ++i) {
      out_scores[q * k + i] = -INFINITY;
      out_ids[q * k + i] = -1;
    }
  }
}

}  // namespace

extern "C" int mrag_topk_merge(const float* scores, const int64_t* ids, int nparts, int64_t nq, int k,
                               float* out_scores, int64_t* out_ids, int nthreads) {
  if (nparts <= 0 || nq < 0 || k <= 0) return fail(MRAG_ERR_INVALID, "bad merge shape nparts=%d nq=%lld k=%d", nparts, (long long)nq, k);
  if (nq == 0) return MRAG_OK;
  if (!scores || !ids || !out_scores || !out_ids) return fail(MRAG_ERR_INVALID, "NULL buffer");
  if (nthreads <= 0) nthreads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  nthreads = (int)std::min<int64_t>(nthreads, std::max<int64_t>(1, nq / 256));
  if (nthreads <= 1) {
    merge_range(scores, ids, nparts, nq, k, out_scores, out_ids, 0, nq);
    return MRAG_OK;
  }
  std::vector<std::thread> th;
  for (int i = 0; i < nthreads; ++i) {
    const int64_t lo = nq * i / nthreads, hi = nq * (i + 1) / nthreads;
    th.emplace_back(merge_range, scores, ids, nparts, nq, k, out_scores, out_ids, lo, hi);
  }
  for (auto& t : th) t.join();
  return MRAG_OK;
}
