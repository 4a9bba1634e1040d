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

// Per-shard lists arrive sorted by (score desc, id asc) (K4 writes them that way), so the merge is a
// k-way head comparison: k * nparts compares per query.  A list that is NOT sorted (generic callers)
// sends its query through partial_sort instead.
void merge_range(const float* scores, const int64_t* ids, int nparts, int64_t nq, int k, float* out_scores,
                 int64_t* out_ids, int64_t q_lo, int64_t q_hi) {
  std::vector<Cand> buf((size_t)nparts * k);
  std::vector<int> head(nparts), len(nparts);
  for (int64_t q = q_lo; q < q_hi; ++q) {
    bool sorted = true;
    for (int p = 0; p < nparts; ++p) {
      const size_t base = ((size_t)p * nq + q) * k;
      int n = 0;
      while (n < k && ids[base + n] >= 0 && !(scores[base + n] != scores[base + n])) ++n;
      for (int i = n; i < k; ++i) sorted &= ids[base + i] < 0;                 // empties only at the tail
      for (int i = 1; i < n; ++i)
        sorted &= !(scores[base + i] > scores[base + i - 1] ||
                    (scores[base + i] == scores[base + i - 1] && ids[base + i] < ids[base + i - 1]));
      head[p] = 0;
      len[p] = n;
    }
    float* os = out_scores + q * k;
    int64_t* oi = out_ids + q * k;
    int n_out = 0;
    if (sorted) {
      for (; n_out < k; ++n_out) {
        int bp = -1;
        float bs = 0.f;
        int64_t bi = 0;
        for (int p = 0; p < nparts; ++p) {
          if (head[p] >= len[p]) continue;
          const size_t at = ((size_t)p * nq + q) * k + head[p];
          const float s = scores[at];
          const int64_t id = ids[at];
          if (bp < 0 || s > bs || (s == bs && id < bi)) { bp = p; bs = s; bi = id; }
        }
        if (bp < 0) break;
        ++head[bp];
        os[n_out] = bs;
        oi[n_out] = bi;
      }
    } else {
      size_t m = 0;
      for (int p = 0; p < nparts; ++p) {
        const size_t base = ((size_t)p * nq + q) * k;
        for (int i = 0; i < k; ++i)
          if (ids[base + i] >= 0 && !(scores[base + i] != scores[base + i])) buf[m++] = Cand{scores[base + i], ids[base + i]};
      }
      const size_t keep = std::min<size_t>(m, (size_t)k);
      std::partial_sort(buf.begin(), buf.begin() + keep, buf.begin() + m, better);
      for (size_t i = 0; i < keep; ++i) { os[i] = buf[i].s; oi[i] = buf[i].id; }
      n_out = (int)keep;
    }
    for (int i = n_out; i < k; ++i) { os[i] = -INFINITY; oi[i] = -1; }
  }
}

}  // namespace

extern "C" int mrag_topk_merge(const float* scores, const int64_t* ids, int nparts, int64_t nq, int k,
                               float* out_scores, int64_t* out_ids, int nthreads) {
  if (nparts <= 0 || nq < 0 || k <= 0) return fail(MRAG_ERR_INVALID, "bad merge shape nparts=%d nq=%lld k=%d", nparts, (long long)nq, k);
  if (nq == 0) return MRAG_OK;
  if (!scores || !ids || !out_scores || !out_ids) return fail(MRAG_ERR_INVALID, "NULL buffer");
  if (nthreads <= 0) nthreads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  nthreads = (int)std::min<int64_t>(nthreads, std::max<int64_t>(1, nq / 1024));
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
